#!/usr/bin/env python3
"""Diagnostic: the matrix-core LinPSF fit (path 1) against the vector-ALU kernels (path 0, held to the oracle by
tests/test_gpu_linpsf.py) over many random scenes -- stamp sizes, series lengths, neighbour counts, jitter scales (1-3 knot
intervals visited, and more: the fallback), NaN pixels, NaN positions.  Prints the largest deviation per scene kind.
  SEEDS=0..40 python tools/linpsf_fuzz.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from photometry_amd import simulate, engine, psf as hpsf
from photometry_amd.device import Context, DeviceCube

lo, hi = [int(x) for x in os.environ.get('SEEDS', '0..24').split('..')]
ctx = Context(0)
prf = simulate.synthetic_prf(seed=3)
model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
worst = {}
for seed in range(lo, hi):
	rng = np.random.default_rng(1000 + seed)
	H, W = int(rng.integers(9, 18)), int(rng.integers(9, 18))
	T = int(rng.choice([17, 33, 64, 100, 257, 700]))
	neigh = int(rng.integers(0, 7))
	jit = float(rng.choice([0.3, 1.0, 1.0, 2.0, 3.0, 5.0]))
	nt = int(rng.integers(20, 90))
	s = simulate.make_scene(nt, T, H, W, seed=2000 + seed, max_neighbours=neigh, neighbour_tmag_range=(8.5, 17.0))
	s.jitter = s.jitter * jit
	simulate.fill_cubes(s, nan_fraction=float(rng.choice([0.0, 0.002, 0.02])))
	sel, star_offsets, target_index = hpsf.select_stars(s.catalog, s.cat_offsets, s.target_starid)
	rs, cs = s.catalog['row_stamp'][sel].astype('float64'), s.catalog['column_stamp'][sel].astype('float64')
	pos_row = rs[:, None] + s.jitter[None, :, 1]
	pos_col = cs[:, None] + s.jitter[None, :, 0]
	if seed % 5 == 0 and len(rs) > 1:     # a star without a position at some cadences
		pos_row[1, ::7] = np.nan
	max_stars = int(np.diff(star_offsets).max())
	coef = engine.linpsf_prf(ctx, ctx.array(model.base_coef), ctx.array(model.weights(s.stamps)))
	cube = DeviceCube.from_host(ctx, s.images)
	out = {}
	for path in (0, 1):
		engine.linpsf_set_path(ctx, path)
		out[path] = engine.linpsf_fit(ctx, cube, coef, ctx.array(model.tx), ctx.array(model.ty), ctx.array(star_offsets), ctx.array(target_index),
			ctx.array(pos_row), ctx.array(pos_col), max_stars).to_host()
	engine.linpsf_set_path(ctx, 1)
	a, b = out[0], out[1]
	scale = np.nanmax(np.abs(a['flux']), axis=1, keepdims=True)
	scale[~(scale > 0)] = 1.0
	assert np.array_equal(np.isnan(a['flux']), np.isnan(b['flux'])), seed
	dev = float(np.nanmax(np.abs(a['flux'] - b['flux']) / scale)) if np.isfinite(a['flux']).any() else 0.0
	okc = np.isfinite(a['contamination'])
	devc = float(np.max(np.abs(a['contamination'][okc] - b['contamination'][okc]) / np.maximum(np.abs(a['contamination'][okc]), 1e-6))) if okc.any() else 0.0
	assert np.array_equal(a['status'], b['status']), (seed, np.flatnonzero(a['status'] != b['status']))
	key = f'jitter x{jit}'
	worst[key] = max(worst.get(key, 0.0), dev)
	print(f'seed {seed}: {nt} targets {H}x{W}x{T}, <= {neigh} neighbours ({max_stars} stars at most), jitter x{jit}: flux {dev:.2e}, contamination {devc:.2e}', flush=True)
	assert dev < 1e-8 and devc < 1e-7, seed
print('largest relative flux deviation between the two mappings, by jitter scale:', {k: f'{v:.1e}' for k, v in sorted(worst.items())})
