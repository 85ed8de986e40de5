#!/usr/bin/env python3
"""Diagnostic: time of the non-linear PSF photometry kernel (tp_psf_fit) on NT targets x T cadences x 15x15."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get('TP_LAB_LIB'):
	from photometry_amd import _lib
	_lib.LIB_PATH = os.environ['TP_LAB_LIB']
from photometry_amd import simulate, engine, psf as hpsf
from photometry_amd.device import Context, DeviceCube
from photometry_amd.plugins import psf_star_selection, mag2flux

Nt, T, H, W = int(os.environ.get('NT', 512)), int(os.environ.get('T', 200)), 15, 15
ctx = Context(0)
s = simulate.make_scene(Nt, T, H, W, seed=7)
simulate.fill_cubes(s, nan_fraction=0.001)
prf = simulate.synthetic_prf(seed=1)
model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
offs, params, mini = [0], [], []
for i in range(Nt):
	c = s.catalog_of(i)
	sel = psf_star_selection(c['row_stamp'], c['column_stamp'], c['tmag'], s.target_pos_row[i] - s.stamps[i][0], s.target_pos_column[i] - s.stamps[i][2], s.target_tmag[i])
	if os.environ.get('MAXSTARS'):
		sel = sel[:int(os.environ['MAXSTARS'])]
	params.append(np.column_stack((c['row_stamp'][sel].astype('float64'), c['column_stamp'][sel].astype('float64'), mag2flux(c['tmag'][sel].astype('float64')))))
	offs.append(offs[-1] + len(sel))
	m = np.zeros((H, W), dtype='uint8')
	r, cc = int(round(s.target_pos_row[i] - s.stamps[i][0])), int(round(s.target_pos_column[i] - s.stamps[i][2]))
	m[max(r-1, 0):r+2, max(cc-1, 0):cc+2] = 1
	mini.append(m)
coef = engine.linpsf_prf(ctx, ctx.array(model.base_coef), ctx.array(model.weights(s.stamps)))
args = (DeviceCube.from_host(ctx, s.images), DeviceCube.from_host(ctx, s.backgrounds), coef, ctx.array(model.tx), ctx.array(model.ty),
	ctx.array(np.asarray(offs, dtype='int64')), ctx.array(np.concatenate(params)), ctx.array(np.stack(mini)))
res = engine.psf_fit(ctx, *args)
ctx.sync()
t0 = time.perf_counter()
res = engine.psf_fit(ctx, *args)
ctx.sync()
dt = time.perf_counter() - t0
nit = res['nit'].to_host()
print(f'{Nt} targets x {T} cadences, {offs[-1]} fitted stars: {dt*1e3:.1f} ms = {Nt/dt:.0f} targets/s; mean iterations {nit.mean():.0f}; '
	f'{dt / max(nit.sum(), 1) * 1e9:.0f} ns per simplex iteration')
if os.environ.get('STATS'):
	ns_of = np.diff(np.asarray(offs))
	tot = nit[:, :T].sum(axis=1)
	print('total iterations', int(tot.sum()), '; slots 768 -> ideal share per slot', int(tot.sum() / 768))
	for k in range(1, 6):
		sel = ns_of == k
		if sel.any():
			print(f'  {k} stars: {int(sel.sum())} targets, iterations per target: mean {tot[sel].mean():.0f}, max {tot[sel].max():.0f}, share of all iterations {tot[sel].sum() / tot.sum():.2f}')
# CLOCKS=1: phases of an iteration from in-kernel clocks -- a scratch build (tools/lab/clock_hooks.patch: made against the round-4 sources, to be re-made before use):
# SRC=psfphot.hip bash tools/lab/build_variants.sh "psfclk:-DTP_LAB_PSF_CLOCK"; git apply -R tools/lab/clock_hooks.patch
if os.environ.get('TP_LAB_LIB') and os.environ.get('CLOCKS'):
	import ctypes
	buf = (ctypes.c_longlong * 8)()
	ctx.lib.tp_lab_psf_clocks(buf)
	c = list(buf)
	it = max(c[5], 1)
	print('workgroup 0 of every launch, both calls: iterations', c[5], '; cycles per iteration (100 MHz clock64 ticks x 24 at 2.4 GHz?):')
	print('  loop total %.0f; prepare %.0f, rebuild %.0f, pixels %.0f, reduce %.0f ticks per iteration' % (c[4] / it, c[0] / it, c[1] / it, c[2] / it, c[3] / it))
if os.environ.get('DUMP'):   # the fit's outputs, for a bit-by-bit comparison of two builds
	np.savez(os.environ['DUMP'], flux=res['flux'].to_host(), nit=nit, cr=res['centroid_row'].to_host(), cc=res['centroid_col'].to_host())
