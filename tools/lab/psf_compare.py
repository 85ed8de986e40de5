#!/usr/bin/env python3
"""lab: tp_psf_fit of the product library against a lab build (TP_OTHER_LIB) on the same scene: iteration counts and fluxes, bit for bit;
the product library twice (is it reproducible?)."""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
	from photometry_amd import _lib
	if sys.argv[1] != 'product':
		_lib.LIB_PATH = sys.argv[1]
	from photometry_amd import simulate, engine, psf as hpsf
	from photometry_amd.device import Context, DeviceCube
	from photometry_amd.plugins import psf_star_selection, mag2flux
	Nt, T, H, W = int(os.environ.get('NT', 256)), int(os.environ.get('T', 6)), 15, 15
	ctx = Context(0)
	s = simulate.make_scene(Nt, T, H, W, seed=7)
	simulate.fill_cubes(s, nan_fraction=0.001)
	prf = simulate.synthetic_prf(seed=1)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	offs, params, mini = [0], [], []
	for i in range(Nt):
		c = s.catalog_of(i)
		sel = psf_star_selection(c['row_stamp'], c['column_stamp'], c['tmag'], s.target_pos_row[i] - s.stamps[i][0], s.target_pos_column[i] - s.stamps[i][2], s.target_tmag[i])
		params.append(np.column_stack((c['row_stamp'][sel].astype('float64'), c['column_stamp'][sel].astype('float64'), mag2flux(c['tmag'][sel].astype('float64')))))
		offs.append(offs[-1] + len(sel))
		m = np.zeros((H, W), dtype='uint8'); m[5:10, 5:10] = 1
		mini.append(m)
	coef = engine.linpsf_prf(ctx, ctx.array(model.base_coef), ctx.array(model.weights(s.stamps)))
	args = (DeviceCube.from_host(ctx, s.images), DeviceCube.from_host(ctx, s.backgrounds), coef, ctx.array(model.tx), ctx.array(model.ty),
		ctx.array(np.asarray(offs, dtype='int64')), ctx.array(np.concatenate(params)), ctx.array(np.stack(mini)))
	res = engine.psf_fit(ctx, *args)
	ctx.sync()
	np.savez(sys.argv[2], nit=res['nit'].to_host()[:, :T], flux=res['flux'].to_host()[:, :T], ns=np.diff(offs))
	sys.exit(0)
here = os.path.abspath(__file__)
first_lib = os.environ.get('TP_FIRST_LIB', 'product')
runs = [(first_lib, '/tmp/psf_a.npz'), (first_lib, '/tmp/psf_b.npz'), (os.environ['TP_OTHER_LIB'], '/tmp/psf_c.npz')]
for lib, out in runs:
	subprocess.run([sys.executable, here, lib, out], check=True, timeout=300)
a, b, c = (np.load(o) for _, o in runs)
print('product twice: iteration counts equal', np.array_equal(a['nit'], b['nit']), '; fluxes equal', np.array_equal(a['flux'], b['flux'], equal_nan=True))
same = a['nit'] == c['nit']
print('product vs other: cadences with equal iteration counts %.4f; fluxes bit-equal %.4f' % (same.mean(), np.mean((a['flux'] == c['flux']) | (np.isnan(a['flux']) & np.isnan(c['flux'])))))
bad = np.argwhere(~same)
for t, k in bad[:8]:
	print('  target', t, 'stars', a['ns'][t], 'cadence', k, 'iterations', a['nit'][t, k], c['nit'][t, k], 'flux', a['flux'][t, k], c['flux'][t, k])
first = {}
for t, k in bad:
	first.setdefault(int(t), int(k))
print('targets that differ:', len(first), '; by stars:', {int(n): int(sum(1 for t in first if a['ns'][t] == n)) for n in np.unique(a['ns'])}, '; first differing cadence histogram', np.bincount(list(first.values()))[:10] if first else None)
