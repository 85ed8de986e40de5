# -*- coding: utf-8 -*-
"""PRF sample sets shared by the LinPSF / PSFPhotometry tests (test infrastructure: uses the oracle's synthetic PRF)."""
import numpy as np


def general_prf(kind):
	"""PRF samples on grids the uniform-grid kernels cannot take (psf.py:119 builds its spline on whatever the file holds)."""
	from oracle import psf as opsf
	if kind == 'spoc':
		return opsf.synthetic_prf(seed=7)
	if kind == 'nsub7':
		return opsf.synthetic_prf(seed=7, nsub=7)
	if kind == 'coarse':      # 24 samples per axis over +-4 px: fewer coefficients than the fast kernels' 13-wide windows need
		return opsf.synthetic_prf(seed=7, nsub=3, halfwidth=4.0)
	if kind == 'rect':        # axes of different lengths: 7 samples per pixel along the columns, 9 along the rows (91 x 117 coefficients)
		a, b = opsf.synthetic_prf(seed=7, nsub=7), opsf.synthetic_prf(seed=7)
		x, y = a['prfColumn'], b['prfRow']
		rng = np.random.default_rng(11)
		vals = np.empty((b['values'].shape[0], len(x), len(y)))
		for i in range(vals.shape[0]):
			sx, sy = 0.9 * (1 + 0.08 * rng.standard_normal()), 0.9 * (1 + 0.08 * rng.standard_normal())
			vals[i] = np.outer(np.exp(-0.5 * (x / sx)**2), np.exp(-0.5 * (y / sy)**2)) + 1e-4 * np.outer(np.exp(-0.5 * (x / (3 * sx))**2), np.exp(-0.5 * (y / (3 * sy))**2))
		b['values'], b['prfColumn'], b['prfRow'] = vals, x, y
		return b
	assert kind == 'warped'   # unevenly spaced samples: every knot interval has its own length
	prf = opsf.synthetic_prf(seed=7)
	x = prf['prfColumn']
	xw = x + 0.03 * np.sin(2.1 * x + 0.4)
	yw = x + 0.02 * np.cos(1.3 * x)
	assert np.all(np.diff(xw) > 0) and np.all(np.diff(yw) > 0)
	rng = np.random.default_rng(7)
	vals = np.empty_like(prf['values'])
	for i in range(vals.shape[0]):
		sx, sy = 0.9 * (1 + 0.08 * rng.standard_normal()), 0.9 * (1 + 0.08 * rng.standard_normal())
		vals[i] = np.outer(np.exp(-0.5 * (xw / sx)**2), np.exp(-0.5 * (yw / sy)**2)) + 1e-4 * np.outer(np.exp(-0.5 * (xw / (3 * sx))**2), np.exp(-0.5 * (yw / (3 * sy))**2))
	prf['values'], prf['prfColumn'], prf['prfRow'] = vals, xw, yw
	return prf
