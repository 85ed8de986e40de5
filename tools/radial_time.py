#!/usr/bin/env python3
"""Diagnostic: time of fit_background_frames on full 2048 x 2048 frames, plain and TESS (radial) branch, per kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from photometry_amd import prepare
from photometry_amd.device import Context

T = int(os.environ.get('NF', 8))
ctx = Context(0)
rng = np.random.default_rng(0)
xc, yc = prepare.CAMERA_CENTRE[(1, 1)]
yy, xx = np.mgrid[0:2048, 0:2048]
r = np.hypot(xx + 44 - xc, yy - yc)
f = np.empty((T, 2048, 2048), dtype='float32')
for k in range(T):
	f[k] = 120 + 0.02 * xx + 40 * np.exp((r - 2400) / 250.0) + rng.normal(0, 4, r.shape)
d = ctx.array(f)
geo = prepare.RadialGeometry((2048, 2048), 1, 1)
MODE = os.environ.get('MODE')   # 'plain' / 'tess': that branch only (the PMC passes of profiles/run_profile.sh: every dispatch then belongs to it)
for name, kw in [(n, k) for n, k in (('plain', {}), ('tess', dict(geometry=geo))) if MODE in (None, n)]:
	prepare.fit_background_frames(ctx, d, **kw).free()
	ctx.profile(True)
	ctx.profile_reset()
	t0 = time.perf_counter()
	out = prepare.fit_background_frames(ctx, d, **kw)
	ctx.sync()
	dt = time.perf_counter() - t0
	print(name, 'ms per frame', round(dt / T * 1e3, 2))
	for kname, (n, ms) in ctx.profile_report().items():
		print('   ', kname, n, 'launches', round(ms / T, 3), 'ms per frame')
	ctx.profile(False)
	b = out.to_host()
	print('    corner / centre background', float(b[0, 0, 0]), float(b[0, 2047, 2047]))
