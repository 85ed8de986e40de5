#!/usr/bin/env python3
"""Diagnostic: fused aperture kernel time for NT targets (launch-geometry / tail effects); run once per NT."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photometry_amd import simulate, engine, pipeline
from photometry_amd.device import Context

ctx = Context(0)
T = 1300
Nt = int(os.environ.get('NT', 10000))
scene = simulate.make_scene(Nt, T, 15, 15, seed=1000)
scene.aperture = None
cubes = engine.synth_fill(ctx, scene, images=False, images_err=True, backgrounds=False, raw=True)
b = pipeline.ApertureBatch(ctx, scene, cubes={'raw': cubes['raw'], 'raw_err': cubes['images_err']})
w = pipeline.ApertureWork(ctx, b, packed=True)
engine.background_stamp(ctx, b.images, out=w.bkg_raw)
engine.smooth_time(ctx, w.bkg_raw, b.n_cad, b.time_smooth, out=w.bkg)
for _ in range(2):
	engine.aperture_photometry(ctx, b, w, subtract=w.bkg, backgrounds=w.bkg)
ctx.sync()
t0 = time.perf_counter()
for _ in range(5):
	engine.aperture_photometry(ctx, b, w, subtract=w.bkg, backgrounds=w.bkg)
ctx.sync()
ms = (time.perf_counter() - t0) / 5 * 1e3
print(Nt, 'targets', round(ms, 3), 'ms', round(ms / Nt * 1e3, 4), 'us/target', flush=True)
