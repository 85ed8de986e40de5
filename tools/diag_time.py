#!/usr/bin/env python3
"""Diagnostic: time of tp_lightcurve_diagnostics on 10 000 light curves x 1300 cadences (after one aperture step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photometry_amd import simulate, engine, pipeline
from photometry_amd.device import Context
ctx = Context(0)
Nt, T = int(os.environ.get('NT', 10000)), 1300
scene = simulate.make_scene(Nt, T, 15, 15, seed=1000)
scene.aperture = None
cubes = engine.synth_fill(ctx, scene, images=True, images_err=True, backgrounds=True, raw=False)
b = pipeline.ApertureBatch(ctx, scene, cubes=cubes)
w = pipeline.ApertureWork(ctx, b)
pipeline.aperture_step(ctx, b, w)
for _ in range(2):
	pipeline.aperture_diagnostics(ctx, b, w)
ctx.sync()
t0 = time.perf_counter()
for _ in range(5):
	pipeline.aperture_diagnostics(ctx, b, w)
ctx.sync()
print('diagnostics', Nt, 'targets:', round((time.perf_counter() - t0) / 5 * 1e3, 3), 'ms', flush=True)
