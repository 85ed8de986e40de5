#!/usr/bin/env python3
"""Diagnostic: time of the stamp-background kernel on the C3 raw cube."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photometry_amd import simulate, engine
from photometry_amd.device import Context

ctx = Context(0)
Nt = int(os.environ.get('NT', 10000))
scene = simulate.make_scene(Nt, 1300, 15, 15, seed=1000)
cubes = engine.synth_fill(ctx, scene, images=False, images_err=False, backgrounds=False, raw=True)
raw = cubes['raw']
out = ctx.zeros((Nt, raw.t_pitch), 'float32')
for _ in range(2):
	engine.background_stamp(ctx, raw, out=out)
ctx.sync()
t0 = time.perf_counter()
for _ in range(5):
	engine.background_stamp(ctx, raw, out=out)
ctx.sync()
print('B* ms', round((time.perf_counter() - t0) / 5 * 1e3, 3))
