#!/usr/bin/env python3
"""Diagnostic: time of the fused kernel with phases switched off (TP_FUSED_DBG bits: 1 no A1, 2 no K2P2, 4 no A6)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photometry_amd import simulate, engine, pipeline
from photometry_amd.device import Context

ctx = Context(0)
Nt = int(os.environ.get('NT', 10000))
scene = simulate.make_scene(Nt, 1300, 15, 15, seed=1000)
scene.aperture = None
cubes = engine.synth_fill(ctx, scene)
batch = pipeline.ApertureBatch(ctx, scene, cubes=cubes)
work = pipeline.ApertureWork(ctx, batch)

def timeit(fn, n=10):
	for _ in range(3):
		fn()
	ctx.sync()
	t0 = time.perf_counter()
	for _ in range(n):
		fn()
	ctx.sync()
	return (time.perf_counter() - t0) / n * 1e3

print('three kernels', round(timeit(lambda: pipeline.aperture_step(ctx, batch, work, fused=False)), 3))
for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
	os.environ['TP_FUSED_DBG'] = str(dbg)
	print('dbg', dbg, round(timeit(lambda: pipeline.aperture_step(ctx, batch, work)), 3), flush=True)
