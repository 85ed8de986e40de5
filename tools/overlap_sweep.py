#!/usr/bin/env python3
"""Diagnostic: step time of the chunked multi-stream aperture pipeline vs chunk / stream counts (no per-kernel events)."""
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

print('serial', round(timeit(lambda: pipeline.aperture_step(ctx, batch, work)), 3))
for cfg in sys.argv[1:]:
	ch, ms, pr = (int(x) for x in cfg.split(','))
	ov = pipeline.OverlappedAperture(ctx, batch, work, n_chunks=ch, n_mask_streams=ms, mask_priority=bool(pr))
	print(cfg, round(timeit(ov.step), 3), flush=True)
	ov.close()
