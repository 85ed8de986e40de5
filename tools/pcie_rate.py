#!/usr/bin/env python3
"""Diagnostic: end-to-end rate when the cubes start in HOST memory (upload + hot path + light-curve download)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photometry_amd import simulate, engine, pipeline
from photometry_amd.device import Context, DeviceCube

ctx = Context(0)
Nt = int(os.environ.get('NT', 1000))
T, H, W = 1300, 15, 15
scene = simulate.make_scene(Nt, T, H, W, seed=1000)
scene.aperture = None
dev = engine.synth_fill(ctx, scene)
host = {k: v.to_host() for k, v in dev.items()}   # (Nt, H, W, T) float32, pageable numpy memory
for v in dev.values():
	v.free()
nbytes = sum(a.nbytes for a in host.values())
for rep in range(3):
	t0 = time.perf_counter()
	cubes = {k: DeviceCube.from_host(ctx, a) for k, a in host.items()}
	ctx.sync()
	t1 = time.perf_counter()
	batch = pipeline.ApertureBatch(ctx, scene, cubes=cubes)
	work = pipeline.ApertureWork(ctx, batch)
	pipeline.aperture_step(ctx, batch, work)
	ctx.sync()
	t2 = time.perf_counter()
	lc = work.lc.to_host()
	t3 = time.perf_counter()
	print(f'rep {rep}: upload {nbytes/1e9:.2f} GB in {t1-t0:.3f} s = {nbytes/1e9/(t1-t0):.1f} GB/s; step(+alloc) {1e3*(t2-t1):.1f} ms; '
		f'download {1e3*(t3-t2):.1f} ms; end-to-end {Nt/(t3-t0):.0f} targets/s', flush=True)
	for v in cubes.values():
		v.free()
