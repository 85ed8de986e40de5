#!/usr/bin/env python3
"""Diagnostic: a device pass over a SMALL group of targets (the later rounds of the stamp-resize loop): the fused launch (one
wavefront per target) against the three stand-alone kernels, per stamp size and group size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photometry_amd import simulate, engine, pipeline
from photometry_amd.device import Context

ctx = Context(0)
T = 1300
for (H, W) in ((15, 15), (15, 25), (25, 25), (35, 35)):
	for n in (8, 64, 512):
		scene = simulate.make_scene(n, T, H, W, seed=5)
		scene.aperture = None
		cubes = engine.synth_fill(ctx, scene)
		batch = pipeline.ApertureBatch(ctx, scene, cubes=cubes)
		work = pipeline.ApertureWork(ctx, batch)
		row = []
		for fused in (True, False):
			pipeline.aperture_step(ctx, batch, work, fused=fused); ctx.sync()
			ctx.profile(True); ctx.profile_reset()
			t0 = time.perf_counter()
			for _ in range(5):
				pipeline.aperture_step(ctx, batch, work, fused=fused)
			ctx.sync()
			dt = (time.perf_counter() - t0) / 5
			rep = {k: round(v[1] / 5, 3) for k, v in ctx.profile_report().items()}
			ctx.profile(False)
			row.append((round(dt * 1e3, 3), rep))
		print(H, W, n, 'fused', row[0], 'three kernels', row[1], flush=True)
