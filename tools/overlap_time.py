#!/usr/bin/env python3
"""Diagnostic: does the VALU-bound B* of step s+1 overlap with the HBM-bound fused kernel of step s on two streams?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photometry_amd import simulate, engine, pipeline
from photometry_amd.device import Context

Nt, T = int(os.environ.get('NT', 10000)), 1300
ctx = Context(0)
ctx2 = Context(0)
scene = simulate.make_scene(Nt, T, 15, 15, seed=1000)
scene.aperture = None
cubes = engine.synth_fill(ctx, scene, images=False, images_err=True, backgrounds=False, raw=True)
batch = pipeline.ApertureBatch(ctx, scene, cubes={'raw': cubes['raw'], 'raw_err': cubes['images_err']})
works = [pipeline.ApertureWork(ctx, batch, packed=True) for _ in range(2)]
ctx.sync()

def sequential(n):
	for s in range(n):
		pipeline.aperture_step(ctx, batch, works[s % 2])
	ctx.sync()

def pipelined(n):
	ev_bkg = [ctx.event() for _ in range(2)]
	ev_used = [ctx2.event() for _ in range(2)]
	for s in range(n):
		b = s % 2
		w = works[b]
		if s >= 2:
			ctx.wait_event(ev_used[b])
		engine.background_stamp(ctx, batch.images, out=w.bkg_raw)
		engine.smooth_time(ctx, w.bkg_raw, batch.n_cad, batch.time_smooth, out=w.bkg)
		ctx.record(ev_bkg[b])
		ctx2.wait_event(ev_bkg[b])
		engine.aperture_photometry(ctx2, batch, w, subtract=w.bkg, backgrounds=w.bkg)
		ctx2.record(ev_used[b])
	ctx.sync(); ctx2.sync()

for name, fn in (('sequential', sequential), ('pipelined', pipelined), ('sequential', sequential), ('pipelined', pipelined)):
	fn(3)
	t0 = time.perf_counter()
	fn(10)
	print(name, 'ms/step', round((time.perf_counter() - t0) / 10 * 1e3, 3), flush=True)
