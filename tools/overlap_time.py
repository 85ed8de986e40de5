#!/usr/bin/env python3
"""Diagnostic: does the vector-ALU-bound B* overlap with the bandwidth-bound fused kernel on two streams?
(a) across steps: step s+1's B* under step s's fused kernel (double-buffered background series);
(b) inside one step: the batch in CH chunks of targets, B*(chunk c+1) under fused(chunk c)."""
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

def chunked(n, ch):
	per = (Nt + ch - 1) // ch
	parts = [(a, min(per, Nt - a)) for a in range(0, Nt, per)]
	bs = [batch.chunk(a, c) for a, c in parts]
	ws = [works[0].chunk(a, c) for a, c in parts]
	evs = [ctx.event() for _ in parts]
	ev_done = ctx2.event()
	for s in range(n):
		if s:
			ctx.wait_event(ev_done) # the step before has read its background series
		for i in range(len(parts)):
			engine.background_stamp(ctx, bs[i].images, out=ws[i].bkg_raw)
			engine.smooth_time(ctx, ws[i].bkg_raw, batch.n_cad, batch.time_smooth, out=ws[i].bkg)
			ctx.record(evs[i])
			ctx2.wait_event(evs[i])
			engine.aperture_photometry(ctx2, bs[i], ws[i], subtract=ws[i].bkg, backgrounds=ws[i].bkg)
		ctx2.record(ev_done)
	ctx.sync(); ctx2.sync()

def timeit(fn, *a):
	fn(3, *a)
	t0 = time.perf_counter()
	fn(10, *a)
	return round((time.perf_counter() - t0) / 10 * 1e3, 3)

for rep in range(2):
	print('sequential', timeit(sequential), ' across steps', timeit(pipelined), ' chunks of one step:',
		{ch: timeit(chunked, ch) for ch in (2, 3, 4, 6, 8)}, flush=True)
