#!/usr/bin/env python3
"""Diagnostic: the configs[2] step in target chunks on two streams -- the mask + extraction launch of chunk c (latency-bound)
under the background + sum-image launch of chunk c + 1 (vector-ALU bound)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photometry_amd import simulate, engine, pipeline
from photometry_amd.device import Context

ctx = Context(0)
Nt = int(os.environ.get('NT', 10000))
scene = simulate.make_scene(Nt, 1300, 15, 15, seed=1000)
scene.aperture = None
cubes = engine.synth_fill(ctx, scene, images=False, images_err=True, backgrounds=False, raw=True)
batch = pipeline.ApertureBatch(ctx, scene, cubes={'raw': cubes['raw'], 'raw_err': cubes['images_err']})
work = pipeline.ApertureWork(ctx, batch)

def step_plain():
	pipeline.aperture_step(ctx, batch, work)

def make_chunked(nch, prio):
	other = Context(ctx.device, high_priority=prio)
	bounds = [(Nt * i) // nch for i in range(nch + 1)]
	bch = [batch.chunk(a, b - a) for a, b in zip(bounds[:-1], bounds[1:])]
	wch = [work.chunk(a, b - a) for a, b in zip(bounds[:-1], bounds[1:])]
	ev = [ctx.event() for _ in range(nch)]
	done = other.event()
	def step():
		for c in range(nch):
			engine.background_sumimage(ctx, bch[c].images, bch[c].quality, batch.time_smooth, bkg_raw=wch[c].bkg_raw, bkg=wch[c].bkg, sumimage=wch[c].sumimage)
			ctx.record(ev[c])
			other.wait_event(ev[c])
			engine.aperture_photometry(other, bch[c], wch[c], subtract=wch[c].bkg, backgrounds=wch[c].bkg, sumimage_given=True)
		other.record(done)
		ctx.wait_event(done)
	return step

def timeit(name, fn, n=8):
	for _ in range(2):
		fn()
	ctx.sync()
	t0 = time.perf_counter()
	for _ in range(n):
		fn()
	ctx.sync()
	print(name, 'ms/step', round((time.perf_counter() - t0) / n * 1e3, 3), flush=True)

timeit('plain', step_plain)
for nch in (2, 3, 4, 8):
	for prio in (False, True):
		timeit(f'chunks={nch} prio={prio}', make_chunked(nch, prio))
