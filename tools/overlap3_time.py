#!/usr/bin/env python3
"""Diagnostic: consecutive configs[2] steps with the LAST partial round of the mask + extraction launch (the targets beyond a
whole number of rounds of 12 per CU) on a second stream, so that the next step's background pass starts under it."""
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
works = [pipeline.ApertureWork(ctx, batch) for _ in range(2)]
other = Context(ctx.device, high_priority=bool(int(os.environ.get('PRIO', 0))))
slots = int(os.environ.get('SLOTS', 3072))
cut = (Nt // slots) * slots
if cut == Nt:
	cut = Nt - slots // 4
print('targets', Nt, 'first launch', cut, 'tail', Nt - cut)
ba, bb = batch.chunk(0, cut), batch.chunk(cut, Nt - cut)
wa = [w.chunk(0, cut) for w in works]
wb = [w.chunk(cut, Nt - cut) for w in works]
ev_sum = [ctx.event() for _ in range(2)]
ev_tail = [other.event() for _ in range(2)]
used = [False, False]

def step_plain(i):
	pipeline.aperture_step(ctx, batch, works[i % 2])

def step_tail(i):
	b = i % 2
	w = works[b]
	if used[b]:
		ctx.wait_event(ev_tail[b])      # the tail of the step that used this buffer two steps ago
	engine.background_sumimage(ctx, batch.images, batch.quality, batch.time_smooth, bkg_raw=w.bkg_raw, bkg=w.bkg, sumimage=w.sumimage)
	ctx.record(ev_sum[b])
	engine.aperture_photometry(ctx, ba, wa[b], subtract=wa[b].bkg, backgrounds=wa[b].bkg, sumimage_given=True)
	other.wait_event(ev_sum[b])
	engine.aperture_photometry(other, bb, wb[b], subtract=wb[b].bkg, backgrounds=wb[b].bkg, sumimage_given=True)
	other.record(ev_tail[b])
	used[b] = True

def timeit(name, fn, n=20):
	for i in range(4):
		fn(i)
	ctx.sync(); other.sync()
	t0 = time.perf_counter()
	for i in range(n):
		fn(i)
	ctx.sync(); other.sync()
	print(name, 'ms/step', round((time.perf_counter() - t0) / n * 1e3, 3), flush=True)

timeit('plain', step_plain)
timeit('tail on a second stream', step_tail)
timeit('plain', step_plain)
