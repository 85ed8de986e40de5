#!/usr/bin/env python3
"""Diagnostic: where the time of the batched drop-in path goes -- tessphot_frames on a synthetic CCD region (N targets on an
FR x FR x T frame stack resident in HBM): device passes against host-side bookkeeping."""
import os, sys, time, cProfile, pstats, io
import numpy as np
if os.environ.get('NODE') is not None:   # experiment: run on the CPUs of one NUMA node (2 x 64 cores, SMT siblings at +128)
	node = int(os.environ['NODE'])
	os.sched_setaffinity(0, set(range(64 * node, 64 * node + 64)) | set(range(128 + 64 * node, 128 + 64 * node + 64)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photometry_amd import pipeline, tessphot_frames
from photometry_amd.device import Context

N, FR, T = int(os.environ.get('N', 5000)), int(os.environ.get('FR', 1024)), int(os.environ.get('T', 100))
rng = np.random.default_rng(2)
row0, col0 = 0, 44
img = np.zeros((FR, FR))
rows = rng.uniform(12, FR - 12, N); cols = rng.uniform(12, FR - 12, N); tmag = rng.uniform(8.5, 14.0, N)
yy, xx = np.mgrid[-4:5, -4:5]
for r, c, m in zip(rows, cols, tmag):
	ri, ci = int(round(r)), int(round(c))
	img[ri - 4:ri + 5, ci - 4:ci + 5] += 10**(-0.4 * (m - 20.451)) * np.exp(-0.5 * ((yy + ri - r)**2 + (xx + ci - c)**2) / 0.9**2) / (2 * np.pi * 0.81)
frames = {}
cube = (img[None] * (1 + 1e-3 * rng.normal(size=T))[:, None, None]).astype('float32')
noise = np.sqrt(np.abs(cube) + 200.0).astype('float32')
frames['images'] = (cube + 30.0 + rng.standard_normal(cube.shape).astype('float32') * noise).astype('float32')
frames['images_err'] = noise
frames['backgrounds'] = np.full(cube.shape, 100.0, dtype='float32')
tstamp = 1500.0 + np.arange(T) * 1800.0 / 86400.0
quality = np.zeros(T, dtype='int32')
cat = {'starid': np.arange(N, dtype='int64') + 1, 'tmag': tmag.astype('float32'), 'row': (rows + row0).astype('float32'), 'column': (cols + col0).astype('float32')}
targets = {'starid': cat['starid'].copy(), 'tmag': tmag, 'row': rows + row0, 'column': cols + col0}
if os.environ.get('NODE') is None and not os.environ.get('NOBIND'):
	from photometry_amd.device import bind_host_to_device
	print('bound to NUMA node', bind_host_to_device(0))
ctx = Context(0)
stack = pipeline.FrameStack(ctx, frames, row0, col0)
ctx.sync()
tessphot_frames(ctx, stack, targets, cat, tstamp, quality) # warm up (the pools of the context are filled)
out = None
for rep in range(2):
	out = None
	t0 = time.perf_counter()
	pr = cProfile.Profile()
	pr.enable()
	out = tessphot_frames(ctx, stack, targets, cat, tstamp, quality)
	pr.disable()
	dt = time.perf_counter() - t0
	ok = int(np.sum((out.status == 1) | (out.status == 3)))
	print(f'tessphot_frames: {N} targets, {T} cadences, {FR}^2 frames: {dt:.3f} s = {N / dt:.0f} targets/s; OK/WARNING {ok}', flush=True)
NB = int(os.environ.get('BATCHES', 0))
if NB:
	# consecutive batches of N targets each (other positions on the same region), two on the device at a time
	from photometry_amd import tessphot_frames_pipelined
	batches = []
	for b in range(NB):
		sel = rng.permutation(N)   # the same stars in another order: the same work per batch
		batches.append({k: np.asarray(v)[sel] for k, v in targets.items()})
	for fl in [int(x) for x in os.environ.get('INFLIGHT', '1,2,3').split(',')]:
		for rep in range(2):
			prp = cProfile.Profile()
			if rep == 1 and os.environ.get('PROFILE_PIPE'):
				prp.enable()
			t0 = time.perf_counter()
			okc = 0
			for res in tessphot_frames_pipelined(ctx, stack, iter(batches), cat, tstamp, quality, in_flight=fl):
				okc += int(np.sum((res.status == 1) | (res.status == 3)))   # the consumer takes what it needs and lets the batch go
				res = None
			dt = time.perf_counter() - t0
			prp.disable()
		if os.environ.get('PROFILE_PIPE') and fl == 3:
			sp = io.StringIO()
			pstats.Stats(prp, stream=sp).sort_stats('tottime').print_stats(22)
			print(sp.getvalue()[:5000])
		print(f'pipelined, {NB} batches of {N}, {fl} in flight: {dt * 1e3:.1f} ms = {NB * N / dt:.0f} targets/s; OK/WARNING {okc}', flush=True)
if os.environ.get('KERNELS'):
	# the device side of one call: per-kernel totals from the library's HIP events (every stream the call used)
	allc = [ctx] + ctx.side_contexts(2)
	for c in allc:
		c.profile(True); c.profile_reset()
	t0 = time.perf_counter()
	out = tessphot_frames(ctx, stack, targets, cat, tstamp, quality)
	dt = time.perf_counter() - t0
	print(f'with the event profile on: {dt * 1e3:.2f} ms')
	for ci, c in enumerate(allc):
		for name, (cnt, ms) in sorted(c.profile_report().items(), key=lambda kv: -kv[1][1]):
			print(f'  stream {ci}: {name:34s} {cnt:4d} launches {ms:8.3f} ms')
		c.profile(False)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6500])
