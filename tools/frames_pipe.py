#!/usr/bin/env python3
"""The pipelined batched entry alone (for rocprofv3 --kernel-trace): NB batches of N targets on an FR x FR x T region, IN_FLIGHT jobs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photometry_amd import pipeline, tessphot_frames_pipelined
from photometry_amd.device import Context
from benchlib.legs.frames import synthetic_region

N, FR, T = int(os.environ.get('N', 2500)), int(os.environ.get('FR', 512)), int(os.environ.get('T', 1300))
NB, FL = int(os.environ.get('BATCHES', 12)), int(os.environ.get('IN_FLIGHT', 4))
frames, tstamp, quality, cat, targets = synthetic_region(np, N, FR, T, 8)
if not os.environ.get('NOBIND'):
	from photometry_amd.device import bind_host_to_device
	bind_host_to_device(0)
ctx = Context(0)
stack = pipeline.FrameStack(ctx, frames, 0, 44)
del frames
ctx.sync()
rng = np.random.default_rng(9)
batches = [{k: np.asarray(v)[rng.permutation(N)] for k, v in targets.items()} for _ in range(NB)]
for rep in range(int(os.environ.get('REPS', 3))):
	t0 = time.perf_counter()
	ok = 0
	for res in tessphot_frames_pipelined(ctx, stack, iter(batches), cat, tstamp, quality, in_flight=FL):
		ok += int(np.sum((res.status == 1) | (res.status == 3)))
		res = None
	dt = time.perf_counter() - t0
	print(f'rep {rep}: {NB} batches of {N}, {FL} in flight: {dt * 1e3:.1f} ms = {NB * N / dt:.0f} targets/s; OK/WARNING {ok}', flush=True)
ctx.close()
