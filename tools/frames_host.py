#!/usr/bin/env python3
"""Where the host thread of the pipelined batched entry spends its time: FramesEngine.submit, tp_frames_wait inside collect, the
assembly of the result after it (IN_FLIGHT jobs, BATCHES batches of N targets)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photometry_amd import pipeline, tessphot_frames_pipelined
from photometry_amd.device import Context, bind_host_to_device
from benchlib.legs.frames import synthetic_region

N, FR, T = int(os.environ.get('N', 2500)), int(os.environ.get('FR', 512)), int(os.environ.get('T', 1300))
NB, FL = int(os.environ.get('BATCHES', 12)), int(os.environ.get('IN_FLIGHT', 4))
frames, tstamp, quality, cat, targets = synthetic_region(np, N, FR, T, 8)
bind_host_to_device(0)
ctx = Context(0)
stack = pipeline.FrameStack(ctx, frames, 0, 44)
del frames
rng = np.random.default_rng(9)
batches = [{k: np.asarray(v)[rng.permutation(N)] for k, v in targets.items()} for _ in range(NB)]
acc = {'submit': 0.0, 'wait': 0.0, 'collect': 0.0}
_submit, _collect = pipeline.FramesEngine.submit, pipeline.FramesJob.collect
def submit(self, *a, **k):
	t = time.perf_counter(); r = _submit(self, *a, **k); acc['submit'] += time.perf_counter() - t; return r
def collect(self):
	lib = self.engine.lib
	t = time.perf_counter(); lib.tp_frames_wait(self.handle); t1 = time.perf_counter(); r = _collect(self)
	acc['wait'] += t1 - t; acc['collect'] += time.perf_counter() - t1; return r
pipeline.FramesEngine.submit, pipeline.FramesJob.collect = submit, collect
for rep in range(int(os.environ.get('REPS', 6))):
	for k in acc: acc[k] = 0.0
	t0 = time.perf_counter()
	for res in tessphot_frames_pipelined(ctx, stack, iter(batches), cat, tstamp, quality, in_flight=FL):
		ok = int(np.sum((res.status == 1) | (res.status == 3)))
		res = None
	dt = time.perf_counter() - t0
	print(f'rep {rep}: {dt * 1e3:.1f} ms = {NB * N / dt:.0f} targets/s; per batch: total {dt / NB * 1e3:.2f} ms, submit {acc["submit"] / NB * 1e3:.2f}, '
		f'waiting for the job {acc["wait"] / NB * 1e3:.2f}, result assembly {acc["collect"] / NB * 1e3:.2f}', flush=True)
ctx.close()
