#!/usr/bin/env python3
"""lab: cProfile of FramesJob.collect (the host's share of a batch of the native frames engine)."""
import os, sys, time, cProfile, pstats, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from photometry_amd import pipeline
from photometry_amd.device import Context, bind_host_to_device
from benchlib.legs.frames import synthetic_region
frames, tstamp, quality, cat, targets = synthetic_region(np, 2500, 512, 1300, 8)
bind_host_to_device(0)
ctx = Context(0)
stack = pipeline.FrameStack(ctx, frames, 0, 44)
eng = pipeline.FramesEngine.of(ctx)
c = eng.catalog(cat)
for _ in range(3):
	eng.submit(stack, targets, c, tstamp, quality).collect()
pr = cProfile.Profile()
tot = 0.0
for _ in range(5):
	job = eng.submit(stack, targets, c, tstamp, quality)
	eng.lib.tp_frames_wait(job.handle)
	t0 = time.perf_counter()
	pr.enable(); res = job.collect(); pr.disable()
	tot += time.perf_counter() - t0
	print('groups', len(res.groups), 'errors', len(res.errors))
	res = None
print('collect ms', tot / 5 * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(14); print(s.getvalue()[:3000])
