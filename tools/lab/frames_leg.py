#!/usr/bin/env python3
"""lab: the frames-to-results leg of bench.py alone (same synthetic region, same calls), REPS times."""
import os, sys, types
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from photometry_amd import pipeline
from photometry_amd.device import Context, bind_host_to_device
from benchlib.legs.frames import leg_frames
if not os.environ.get('NOBIND'):
	print('numa', bind_host_to_device(0))
ctx = Context(0)
extra = []
for _ in range(int(os.environ.get('EXTRA_STREAMS', 0))):     # experiment: more streams in the process (hardware queues are shared)
	c = Context(0)
	a = c.zeros((1024,), 'float32'); c.sync()
	extra.append((c, a))
if os.environ.get('HOLD_GB'):                                 # experiment: device memory held by the rest of the process
	hold = ctx.empty((int(float(os.environ['HOLD_GB']) * 2**28),), 'float32'); hold.fill(0); ctx.sync()
args = types.SimpleNamespace(frames_targets=int(os.environ.get('N', 2500)), seed=int(os.environ.get('SEED', 0)))
for rep in range(int(os.environ.get('REPS', 2))):
	r = leg_frames(ctx, args, 1300, np, pipeline, FR=int(os.environ.get('FR', 512)), NB=int(os.environ.get('NB', 12)), runs=int(os.environ.get('RUNS', 14)))
	print('single %.0f targets/s' % r['targets_per_s'], ['%.2f' % (x * 1e3) for x in r['seconds_all_calls']], 'pipelined %.0f' % r['pipelined']['targets_per_s'],
		['%.1f' % (x * 1e3) for x in r['pipelined']['seconds_all_runs']], flush=True)
