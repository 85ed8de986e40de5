#!/usr/bin/env python3
"""lab: stress of the frames engine's shared stream pool -- many batches of random sizes, four jobs in flight, every batch compared
with a call of its own (everything a caller can see, tests/test_gpu_resize.py::_compare_frames_results)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from photometry_amd import pipeline
from photometry_amd.device import Context
from benchlib.legs.frames import synthetic_region
from test_gpu_resize import _compare_frames_results
ctx = Context(0)
N, FR, T = 3000, 512, int(os.environ.get('T', 200))
frames, tstamp, quality, cat, targets = synthetic_region(np, N, FR, T, int(os.environ.get('SEED', 3)))
if os.environ.get('BRIGHT'):   # bright stars with bleed trails: several resize rounds and size groups per batch
	rb = np.random.default_rng(9)
	pick = rb.random(N) < 0.05
	newmag = rb.uniform(5.0, 7.5, N)
	for i in np.flatnonzero(pick):
		r, c = int(round(targets['row'][i])), int(round(targets['column'][i] - 44))
		flux = 10**(-0.4 * (newmag[i] - 20.451))
		half = int(rb.integers(8, 40))
		r0, r1 = max(r - half, 0), min(r + half + 1, FR)
		if 0 <= c < FR - 1:
			frames['images'][:, r0:r1, c:c + 2] += np.float32(0.02 * flux)
			frames['images'][:, max(r - 2, 0):r + 3, max(c - 2, 0):c + 3] += np.float32(0.1 * flux)
	targets['tmag'] = np.where(pick, newmag, targets['tmag'])
	cat['tmag'] = targets['tmag'].astype('float32')
stack = pipeline.FrameStack(ctx, frames, 0, 44)
rng = np.random.default_rng(5)
NB = int(os.environ.get('NB', 40))
batches = []
for _ in range(NB):
	m = int(rng.choice([1, 7, 60, 300, 1200, 2500, 3000]))
	sel = rng.permutation(N)[:m]
	batches.append({k: np.asarray(v)[sel] for k, v in targets.items()})
alone = [pipeline.aperture_frames(ctx, stack, b, cat, tstamp, quality, engine='native') for b in batches]
for rep in range(int(os.environ.get('REPS', 3))):
	piped = list(pipeline.aperture_frames_pipelined(ctx, stack, iter(batches), cat, tstamp, quality, in_flight=4))
	assert len(piped) == NB
	for a, b in zip(alone, piped):
		_compare_frames_results(a, b)
	print('run', rep, 'ok:', NB, 'batches,', sum(int(b.n) for b in piped), 'targets,', sum(len(b.groups) for b in piped), 'device passes', flush=True)
