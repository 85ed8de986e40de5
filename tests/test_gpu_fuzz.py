# -*- coding: utf-8 -*-
"""
A wider net than the committed seeds of tests/test_gpu_k2p2.py: the K2P2 mask builder (A2-A5b, A7) of the device against the
oracle on 40 further scenes of the five kinds (faint 15x15, small 11x11, crowded 21x17, bright with bleed columns, tiny 6x7;
degenerate sum images included) -- every status, mask, flag, contamination and in-mask star list equal, the threshold within
2e-6, and NO target needing the razor-edge escape of ``k2p2_common.compare``.  The oracle runs in a process pool (the GPU box
gives 16 cores); ``tools/fuzz_parity.py`` is the same loop for arbitrary seed ranges.
Reference: k2p2v2.py:388-623, photometry.py:93-131, 220-254.
"""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KINDS = ('faint15', 'small11', 'crowded', 'bright', 'tiny')
SEEDS = range(300, 308)


def _oracle_job(job):
	from k2p2_common import make_cases, oracle_batch
	s, S = make_cases(*job)
	return job, oracle_batch(s, S)


def test_k2p2_fuzz_forty_scenes():
	from multiprocessing import get_context
	import test_gpu_k2p2 as tg
	from k2p2_common import make_cases, compare
	from photometry_amd.device import Context
	jobs = [(k, sd) for sd in SEEDS for k in KINDS]
	assert len(jobs) >= 40
	nproc = max(1, min(14, len(os.sched_getaffinity(0)) - 1))
	# the oracle workers are forked BEFORE this process opens the GPU
	with get_context('fork').Pool(nproc) as pool:
		refs = dict(pool.map(_oracle_job, jobs, chunksize=1))
	ctx = Context(0)
	tot = {'targets': 0, 'n_exact': 0, 'n_error_agree': 0, 'n_razor': 0}
	worst = 0.0
	for job in jobs:
		s, S = make_cases(*job)
		got = tg.run_device(ctx, s, S)
		st = compare(s, S, got, refs[job])
		tot['targets'] += s.n_targets
		tot['n_exact'] += st['n_exact']
		tot['n_razor'] += st['n_razor']
		tot['n_error_agree'] += st['n_error_agree']
		worst = max(worst, st['max_dcut'])
	ctx.close()
	print('K2P2 fuzz:', tot, 'largest |dCUT|', worst)
	assert tot['targets'] >= 1000 and tot['n_razor'] == 0 and tot['n_exact'] + tot['n_error_agree'] == tot['targets'] and tot['n_exact'] > 0


def test_frames_engine_fuzz_native_equals_python_rounds():
	"""The native job engine of the batched drop-in entry against the Python rounds on eight further random regions (crowded
	fields with bright stars near the frame limits: several resize rounds, size groups, quick breaks, minimum apertures, stamps
	clipped by the region): every target equal in every field a caller can see."""
	from photometry_amd import pipeline
	from photometry_amd.device import Context
	from test_gpu_resize import _compare_frames_results
	ctx = Context(0)
	tot = {'targets': 0, 'resized': 0, 'messages': 0, 'errors': 0, 'passes': 0}
	import os
	lo, hi = [int(x) for x in os.environ.get('TESSPHOT_FUZZ_SEEDS', '201..209').split('..')]   # (a wider net by hand: TESSPHOT_FUZZ_SEEDS=300..340)
	for seed in range(lo, hi):
		rng = np.random.default_rng(seed)
		N, FR, T = int(rng.integers(150, 500)), int(rng.integers(96, 200)), 24
		rows, cols = rng.uniform(2, FR - 2, N), rng.uniform(2, FR - 2, N)      # (stars right at the limits: clipped default stamps)
		tmag = np.where(rng.random(N) < 0.08, rng.uniform(4.5, 7.5, N), rng.uniform(8.0, 14.5, N))
		img = np.zeros((FR + 32, FR + 32))
		yy, xx = np.mgrid[-8:9, -8:9]
		for r, c, m in zip(rows, cols, tmag):
			ri, ci = int(round(r)) + 16, int(round(c)) + 16
			flux = 10**(-0.4 * (m - 20.451))
			img[ri - 8:ri + 9, ci - 8:ci + 9] += flux * np.exp(-0.5 * ((yy + ri - 16 - r)**2 + (xx + ci - 16 - c)**2) / 0.81) / (2 * np.pi * 0.81)
			if m < 7.5:   # a bleed trail along the column, two pixels wide
				half = int(rng.integers(8, 40))
				img[max(ri - half, 0):ri + half + 1, ci:ci + 2] += 0.02 * flux
		img = img[16:-16, 16:-16]
		base = (img[None] * (1 + 1e-3 * rng.normal(size=T))[:, None, None]).astype('float32')
		noise = np.sqrt(np.abs(base) + 200.0).astype('float32')
		images = (base + 30.0 + rng.standard_normal(base.shape).astype('float32') * noise).astype('float32')
		images[rng.random(images.shape) < 3e-4] = np.nan
		fr = {'images': images, 'images_err': noise, 'backgrounds': np.full((T, FR, FR), 100.0, dtype='float32')}
		tstamp = 1500.0 + np.arange(T) * 1800.0 / 86400.0
		q = np.zeros(T, dtype='int32')
		q[int(rng.integers(0, T))] = 32
		row0, col0 = int(rng.integers(0, 300)), 44 + int(rng.integers(0, 300))
		cat = {'starid': np.arange(N, dtype='int64') + 1, 'tmag': tmag.astype('float32'), 'row': (rows + row0).astype('float32'), 'column': (cols + col0).astype('float32')}
		sel = rng.permutation(N)[:int(0.8 * N)]
		tg = {'starid': cat['starid'][sel].copy(), 'tmag': tmag[sel], 'row': rows[sel] + row0, 'column': cols[sel] + col0}
		stack = pipeline.FrameStack(ctx, fr, row0, col0)
		py = pipeline.aperture_frames(ctx, stack, tg, cat, tstamp, q, engine='python')
		nat = pipeline.aperture_frames(ctx, stack, tg, cat, tstamp, q, engine='native')
		_compare_frames_results(py, nat)
		tot['targets'] += nat.n
		tot['resized'] += int((nat.stamp_resizes > 0).sum())
		tot['messages'] += len(nat.errors)
		tot['errors'] += int((nat.status == 2).sum())
		tot['passes'] += len(nat.groups)
		del py, nat, stack
	ctx.close()
	print('frames engine fuzz:', tot)
	assert tot['targets'] > 1500 and tot['resized'] > 50 and tot['messages'] > 20 and tot['passes'] > 40


def test_frames_engine_pool_under_four_jobs_in_flight(monkeypatch):
	"""The engine's shared stream pool under load: 16 batches of random sizes (1 .. 3 000 targets of a region with bright stars and
	bleed trails: several resize rounds and size groups per batch), four jobs in flight, every batch equal to a call of its own
	(tools/lab/frames_stress.py; by hand with NB=30 T=200: 22 273 targets, 873 device passes per run)."""
	import os, runpy
	monkeypatch.setenv('BRIGHT', '1')
	monkeypatch.setenv('NB', '16')
	monkeypatch.setenv('T', '48')
	monkeypatch.setenv('REPS', '2')
	runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'lab', 'frames_stress.py'), run_name='__main__')
