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
