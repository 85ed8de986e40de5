#!/usr/bin/env python3
"""Diagnostic: K2P2 mask parity (device against oracle) over many seeds and scene kinds -- a wider net than the committed test
seeds.  Prints the statistics of tests/k2p2_common.compare per (kind, seed); any non-razor mismatch raises.
  SEEDS=100..140 python tools/fuzz_parity.py            (oracle in a process pool)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from multiprocessing import Pool

KINDS = tuple(os.environ.get('KINDS', 'faint15,small11,crowded,bright,tiny').split(','))   # also: wide, huge (tests/k2p2_common.py)

def oracle_job(job):
	from k2p2_common import make_cases, oracle_batch
	kind, seed = job
	s, S = make_cases(kind, seed)
	return job, oracle_batch(s, S)

if __name__ == '__main__':
	lo, hi = [int(x) for x in os.environ.get('SEEDS', '100..116').split('..')]
	jobs = [(k, sd) for sd in range(lo, hi) for k in KINDS]
	t0 = time.time()
	with Pool(int(os.environ.get('PROCS', '14'))) as pool:
		refs = dict(pool.map(oracle_job, jobs, chunksize=1))
	print(f'oracle: {len(jobs)} scenes in {time.time() - t0:.1f} s', flush=True)
	if os.environ.get('TP_LAB_LIB'):
		from photometry_amd import _lib
		_lib.LIB_PATH = os.environ['TP_LAB_LIB']
	import test_gpu_k2p2 as tg
	from k2p2_common import make_cases, compare
	from photometry_amd.device import Context
	ctx = Context(0)
	tot = {'targets': 0, 'n_exact': 0, 'n_razor': 0}
	worst = 0.0
	for job in jobs:
		s, S = make_cases(*job)
		got = tg.run_device(ctx, s, S)
		try:
			st = compare(s, S, got, refs[job])
		except AssertionError as e:
			print('MISMATCH in', job, e, flush=True)
			continue
		tot['targets'] += s.n_targets; tot['n_exact'] += st['n_exact']; tot['n_razor'] += st['n_razor']
		worst = max(worst, st['max_dcut'])
		if st['n_razor']:
			print('razor-edge target(s) in', job, st, flush=True)
	print('total', tot, 'largest |dCUT|', worst, flush=True)
