# -*- coding: utf-8 -*-
"""
world_size-2 test of the multi-GPU host logic on CPU (gloo): target sharding, the gather of the
per-rank light-curve blocks and their reassembly.  On the GPU the transport is tp_comm_gather (RCCL);
here the same blocks travel through torch.distributed.gather.
"""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, outfile):
	sys.path.insert(0, ROOT)
	import torch
	import torch.distributed as dist
	from photometry_amd import comm as tpcomm
	os.environ['MASTER_ADDR'] = '127.0.0.1'
	os.environ['MASTER_PORT'] = str(port)
	dist.init_process_group('gloo', rank=rank, world_size=world)
	n_total, T = 11, 9
	rng = np.random.default_rng(123) # same on every rank: the "global" result
	full = rng.normal(size=(5, n_total, T))
	a, b = tpcomm.shard_range(n_total, world, rank)
	cap = max(tpcomm.shard_sizes(n_total, world))
	mine = np.zeros((5, cap, T))
	mine[:, :b-a] = full[:, a:b] # what this rank's pipeline would have produced for its shard
	send = torch.from_numpy(mine)
	recv = [torch.zeros_like(send) for _ in range(world)] if rank == 0 else None
	dist.gather(send, recv, dst=0)
	if rank == 0:
		got = tpcomm.assemble_gathered([r.numpy() for r in recv], tpcomm.shard_sizes(n_total, world))
		np.save(outfile, np.array([np.array_equal(got, full)]))
	dist.barrier()
	dist.destroy_process_group()


def test_shard_gather_assemble_two_ranks(tmp_path):
	torch = pytest.importorskip('torch')
	import torch.multiprocessing as mp
	out = str(tmp_path / 'ok.npy')
	port = 29500 + (os.getpid() % 2000)
	mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
	assert bool(np.load(out)[0])


def test_bench_spawns_its_own_ranks_and_fails_loudly_without_gpu():
	"""`python bench.py --gpus 2` (no launcher, no WORLD_SIZE) must start two ranks itself; without a GPU every rank must
	fail loudly -- never fall back to one rank or to the CPU."""
	import subprocess
	import ctypes
	from photometry_amd import _lib
	n = ctypes.c_int(0)
	_lib.load().tp_device_count(ctypes.byref(n))
	if n.value > 0:
		pytest.skip("a GPU is visible")
	env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
	r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--targets', '8'],
		env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
	assert r.returncode != 0
	err = r.stderr.decode()
	assert err.count('bench.py needs a GPU') >= 2, err[-2000:]   # both ranks started, both refused to run
	assert '{' not in r.stdout.decode()   # no result line
