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


def _block_worker(rank, world, port, outfile):
	"""The packed per-step output block of BASELINE configs[4] (aperture + PSF): every rank packs its shard into ONE byte
	block (comm.packed_block_layout), the blocks are gathered as single messages, rank 0 reassembles global arrays."""
	sys.path.insert(0, ROOT)
	import torch
	import torch.distributed as dist
	from photometry_amd import comm as tpcomm
	os.environ['MASTER_ADDR'] = '127.0.0.1'
	os.environ['MASTER_PORT'] = str(port)
	dist.init_process_group('gloo', rank=rank, world_size=world)
	n_total, T, H, W = 13, 7, 5, 4
	rng = np.random.default_rng(321) # same on every rank: the "global" result
	full = {'lc': rng.normal(size=(5, n_total, T)), 'contamination': rng.random(n_total), 'status': rng.integers(1, 6, n_total).astype('int32'),
		'flags': rng.integers(0, 8, n_total).astype('int32'), 'mask': rng.integers(0, 2, (n_total, H, W)).astype('uint8'),
		'psf_flux': rng.normal(size=(n_total, T)), 'psf_contamination': rng.random(n_total), 'psf_status': rng.integers(1, 4, n_total).astype('int32')}
	sizes = tpcomm.shard_sizes(n_total, world)
	cap = max(sizes)
	layout, nbytes = tpcomm.packed_block_layout(cap, T, H, W, psf=True)
	a, b = tpcomm.shard_range(n_total, world, rank)
	block = np.zeros(nbytes, dtype='uint8')
	views = tpcomm.unpack_block(block, layout)
	for name, arr in full.items():   # what this rank's pipeline would have written into its block
		if name == 'lc':
			views[name][:, :b-a] = arr[:, a:b]
		else:
			views[name][:b-a] = arr[a:b]
	send = torch.from_numpy(block)
	recv = [torch.zeros_like(send) for _ in range(world)] if rank == 0 else None
	dist.gather(send, recv, dst=0)
	if rank == 0:
		got = tpcomm.assemble_blocks([r.numpy() for r in recv], layout, sizes)
		ok = set(got) == set(full) and all(np.array_equal(got[k], full[k]) and got[k].dtype == full[k].dtype for k in full)
		# every field starts on a 256-byte boundary and the fields do not overlap
		offs = sorted((o, int(np.prod(sh)) * np.dtype(dt).itemsize) for o, sh, dt in layout.values())
		ok = ok and all(o % 256 == 0 for o, _ in offs) and all(offs[i][0] + offs[i][1] <= offs[i+1][0] for i in range(len(offs) - 1))
		np.save(outfile, np.array([ok]))
	dist.barrier()
	dist.destroy_process_group()


def test_packed_block_with_psf_outputs_two_ranks(tmp_path):
	torch = pytest.importorskip('torch')
	import torch.multiprocessing as mp
	out = str(tmp_path / 'ok2.npy')
	port = 31500 + (os.getpid() % 2000)
	mp.spawn(_block_worker, args=(2, port, out), nprocs=2, join=True)
	assert bool(np.load(out)[0])


def test_packed_block_layout_without_psf_is_the_aperture_block():
	from photometry_amd import comm as tpcomm
	l0, n0 = tpcomm.packed_block_layout(10, 20, 3, 3, psf=False)
	l1, n1 = tpcomm.packed_block_layout(10, 20, 3, 3, psf=True)
	assert list(l0) == ['lc', 'contamination', 'status', 'flags', 'mask'] and n1 > n0
	assert all(l1[k] == l0[k] for k in l0)   # the PSF fields are appended: the aperture fields keep their place


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
