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


class _NumpyWorker(object):
	"""A rank's share of a made-up batch, computed with numpy from the GLOBAL target index, so that any division of the targets
	over ranks must reproduce the single-rank result: light curves, masks, statuses, and a ragged catalogue whose in-mask flags
	give the skip lists.  Follows the ``sharded.ShardWorker`` protocol with a host block (``ctx`` is None)."""
	ctx = None
	nbuf = 2
	T, H, W = 3, 2, 2

	def __init__(self, n_total, world, rank):
		from photometry_amd import comm as tpcomm
		self.a, self.b = tpcomm.shard_range(n_total, world, rank)
		self.n_local = self.b - self.a
		self.capacity = max(tpcomm.shard_sizes(n_total, world))
		g = self.global_fields(n_total)
		self.cat_offsets_local = g['cat_offsets'][self.a:self.b + 1] - g['cat_offsets'][self.a]
		self.n_cat_local = int(self.cat_offsets_local[-1])
		cat_cap = int(max(g['cat_offsets'][tpcomm.shard_range(n_total, world, r)[1]] - g['cat_offsets'][tpcomm.shard_range(n_total, world, r)[0]] for r in range(world)))
		self.layout, self.block_nbytes = tpcomm.packed_block_layout(self.capacity, self.T, self.H, self.W, psf=True, n_cat=max(cat_cap, 1))
		self.blocks = [np.zeros(self.block_nbytes, dtype='uint8') for _ in range(self.nbuf)]
		self.g = g
		self.n_steps = 0

	@classmethod
	def global_fields(cls, n):
		i = np.arange(n)
		ncat = 1 + (i % 3)                                              # 1..3 catalogue rows per target: itself + neighbours
		off = np.concatenate(([0], np.cumsum(ncat)))
		starid = 1000 + i
		tmag = 6.0 + ((i * 7919) % 1000) / 100.0
		cat_starid = np.concatenate([[starid[j]] + [starid[(j + k) % n] for k in range(1, ncat[j])] for j in range(n)]) if n < 5000 else None
		if cat_starid is None:                                          # vectorised for the large case
			rows = np.repeat(i, ncat)
			k = np.arange(off[-1]) - off[rows]
			cat_starid = starid[(rows + k) % n]
		return {'cat_offsets': off, 'starid': starid, 'tmag': tmag, 'cat_starid': cat_starid, 'ncat': ncat}

	def step(self, b):
		from photometry_amd import comm as tpcomm
		self.n_steps += 1
		v = tpcomm.unpack_block(self.blocks[b], self.layout)
		i = np.arange(self.a, self.b)
		n = self.n_local
		v['lc'][:, :n, :] = (i[None, :, None] * 0.5 + np.arange(5)[:, None, None] * 1e3 + np.arange(self.T)[None, None, :] + self.n_steps * 1e-3)
		# flux, flux_err, flux_background are float32 sums widened on store (photometry.py:172-201), the centroids genuine float64
		v['lc'][:3, :n, :] = v['lc'][:3, :n, :].astype('float32')
		v['lc'][0, :n, 1][i % 7 == 0] = np.nan
		v['contamination'][:n] = (i % 17) / 17.0
		v['status'][:n] = 1 + (i % 3 == 0) * 2                            # OK / WARNING
		v['flags'][:n] = i % 5
		v['mask'][:n] = ((i[:, None, None] + np.arange(self.H)[None, :, None] + np.arange(self.W)[None, None, :]) % 2).astype('uint8')
		v['psf_flux'][:n] = i[:, None] * 2.0 + np.arange(self.T)[None, :]
		v['psf_contamination'][:n] = (i % 11) / 11.0
		v['psf_status'][:n] = 1 + (i % 4 == 0)
		# catalogue flags: the target itself always, its first neighbour when the target index is a multiple of 5
		rows = np.repeat(np.arange(n), np.diff(self.cat_offsets_local))
		k = np.arange(self.n_cat_local) - self.cat_offsets_local[rows]
		v['cat_in_mask'][:self.n_cat_local] = ((k == 0) | ((k == 1) & (i[rows] % 5 == 0))).astype('uint8')

	def block(self, b):
		return self.blocks[b]

	def sync(self):
		pass


def _sharded_run(n_total, world, rank, group=None, steps=3, when='step', calls=1, compact=True):
	from photometry_amd import sharded
	w = _NumpyWorker(n_total, world, rank)
	run = sharded.ShardedRun(w, n_total, rank=rank, world=world, group=group, gather='auto', when=when, compact=compact)
	assert run.send_nbytes == (tpcomm_compact_nbytes(w) if compact else w.block_nbytes)
	for _ in range(calls):                     # run_steps is re-entrant: the state of the double buffer lives on the instance
		run.run_steps(steps, collect=True)
	run.barrier()
	res = run.collect()
	out = None
	if rank == 0:
		g = w.g
		skip = run.skip_lists(res['cat_in_mask'], g['cat_offsets'], g['cat_starid'], g['starid'])
		res['final_status'] = run.replay(res, g['starid'], g['tmag'], skip)
		res['n_skip'] = np.array([sum(len(s) for s in skip)])
		res['mode'] = run.mode
		res['n_gathers'] = np.array([len(run.gather_ms), len(run.final_ms)])
		out = res
	run.close()
	return out


def tpcomm_compact_nbytes(w):
	from photometry_amd import comm as tpcomm
	n = tpcomm.compact_block_layout(w.layout)[1]
	assert n < w.block_nbytes or w.capacity * w.T < 1000   # (tiny blocks: the alignment of one more field can outweigh the halved planes)
	return n


def _sharded_worker(rank, world, port, n_total, outfile, kind='gloo', when='step', calls=1, compact=True):
	sys.path.insert(0, ROOT)
	os.environ['MASTER_ADDR'] = '127.0.0.1'
	os.environ['MASTER_PORT'] = str(port)
	os.environ['RANK'], os.environ['LOCAL_RANK'], os.environ['WORLD_SIZE'] = str(rank), str(rank), str(world)
	from photometry_amd import sharded
	assert sharded.rank_environment() == (rank, rank, world)
	group = sharded.init_host_group(rank, world, kind=kind)
	assert type(group).__name__ == {'gloo': 'TorchGroup', 'socket': 'SocketGroup'}[kind]
	if kind == 'socket':
		assert 'torch' not in sys.modules or os.environ.get('TP_TEST_SPAWNED_BY_TORCH') == '1'
	res = _sharded_run(n_total, world, rank, group=group, when=when, calls=calls, compact=compact)
	if rank == 0:
		mode = res.pop('mode')
		assert mode == 'host', mode
		np.savez(outfile, **res)
	group.barrier()
	group.close()


def _compare_with_single_process(got, n_total, steps_total=3, when='step'):
	ref = _sharded_run(n_total, 1, 0, steps=steps_total)
	assert ref.pop('mode') == 'none (single rank)'
	ng, nr = got.pop('n_gathers'), ref.pop('n_gathers')
	assert list(nr) == [0, 0]
	assert set(got) == set(ref)
	for k in ref:
		np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
		assert got[k].dtype == ref[k].dtype, k
	assert got['lc'].shape == (5, n_total, _NumpyWorker.T) and int(got['n_skip'][0]) > 0
	assert (got['final_status'] == 5).any() and (got['final_status'] != ref['status']).any()   # the replay did something
	return list(ng)


@pytest.mark.parametrize('n_total,world', [(100003, 3), (11, 2)])
def test_sharded_run_uneven_shards_equals_single_process(tmp_path, n_total, world):
	"""The package's sharded entry (photometry_amd.sharded.ShardedRun) on CPU ranks: 100 003 targets over 3 ranks (shards of
	33 335 / 33 335 / 33 333 behind a padded capacity, ragged catalogues of different lengths), three double-buffered steps
	with a gather each, reassembly, skip lists and the master's skip-target replay -- against the same run in ONE process."""
	pytest.importorskip('torch')
	import torch.multiprocessing as mp
	out = str(tmp_path / 'sharded.npz')
	port = 33500 + (os.getpid() % 2000)
	mp.spawn(_sharded_worker, args=(world, port, n_total, out), nprocs=world, join=True)
	assert _compare_with_single_process(dict(np.load(out)), n_total) == [3, 0]


def _spawn_plain(target, args, world):
	"""Ranks as plain child processes (multiprocessing 'spawn': fresh interpreters, as on a GPU node), no torch anywhere."""
	import multiprocessing
	ctx = multiprocessing.get_context('spawn')
	procs = [ctx.Process(target=target, args=(r,) + args) for r in range(world)]
	for p in procs:
		p.start()
	for p in procs:
		p.join(600)
	assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


@pytest.mark.parametrize('when,calls,compact', [('step', 1, True), ('final', 1, True), ('step', 2, True), ('final', 1, False)])
def test_sharded_run_over_the_socket_group_without_torch(tmp_path, when, calls, compact):
	"""The same run with the host group on plain TCP sockets (hostgroup.SocketGroup: rank 0 on an ephemeral port announced in a
	rendezvous file; no PyTorch imported in any rank), gathering every step, once at the end, and with run_steps called twice
	back to back (the second call must wait for the first call's gathers before it overwrites their blocks).  What travels is the
	COMPACT block (flux / flux_err / flux_background as float32: comm.compact_block_layout) unless ``compact`` is off: the results
	are the same bit for bit either way, and equal to the single-process run, which gathers nothing."""
	out = str(tmp_path / 'sock.npz')
	n_total, world = 1003, 3
	os.environ['TESSPHOT_RDZV_ID'] = 'test_%d_%s_%d' % (os.getpid(), when, calls)
	try:
		_spawn_plain(_sharded_worker, (world, 0, n_total, out, 'socket', when, calls, compact), world)
	finally:
		del os.environ['TESSPHOT_RDZV_ID']
	ng = _compare_with_single_process(dict(np.load(out)), n_total, steps_total=3 * calls)
	assert ng == {('step', 1): [3, 0], ('final', 1): [0, 1], ('step', 2): [6, 0]}[(when, calls)]


def test_compact_block_round_trip_and_size():
	"""comm.compact_block / expand_block: the identity on a block whose first three light-curve planes are float32 values (NaN
	included), field for field and padding included; at configs[4]'s shape the block a rank sends shrinks from 783 to 588 MB."""
	from photometry_amd import comm as tpcomm
	lay, nb = tpcomm.packed_block_layout(7, 13, 5, 5, psf=True, n_cat=20, extras=True)
	rng = np.random.default_rng(0)
	blk = np.zeros(nb, 'uint8')
	u = tpcomm.unpack_block(blk, lay)
	u['lc'][:3] = rng.normal(size=(3, 7, 13)).astype('float32')
	u['lc'][0, 0, 0] = np.nan
	u['lc'][1, 2, 3] = -np.inf
	u['lc'][3:] = rng.normal(size=(2, 7, 13))
	for k in u:
		if k != 'lc':
			u[k][...] = rng.integers(0, 200, u[k].shape).astype(u[k].dtype)
	c = tpcomm.compact_block(blk, lay)
	clay, cn, fields = tpcomm.compact_block_layout(lay)
	assert len(c) == cn < nb and all(off % 256 == 0 for off, _, _ in clay.values())
	np.testing.assert_array_equal(tpcomm.expand_block(c, lay), blk)
	lay4, nb4 = tpcomm.packed_block_layout(12500, 1300, 15, 15, psf=True)
	cn4 = tpcomm.compact_block_layout(lay4)[1]
	assert nb4 > 780e6 and cn4 < 590e6


def test_strong_scaling_100k_targets_over_8_cpu_ranks(tmp_path):
	"""BASELINE configs[4]'s division in small: 100 000 targets over 8 ranks (12 500 each, tiny cadence count) through
	sharded.ShardedRun on CPU ranks, final gather, reassembled and replayed == the single-process run."""
	out = str(tmp_path / 'strong.npz')
	n_total, world = 100000, 8
	from photometry_amd import comm as tpcomm
	assert tpcomm.shard_sizes(n_total, world) == [12500] * 8
	os.environ['TESSPHOT_RDZV_ID'] = 'test_strong_%d' % os.getpid()
	try:
		_spawn_plain(_sharded_worker, (world, 0, n_total, out, 'socket', 'final', 1), world)
	finally:
		del os.environ['TESSPHOT_RDZV_ID']
	assert _compare_with_single_process(dict(np.load(out)), n_total) == [0, 1]


def _group_worker(rank, world, outdir):
	sys.path.insert(0, ROOT)
	from photometry_amd import hostgroup
	g = hostgroup.SocketGroup(rank, world, rendezvous_timeout=60)
	assert g.max(rank * 1.5) == (world - 1) * 1.5 and g.min(rank + 2) == 2
	assert g.allgather_int(rank * rank) == [r * r for r in range(world)]
	assert g.broadcast_bytes(b'\x01' * 128 if rank == 0 else b'', src=0) == b'\x01' * 128
	a = np.full((3, 5), rank, dtype='int32')
	got = g.gather_array(a)
	if rank == 0:
		assert [int(x[0, 0]) for x in got] == list(range(world)) and got[1].dtype == a.dtype and got[1].shape == a.shape
		assert not os.path.exists(hostgroup.rendezvous_file())       # removed once every rank is connected
	else:
		assert got is None
	g.barrier()
	g.close()
	open(os.path.join(outdir, 'ok%d' % rank), 'w').close()


def test_socket_group_collectives_and_stale_rendezvous_file(tmp_path):
	"""hostgroup.SocketGroup on 4 ranks; a stale rendezvous file of an 'earlier launch' (a dead port) is in the way: the ranks
	must retry until rank 0 has replaced it."""
	from photometry_amd import hostgroup
	os.environ['TESSPHOT_RDZV_ID'] = 'test_group_%d' % os.getpid()
	try:
		with open(hostgroup.rendezvous_file(), 'w') as fh:
			fh.write('1 999999\n')                                   # port 1: nobody listens there
		_spawn_plain(_group_worker, (4, str(tmp_path)), 4)
	finally:
		del os.environ['TESSPHOT_RDZV_ID']
	assert sorted(os.listdir(tmp_path)) == ['ok0', 'ok1', 'ok2', 'ok3']


def test_socket_group_under_torch_distributed_run(tmp_path):
	"""As the driver starts the N > 1 bench: ``python -m torch.distributed.run --master-port P``.  MASTER_PORT belongs to the
	launcher's own store there; the socket group must come up beside it (rendezvous file named after the launcher's pid), and
	the ranks themselves import no torch."""
	pytest.importorskip('torch')
	import subprocess
	script = tmp_path / 'rank.py'
	script.write_text(
		"import os, sys\nsys.path.insert(0, %r)\nfrom photometry_amd import sharded\n"
		"rank, lr, world = sharded.rank_environment()\ng = sharded.init_host_group(rank, world)\n"
		"assert type(g).__name__ == 'SocketGroup' and g.max(rank) == world - 1 and g.allgather_int(rank) == list(range(world))\n"
		"assert 'torch' not in sys.modules\ng.barrier(); g.close()\nopen(os.path.join(%r, 'ok%%d' %% rank), 'w').close()\n" % (ROOT, str(tmp_path)))
	port = 27500 + (os.getpid() % 2000)
	env = {k: v for k, v in os.environ.items() if k != 'TESSPHOT_RDZV_ID'}
	r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '3', '--master-addr', '127.0.0.1',
		'--master-port', str(port), str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
	assert r.returncode == 0, r.stdout.decode()[-3000:]
	assert sorted(f for f in os.listdir(tmp_path) if f.startswith('ok')) == ['ok0', 'ok1', 'ok2']


def test_spawn_ranks_ends_the_survivors_when_a_rank_fails(tmp_path):
	"""sharded.spawn_ranks: rank 1 exits with an error at once, rank 0 would wait for it for ever: the launcher must end it and
	return the failure."""
	import time
	from photometry_amd import sharded
	script = tmp_path / 'rank.py'
	script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\nassert os.environ['TESSPHOT_RDZV_ID']\ntime.sleep(600)\n")
	t0 = time.time()
	assert sharded.spawn_ranks(str(script), [], 2, grace_s=5.0) == 7
	assert time.time() - t0 < 60


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


def _net_group_worker(rank, world, outdir, port):
	sys.path.insert(0, ROOT)
	os.environ['MASTER_ADDR'] = '127.0.0.1'
	os.environ['TESSPHOT_RDZV_PORT'] = str(port)
	os.environ['TESSPHOT_RDZV_SECRET'] = 'net-test-secret'
	os.environ.pop('TESSPHOT_RDZV_ID', None)
	from photometry_amd import hostgroup
	assert hostgroup.network_rendezvous_port() == port
	if rank == 1:
		# a stranger reaches rank 0's port first and announces an absurd message: rank 0 must drop it and go on accepting
		import socket, struct, time
		for _ in range(200):
			try:
				s = socket.create_connection(('127.0.0.1', port), timeout=1.0)
				break
			except OSError:
				time.sleep(0.05)
		s.sendall(struct.pack('<Q', 1 << 62))
		s.close()
		s = socket.create_connection(('127.0.0.1', port), timeout=1.0)
		s.sendall(struct.pack('<Q', 12) + b'not-the-magic')       # a well-formed stranger too
		s.close()
	g = hostgroup.SocketGroup(rank, world, rendezvous_timeout=60)
	assert g.max(rank * 2.0) == (world - 1) * 2.0 and g.allgather_int(rank + 10) == [r + 10 for r in range(world)]
	a = np.full((2, 3), rank, dtype='float32')
	got = g.gather_array(a, dst=0)
	if rank == 0:
		assert [int(x[0, 0]) for x in got] == list(range(world))
	g.barrier()
	g.close()
	open(os.path.join(outdir, 'ok%d' % rank), 'w').close()


def test_socket_group_network_rendezvous_and_strangers(tmp_path):
	"""The several-node rendezvous of hostgroup.SocketGroup (TESSPHOT_RDZV_PORT: rank 0 listens on a known port, no rendezvous file),
	with the handshake secret, while strangers connect to the port: one announces a message of 2^62 bytes, one sends a well-formed
	message that is not the handshake -- rank 0 drops both (the length is checked before anything is allocated) and the group comes up."""
	import socket
	s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
	_spawn_plain(_net_group_worker, (3, str(tmp_path), port), 3)
	assert sorted(os.listdir(tmp_path)) == ['ok0', 'ok1', 'ok2']


def test_network_rendezvous_port_from_the_launcher_environment(monkeypatch):
	"""One node: no port (the rendezvous file).  A launcher environment that spans nodes (WORLD_SIZE > LOCAL_WORLD_SIZE): MASTER_PORT + 1."""
	from photometry_amd import hostgroup
	for k in ('TESSPHOT_RDZV_PORT', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_PORT'):
		monkeypatch.delenv(k, raising=False)
	assert hostgroup.network_rendezvous_port() is None
	monkeypatch.setenv('WORLD_SIZE', '8'); monkeypatch.setenv('LOCAL_WORLD_SIZE', '8'); monkeypatch.setenv('MASTER_PORT', '29500')
	assert hostgroup.network_rendezvous_port() is None
	monkeypatch.setenv('WORLD_SIZE', '16')
	assert hostgroup.network_rendezvous_port() == 29501
	monkeypatch.setenv('TESSPHOT_RDZV_PORT', '40000')
	assert hostgroup.network_rendezvous_port() == 40000
