# -*- coding: utf-8 -*-
"""
The sharded run: one process per GPU, the targets of a batch divided over the ranks by index, every rank running the
per-target stages on its share, the per-step output blocks gathered to rank 0 and put back in global target order, the
master-side skip-target bookkeeping replayed on the gathered results.

This is what ``run_tessphot_mpi.py:74-209`` is to the reference (a master handing one pickled task at a time to MPI workers and
saving what comes back, ``taskmanager.py:460-532``) -- re-designed for a node of GPUs: targets are independent (SURVEY.md
section 8e), so the division is static (``comm.shard_range``), there is **no data-path collective** except the gather of each
step's output block, and that gather is issued on a second stream from the other half of a double-buffered block so that it
overlaps the next step's compute.

Pieces (``bench.py`` and ``INTEGRATION.md`` section 3 call these; ``tests/test_distributed_gloo.py`` runs the whole entry on
CPU ranks with uneven shards):

* :func:`spawn_ranks` -- start the ranks as fresh child processes BEFORE anything touches the GPU;
* :func:`rank_environment`, :func:`init_host_group` -- rank / world from the launcher's environment, the gloo group used for
  rendezvous, barriers and (on a box where the ranks share a device) the host fallback of the gather;
* :class:`ShardWorker` -- what a rank does per step (protocol); :class:`DeviceShardWorker` -- the BASELINE workloads on the
  device (configs[2] aperture + background, configs[4] aperture + PSF) on synthetic cubes generated in HBM;
* :class:`ShardedRun` -- the step loop with the double-buffered gather (RCCL: ``tp_comm_gather``; host fallback: gloo),
  reassembly (``comm.assemble_blocks``: every rank sends the same padded capacity, the real sizes trim it) and the replay of
  the skip lists (``comm.replay_skip_targets``).
"""

import os
import socket
import subprocess
import sys
import time
import numpy as np
from . import comm as tpcomm


# --------------------------------------------------------------------------------------------------
# ranks
# --------------------------------------------------------------------------------------------------
def spawn_ranks(script, argv, n_ranks):
	"""
	Start one fresh child process per rank (``script argv...`` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) BEFORE
	anything in this process touches the GPU, relay rank 0's standard output, return the worst exit code.
	"""
	with socket.socket() as s:
		s.bind(('127.0.0.1', 0))
		port = s.getsockname()[1]
	procs = []
	for r in range(int(n_ranks)):
		env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR='127.0.0.1',
			MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
		procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
			stdout=None if r == 0 else subprocess.DEVNULL))
	rc = 0
	for p in procs:
		rc = max(rc, abs(p.wait()))
	return rc


def rank_environment():
	"""``(rank, local_rank, world)`` as torch.distributed.run (or :func:`spawn_ranks`) exports them; ``(0, 0, 1)`` without a launcher."""
	return int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def init_host_group(rank, world):
	"""
	The gloo process group of a multi-rank run (rendezvous, barriers, max over ranks, the unique id of the RCCL communicator).
	torch is plumbing only and is imported only here, BEFORE the HIP library, so that one HIP runtime is shared.  gloo announces
	its connections on stdout (C level): stdout is kept clean for the caller's one result line.  Returns ``(torch, dist)``.
	"""
	import torch
	import torch.distributed as dist
	os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
	sys.stdout.flush()
	saved = os.dup(1)
	os.dup2(2, 1)
	try:
		dist.init_process_group(backend='gloo', rank=rank, world_size=world)
	finally:
		os.dup2(saved, 1)
		os.close(saved)
	return torch, dist


# --------------------------------------------------------------------------------------------------
# what a rank does
# --------------------------------------------------------------------------------------------------
class ShardWorker(object):
	"""
	Protocol of a rank's share of the batch (duck-typed; :class:`DeviceShardWorker` is the device implementation, the CPU tests
	bring a numpy one):

	* ``n_local`` real number of targets of this rank, ``capacity`` the padded number every rank's block is laid out for
	  (``max(comm.shard_sizes)``), ``layout`` / ``block_nbytes`` from ``comm.packed_block_layout(capacity, ...)``; when the
	  layout carries ``cat_in_mask``, ``n_cat_local`` = the rank's real number of catalogue rows;
	* ``nbuf`` output blocks (2: double-buffered); ``step(b)`` queues one pass of the per-target stages writing block ``b``;
	* ``block(b)`` the block: a ``DeviceArray`` (``ctx`` is then the rank's context) or a numpy ``uint8`` array (``ctx`` None);
	* ``sync()`` waits for everything queued.
	"""
	ctx = None
	nbuf = 2


class DeviceShardWorker(ShardWorker):
	"""
	A rank's share of a BASELINE workload on synthetic cubes generated in HBM (``tp_synth_fill``; ``simulate.make_scene`` seeds
	the scene per rank):

	* ``'c2'`` configs[2] aperture + background: raw flux + error cubes resident; a step = ``tp_background_sumimage`` (B*, B2 and
	  the sum image in one pass over the raw cube) + ``tp_aperture_photometry_from_sumimage``;
	* ``'c4'`` configs[4] aperture + PSF: the same followed by the LinPSF fit (``tp_linpsf_prf`` + ``tp_linpsf_fit``) on the same
	  raw cube with the step's background series subtracted on the fly; the LinPSF light curve, contamination and status are
	  part of the block.
	"""

	def __init__(self, ctx, scene, capacity=None, psf=False, nbuf=1, extras=False, cat_capacity=0):
		from . import engine, pipeline
		self.ctx, self.scene, self.psf = ctx, scene, bool(psf)
		self.n_local = scene.n_targets
		self.n_cat_local = int(scene.cat_offsets[-1])
		self.capacity = int(capacity) if capacity is not None else scene.n_targets
		self.nbuf = int(nbuf)
		# resident inputs: raw flux + error cubes (2 x 11.8 GB at 10 000 x 1300 x 15 x 15); `extras` adds the
		# background-subtracted images and the background cube of the reference's per-target stage (the premade-cube leg)
		self.cubes = engine.synth_fill(ctx, scene, images=extras, images_err=True, backgrounds=extras, raw=True)
		self.cubes['raw_err'] = self.cubes['images_err']
		self.batch = pipeline.ApertureBatch(ctx, scene, cubes={'raw': self.cubes['raw'], 'raw_err': self.cubes['raw_err']})
		self.works = [pipeline.ApertureWork(ctx, self.batch, packed=True, psf=self.psf, capacity=self.capacity, cat_capacity=cat_capacity) for _ in range(self.nbuf)]
		for w in self.works[1:]:   # the background series and the sum image are scratch of the step, not outputs: shared
			w.bkg_raw, w.bkg, w.sumimage = self.works[0].bkg_raw, self.works[0].bkg, self.works[0].sumimage
		self.layout, self.block_nbytes = self.works[0].block_layout, self.works[0].block.nbytes
		self.lin = self.lin_out = None
		if self.psf:
			from . import psf as hpsf, simulate
			prf = simulate.synthetic_prf(seed=1)   # synthetic stand-in for the SPOC PRF file (git-LFS object upstream)
			self.lin = pipeline.LinPSFBatch(ctx, scene, hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow']),
				images=self.cubes['raw'], subtract=self.works[0].bkg, work=self.works[0])
			self.lin_out = [self.lin.out] + [self.lin.result_for(w) for w in self.works[1:]]
		self._pipeline = pipeline

	def step(self, b=0):
		self._pipeline.aperture_step(self.ctx, self.batch, self.works[b])
		if self.psf:
			self._pipeline.linpsf_step(self.ctx, self.lin, out=self.lin_out[b])

	def block(self, b=0):
		return self.works[b].block

	def sync(self):
		self.ctx.sync()


# --------------------------------------------------------------------------------------------------
# the run
# --------------------------------------------------------------------------------------------------
class ShardedRun(object):
	"""
	The step loop of one rank with the per-step gather of the output block to rank 0.

	``gather``: ``'rccl'`` (device blocks, ``tp_comm_gather`` on a second high-priority stream: direct send / recv pairs in one
	group, so the root receives from its N - 1 peers at once), ``'host'`` (the block goes through host memory and
	``dist.gather``: CPU workers, and the control-flow fallback when ranks share a device), ``'auto'`` (RCCL when every rank
	has a device of its own and the communicator comes up on all of them, else host), ``'none'``.
	With one rank nothing is gathered: the block of the last step is the result.
	"""

	def __init__(self, worker, n_total, rank=0, world=1, dist=None, torch=None, gather='auto', shared_device=False):
		self.worker, self.n_total, self.rank, self.world, self.dist, self.torch = worker, int(n_total), int(rank), int(world), dist, torch
		self.sizes = tpcomm.shard_sizes(n_total, world)
		self.range = tpcomm.shard_range(n_total, world, rank)
		if worker.n_local != self.sizes[rank]:
			raise ValueError(f'rank {rank} holds {worker.n_local} targets, its shard of {n_total} over {world} ranks is {self.sizes[rank]}')
		if world > 1 and worker.capacity < max(self.sizes):
			raise ValueError('the block capacity must be the largest shard: every rank sends the same number of bytes')
		self.nbuf = worker.nbuf
		self.cat_sizes = None
		if 'cat_in_mask' in worker.layout:
			self.cat_sizes = [int(worker.n_cat_local)]
			if world > 1:
				self.cat_sizes = [None] * world
				dist.all_gather_object(self.cat_sizes, int(worker.n_cat_local))
		self.comm_ctx = None
		self.recv = [None] * self.nbuf
		self.gather_ms = []
		self.last_buffer = None
		self.mode = 'none (single rank)' if world == 1 else 'disabled'
		if world > 1 and gather != 'none':
			self.mode = self._open_gather(gather, shared_device)
		ctx = worker.ctx
		self.ev_done = [ctx.event() for _ in range(self.nbuf)] if ctx is not None else None
		self.ev_free = [ctx.event() for _ in range(self.nbuf)] if ctx is not None else None
		self._host_recv = None

	@property
	def gathers(self):
		return self.world > 1 and not self.mode.startswith(('none', 'disabled'))

	def _open_gather(self, gather, shared_device):
		w = self.worker
		if w.ctx is None or gather == 'host':
			return 'host (gloo)'
		from .device import Context
		# its copy kernels must not queue behind a grid that fills every CU
		self.comm_ctx = Context(w.ctx.device, high_priority=True)
		ok, note = 1, None
		if shared_device:
			ok, note = 0, 'ranks share a GPU: RCCL needs one device per rank'
		else:
			try:
				tpcomm.init_from_torch(self.comm_ctx, self.dist, self.rank, self.world)
			except Exception as e: # noqa: B902
				ok, note = 0, f'RCCL communicator not created ({e})'
		t = self.torch.tensor([ok], dtype=self.torch.int32)
		self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
		if int(t[0]) == 1:
			if self.rank == 0:
				self.recv = [w.ctx.empty((self.world, w.block_nbytes), 'uint8') for _ in range(self.nbuf)]
			return 'rccl'
		if gather == 'rccl':
			raise RuntimeError('RCCL gather asked for but not available: ' + (note or 'another rank failed'))
		# control-flow fallback (never on a real multi-GPU node): the block goes through host memory and gloo
		return 'host-gloo fallback: ' + (note or 'RCCL unavailable on another rank')

	# ---- one step's gather ---------------------------------------------------------------------
	def _gather(self, b):
		w = self.worker
		if self.mode == 'rccl':
			self.comm_ctx.wait_event(self.ev_done[b])
			self.comm_ctx.timer_start(b)
			tpcomm.gather(self.comm_ctx, w.block(b), self.recv[b], root=0)
			self.comm_ctx.timer_stop(b)
			self.comm_ctx.record(self.ev_free[b])
			return
		w.sync()
		t0 = time.perf_counter()
		blk = w.block(b)
		h = self.torch.from_numpy(np.ascontiguousarray(blk.to_host() if hasattr(blk, 'to_host') else blk))
		out = [self.torch.empty_like(h) for _ in range(self.world)] if self.rank == 0 else None
		self.dist.gather(h, out, dst=0)
		if self.rank == 0:
			self._host_recv = [o.numpy() for o in out]
		self.gather_ms.append((time.perf_counter() - t0) * 1e3)

	def run_steps(self, n, collect=False):
		"""``n`` steps; with ``collect`` the durations of the gathers are recorded (``gather_ms``)."""
		w, ctx = self.worker, self.worker.ctx
		rccl = self.mode == 'rccl'
		for s in range(n):
			b = s % self.nbuf
			if rccl and s >= self.nbuf:
				if collect:
					self.gather_ms.append(self.comm_ctx.timer_ms(b))   # waits for gather s - nbuf (long finished)
				ctx.wait_event(self.ev_free[b])                        # block b has left for rank 0: it may be overwritten
			w.step(b)
			self.last_buffer = b
			if self.gathers:
				if ctx is not None:
					ctx.record(self.ev_done[b])
				self._gather(b)
		if rccl and collect:
			for s in range(max(0, n - self.nbuf), n):
				self.gather_ms.append(self.comm_ctx.timer_ms(s % self.nbuf))

	def sync(self):
		self.worker.sync()
		if self.comm_ctx is not None:
			self.comm_ctx.sync()

	def barrier(self):
		if self.dist is not None:
			self.dist.barrier()

	def max_over_ranks(self, value):
		if self.dist is None:
			return float(value)
		t = self.torch.tensor([float(value)], dtype=self.torch.float64)
		self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
		return float(t[0])

	# ---- results -------------------------------------------------------------------------------
	def collect(self):
		"""
		Rank 0: the arrays of the LAST step in global target order, ``{name: array}`` as ``comm.assemble_blocks`` lays them out
		(``lc`` ``(5, N, T)``, everything else with the target axis first), trimmed to the real shard sizes.  Other ranks: None.
		"""
		self.sync()
		if self.rank != 0 or self.last_buffer is None:
			return None
		w, b = self.worker, self.last_buffer
		if not self.gathers:
			blk = w.block(b)
			blocks = [blk.to_host() if hasattr(blk, 'to_host') else np.asarray(blk)]
		elif self.mode == 'rccl':
			blocks = list(self.recv[b].to_host())
		else:
			blocks = self._host_recv
		return tpcomm.assemble_blocks(blocks, w.layout, self.sizes, cat_sizes=self.cat_sizes)

	@staticmethod
	def skip_lists(cat_in_mask, cat_offsets, cat_starid, target_starid):
		"""The reference's ``skip_targets`` per target (photometry.py:269-272): the catalogue stars inside its mask, itself excluded."""
		out = []
		for i in range(len(target_starid)):
			a, b = int(cat_offsets[i]), int(cat_offsets[i + 1])
			inside = np.asarray(cat_in_mask[a:b]).astype(bool)
			out.append([int(s) for s in np.asarray(cat_starid[a:b])[inside] if int(s) != int(target_starid[i])])
		return out

	@staticmethod
	def replay(result, starids, tmags, skip_lists, priorities=None):
		"""The master's bookkeeping on the gathered results (taskmanager.py:460-532): final statuses in global target order."""
		return tpcomm.replay_skip_targets(starids, tmags, skip_lists, result['status'], priorities=priorities)

	def close(self):
		if self.comm_ctx is not None:
			self.comm_ctx.close()
			self.comm_ctx = None
