# -*- coding: utf-8 -*-
"""
The sharded run: one process per GPU, the targets of a batch divided over the ranks by index, every rank running the
per-target stages on its share, the per-step output blocks gathered to rank 0 and put back in global target order, the
master-side skip-target bookkeeping replayed on the gathered results.

This is what ``run_tessphot_mpi.py:74-209`` is to the reference (a master handing one pickled task at a time to MPI workers and
saving what comes back, ``taskmanager.py:460-532``) -- re-designed for a node of GPUs: targets are independent (SURVEY.md
section 8e), so the division is static (``comm.shard_range``), there is **no data-path collective** except the gather of each
step's output block, and that gather is issued on a second stream from the other half of a double-buffered block so that it
overlaps the next step's compute (``when='step'``) -- or, the default, issued ONCE after the last step (``when='final'``).

Pieces (``bench.py`` and ``INTEGRATION.md`` section 3 call these; ``tests/test_distributed_gloo.py`` runs the whole entry on
CPU ranks with uneven shards):

* :func:`spawn_ranks` -- start the ranks as fresh child processes BEFORE anything touches the GPU;
* :func:`rank_environment`, :func:`init_host_group` -- rank / world from the launcher's environment, the host group
  (``hostgroup.SocketGroup`` by default: TCP, standard library only; ``kind='gloo'``: ``torch.distributed``) used for
  rendezvous, barriers and (on a box where the ranks share a device) the host fallback of the gather;
* :class:`ShardWorker` -- what a rank does per step (protocol); :class:`DeviceShardWorker` -- the BASELINE workloads on the
  device (configs[2] aperture + background, configs[4] aperture + PSF) on synthetic cubes generated in HBM;
* :class:`ShardedRun` -- the step loop with the gather (RCCL: ``tp_comm_gather``; host fallback: the host group) of the
  COMPACT block (``comm.compact_block_layout``: the float32-exact light-curve planes travel as float32), once after the last
  step by default (north_star's "final light-curve gather") or double-buffered under every next step,
  reassembly (``comm.assemble_blocks``: every rank sends the same padded capacity, the real sizes trim it) and the replay of
  the skip lists (``comm.replay_skip_targets``).
"""

import os
import subprocess
import sys
import time
import numpy as np
from . import comm as tpcomm


# --------------------------------------------------------------------------------------------------
# ranks
# --------------------------------------------------------------------------------------------------
def spawn_ranks(script, argv, n_ranks, poll_s=0.2, grace_s=10.0):
	"""
	Start one fresh child process per rank (``script argv...`` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR and a rendezvous
	id of this launch set) BEFORE anything in this process touches the GPU, relay rank 0's standard output, return the worst
	exit code.  The ranks are watched together: when one fails, the others -- who would wait for it in a rendezvous or a
	barrier until a timeout -- are terminated and the failure's code is returned.
	"""
	import uuid
	rdzv = uuid.uuid4().hex
	procs = []
	for r in range(int(n_ranks)):
		env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR='127.0.0.1',
			TESSPHOT_RDZV_ID=rdzv, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
		env.pop('MASTER_PORT', None)
		procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
			stdout=None if r == 0 else subprocess.DEVNULL))
	rc = 0
	live = list(procs)
	while live:
		time.sleep(poll_s)
		for p in list(live):
			code = p.poll()
			if code is None:
				continue
			live.remove(p)
			if code != 0 and rc == 0:
				rc = abs(code) or 1
				for q in live:          # exactly the processes started above, by handle -- never by pattern
					q.terminate()
				deadline = time.monotonic() + grace_s
				for q in live:
					try:
						q.wait(max(0.0, deadline - time.monotonic()))
					except subprocess.TimeoutExpired:
						q.kill()
	for p in procs:
		p.wait()
	return rc


def rank_environment():
	"""``(rank, local_rank, world)`` as torch.distributed.run (or :func:`spawn_ranks`) exports them; ``(0, 0, 1)`` without a launcher."""
	return int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def init_host_group(rank, world, kind='socket'):
	"""
	The host-side group of a multi-rank run (rendezvous, barriers, max over ranks, the id of the RCCL communicator):
	``hostgroup.SocketGroup`` by default -- standard library only, **no PyTorch in a multi-GPU run** -- or, with
	``kind='gloo'``, ``torch.distributed``'s gloo group (torch is then imported here, BEFORE the HIP library).
	"""
	from . import hostgroup
	return hostgroup.open_group(rank, world, kind)


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
	The step loop of one rank with the gather of the output block to rank 0.

	``gather``: ``'rccl'`` (device blocks, ``tp_comm_gather`` on a second high-priority stream: direct send / recv pairs in one
	group, so the root receives from its N - 1 peers at once), ``'host'`` (the block goes through host memory and the host
	group: CPU workers, and the control-flow fallback when ranks share a device), ``'auto'`` (RCCL when every rank has a device
	of its own and the communicator comes up on all of them, else host), ``'none'``.
	``when``: ``'final'`` (the default) -- one gather after the last step of a ``run_steps`` call (north_star's "final light-curve
	gather": nothing to hide, nothing in the way of the steps); ``'step'`` -- the block of EVERY step is gathered, on the second
	stream from the other half of the double-buffered block, under the next step's compute.  :meth:`choose_when` picks between
	them from what was measured.  With one rank nothing is gathered: the block of the last step is the result.
	``compact`` (default): what travels is the compact block -- flux, flux_err and flux_background as the float32 values they
	are (``comm.compact_block_layout``; ``tp_block_compact`` on the gather's stream), widened again on rank 0: bit-identical
	results, ~25-30 % fewer bytes per rank.

	``group``: the host-side group (``hostgroup.SocketGroup`` / ``TorchGroup``); ``dist`` + ``torch`` (an initialised
	``torch.distributed`` group) are still accepted and wrapped.  ``run_steps`` may be called any number of times: which blocks
	have a gather in flight is kept on the instance.
	"""

	def __init__(self, worker, n_total, rank=0, world=1, group=None, dist=None, torch=None, gather='auto', when='final', shared_device=False, compact=True):
		from . import hostgroup
		if group is None:
			group = hostgroup.TorchGroup(dist, torch, rank, world) if (dist is not None and world > 1) else hostgroup.SingleGroup()
		if world > 1 and group.world != world:
			raise ValueError(f'the host group has {group.world} ranks, the run {world}')
		if when not in ('step', 'final'):
			raise ValueError("when must be 'step' or 'final'")
		self.worker, self.n_total, self.rank, self.world, self.group = worker, int(n_total), int(rank), int(world), group
		self.when = when
		self.sizes = tpcomm.shard_sizes(n_total, world)
		self.range = tpcomm.shard_range(n_total, world, rank)
		if worker.n_local != self.sizes[rank]:
			raise ValueError(f'rank {rank} holds {worker.n_local} targets, its shard of {n_total} over {world} ranks is {self.sizes[rank]}')
		if world > 1 and worker.capacity < max(self.sizes):
			raise ValueError('the block capacity must be the largest shard: every rank sends the same number of bytes')
		self.nbuf = worker.nbuf
		self.compact = bool(compact)
		self.send_nbytes = tpcomm.compact_block_layout(worker.layout)[1] if self.compact else worker.block_nbytes
		self.send = [None] * self.nbuf    # RCCL: the compact blocks on the device (made on the gather's stream)
		self.cat_sizes = None
		if 'cat_in_mask' in worker.layout:
			self.cat_sizes = group.allgather_int(worker.n_cat_local) if world > 1 else [int(worker.n_cat_local)]
		self.comm_ctx = None
		self.recv = [None] * self.nbuf
		self.gather_ms = []              # durations of the per-step gathers that were collected
		self.final_ms = []               # durations of the final gathers that were collected
		self.last_buffer = None
		self._gathered_buffer = None     # the block rank 0 last received
		self._in_flight = [False] * self.nbuf   # a gather of block b has been issued and block b not waited for since
		self._timer = [None] * self.nbuf        # a started timer of block b that has not been read: the list its duration goes to (or False)
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
			return 'host'
		from .device import Context
		# its copy kernels must not queue behind a grid that fills every CU
		self.comm_ctx = Context(w.ctx.device, high_priority=True)
		ok, note = 1, None
		if shared_device:
			ok, note = 0, 'ranks share a GPU: RCCL needs one device per rank'
		else:
			try:
				tpcomm.init_from_group(self.comm_ctx, self.group)
			except Exception as e: # noqa: B902
				ok, note = 0, f'RCCL communicator not created ({e})'
		if int(self.group.min(ok)) == 1:
			if self.rank == 0:
				self.recv = [w.ctx.empty((self.world, self.send_nbytes), 'uint8') for _ in range(self.nbuf)]
			if self.compact:
				self.send = [w.ctx.empty((self.send_nbytes,), 'uint8') for _ in range(self.nbuf)]
			return 'rccl'
		if gather == 'rccl':
			raise RuntimeError('RCCL gather asked for but not available: ' + (note or 'another rank failed'))
		# control-flow fallback (never on a real multi-GPU node): the block goes through host memory and the host group
		return 'host fallback: ' + (note or 'RCCL unavailable on another rank')

	# ---- one gather ----------------------------------------------------------------------------
	def _read_timer(self, b):
		"""The duration of the last timed gather of block ``b`` goes where it was asked for (waits for that gather)."""
		if self._timer[b] is not None:
			ms = self.comm_ctx.timer_ms(b)
			if self._timer[b] is not False:
				self._timer[b].append(ms)
			self._timer[b] = None

	def _gather(self, b, into):
		"""Queue (RCCL) or do (host) the gather of block ``b``; ``into``: the list its duration is appended to, or None."""
		w = self.worker
		self._gathered_buffer = b
		if self.mode == 'rccl':
			w.ctx.record(self.ev_done[b])
			self._read_timer(b)
			self.comm_ctx.wait_event(self.ev_done[b])
			self.comm_ctx.timer_start(b)
			blk = w.block(b)
			if self.compact:
				tpcomm.device_compact_block(self.comm_ctx, blk, self.send[b], w.layout)
				blk = self.send[b]
			tpcomm.gather(self.comm_ctx, blk, self.recv[b], root=0)
			self.comm_ctx.timer_stop(b)
			self.comm_ctx.record(self.ev_free[b])
			self._timer[b] = into if into is not None else False
			self._in_flight[b] = True
			return
		w.sync()
		t0 = time.perf_counter()
		blk = w.block(b)
		blk = blk.to_host() if hasattr(blk, 'to_host') else blk
		if self.compact:
			blk = tpcomm.compact_block(blk, w.layout)
		got = self.group.gather_array(blk, dst=0)
		if self.rank == 0:
			self._host_recv = got
		if into is not None:
			into.append((time.perf_counter() - t0) * 1e3)

	def run_steps(self, n, collect=False, when=None):
		"""``n`` steps; with ``collect`` the durations of the gathers are recorded (``gather_ms`` per step, ``final_ms``)."""
		w, ctx = self.worker, self.worker.ctx
		when = when or self.when
		rccl = self.mode == 'rccl'
		b = None
		for s in range(n):
			b = s % self.nbuf
			if rccl and self._in_flight[b]:
				# a gather of block b is (or was) on its way -- whichever call issued it: the block may be overwritten only once
				# it has left for rank 0
				self._read_timer(b)
				ctx.wait_event(self.ev_free[b])
				self._in_flight[b] = False
			w.step(b)
			self.last_buffer = b
			if self.gathers and when == 'step':
				self._gather(b, self.gather_ms if collect else None)
		if self.gathers and when == 'final' and b is not None:
			self._gather(b, self.final_ms if collect else None)
		if rccl and collect:
			for k in range(self.nbuf):
				self._read_timer(k)

	def choose_when(self, step_ms_without_gather):
		"""
		``'final'`` when the per-step gather cannot hide under a step (its measured mean duration, the slowest rank's, exceeds the
		step without a gather, the slowest rank's), else ``'step'``; the same answer on every rank.  Sets and returns ``self.when``.
		"""
		if self.gathers:
			mean = (sum(self.gather_ms) / len(self.gather_ms)) if self.gather_ms else 0.0
			self.when = 'final' if self.group.max(mean) > self.group.max(step_ms_without_gather) else 'step'
		return self.when

	def sync(self):
		self.worker.sync()
		if self.comm_ctx is not None:
			self.comm_ctx.sync()

	def barrier(self):
		self.group.barrier()

	def max_over_ranks(self, value):
		return self.group.max(value)

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
		if self.gathers and self._gathered_buffer != b:
			raise RuntimeError('the block of the last step has not been gathered')
		if not self.gathers:
			blk = w.block(b)
			blocks = [blk.to_host() if hasattr(blk, 'to_host') else np.asarray(blk)]
		elif self.mode == 'rccl':
			blocks = list(self.recv[b].to_host())
		else:
			blocks = self._host_recv
		if self.gathers and self.compact:
			blocks = [tpcomm.expand_block(c, w.layout) for c in blocks]
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
