# -*- coding: utf-8 -*-
"""
Multi-GPU plumbing: target sharding (host logic) and the single RCCL gather of the
light-curve block (``tp_comm_*`` in the C ABI).

The reference distributes work with an MPI task-pull loop of pickled dicts
(run_tessphot_mpi.py:74-209); here targets are independent units sharded statically by index
(SURVEY.md section 8e), one process per GPU, no data-path collective except the final gather.
"""

import ctypes
import numpy as np
from . import _lib


def shard_range(n_items, world, rank):
	"""Contiguous block partition: rank r gets [r*ceil(n/world), (r+1)*ceil(n/world)) clipped to n."""
	per = -(-int(n_items) // int(world))
	a = min(rank * per, n_items)
	b = min(a + per, n_items)
	return a, b


def shard_sizes(n_items, world):
	return [shard_range(n_items, world, r)[1] - shard_range(n_items, world, r)[0] for r in range(world)]


def packed_block_layout(n_targets, n_cad, height, width, psf=False, align=256, n_cat=0, extras=False):
	"""
	Layout of the per-step output block of a rank: ONE allocation, so that the gather is one
	message per rank.  Returns ``(layout, nbytes)`` with ``layout[name] = (offset, shape, dtype)``:

	* ``lc`` float64 ``(5, Nt, T)``: flux, flux_err, flux_background, centroid column / row (the plugin's columns,
	  BasePhotometry.py:425-429); ``contamination`` float64, ``status`` / ``flags`` int32, ``mask`` uint8 ``(Nt, H, W)``;
	* with ``psf`` (BASELINE configs[4], aperture + PSF): ``psf_flux`` float64 ``(Nt, T)`` (the LinPSF light curve,
	  linpsf_photometry.py:168), ``psf_contamination`` float64 (PSF_CONT, :203-211), ``psf_status`` int32;
	* with ``n_cat`` > 0: ``cat_in_mask`` uint8 ``(n_cat,)``, one flag per row of the rank's (ragged) catalogue: the star lies
	  in the target's mask -- what the master's skip-target bookkeeping needs (photometry.py:269-272).  ``n_cat`` is a capacity
	  like ``n_targets``: the same on every rank;
	* with ``extras``: ``sumimage`` float64 ``(Nt, H, W)`` and ``diagnostics`` float64 ``(Nt, 10)`` (``engine.DIAGNOSTICS_COLUMNS``).
	"""
	Nt, T, H, W = int(n_targets), int(n_cad), int(height), int(width)
	fields = [('lc', (5, Nt, T), 'float64'), ('contamination', (Nt,), 'float64'), ('status', (Nt,), 'int32'), ('flags', (Nt,), 'int32'),
		('mask', (Nt, H, W), 'uint8')]
	if psf:
		fields += [('psf_flux', (Nt, T), 'float64'), ('psf_contamination', (Nt,), 'float64'), ('psf_status', (Nt,), 'int32')]
	if n_cat:
		fields += [('cat_in_mask', (int(n_cat),), 'uint8')]
	if extras:   # what the batched drop-in entry returns besides: the sum image and the light-curve diagnostics (one download per group)
		fields += [('sumimage', (Nt, H, W), 'float64'), ('diagnostics', (Nt, 10), 'float64')]
	layout, off = {}, 0
	for name, shape, dtype in fields:
		layout[name] = (off, shape, dtype)
		off = -(-(off + int(np.prod(shape)) * np.dtype(dtype).itemsize) // align) * align
	return layout, off


#: the planes of ``lc`` that are float32 sums widened on store (AperturePhotometry/photometry.py:172-201: flux, flux_err,
#: flux_background) -- a gathered block carries them as float32 and loses nothing; planes 3, 4 (the centroids) are float64
LC_FLOAT32_PLANES = 3


def compact_block_layout(layout, align=256):
	"""
	The layout of the block a rank SENDS (``run_tessphot_mpi.py:151-196``: the result message of a worker): as
	:func:`packed_block_layout`, but ``lc`` is split into ``lc32`` float32 ``(3, Nt, T)`` -- flux, flux_err, flux_background,
	exact in float32 -- and ``lc_centroid`` float64 ``(2, Nt, T)``.  Returns ``(clayout, nbytes, fields)``; ``fields`` =
	``[(src_offset, dst_offset, count, kind)]`` for ``tp_block_compact`` / :func:`compact_block` (kind 0: ``count`` bytes copied,
	kind 1: ``count`` float64 -> float32).
	"""
	clayout, fields, off = {}, [], 0

	def put(name, shape, dtype, src_off, kind):
		nonlocal off
		n = int(np.prod(shape))
		clayout[name] = (off, shape, dtype)
		fields.append((int(src_off), int(off), n if kind == 1 else n * np.dtype(dtype).itemsize, kind))
		off = -(-(off + n * np.dtype(dtype).itemsize) // align) * align

	for name, (src, shape, dtype) in layout.items():
		if name == 'lc':
			ncol, Nt, T = shape
			assert dtype == 'float64' and ncol == 5
			put('lc32', (LC_FLOAT32_PLANES, Nt, T), 'float32', src, 1)
			put('lc_centroid', (ncol - LC_FLOAT32_PLANES, Nt, T), 'float64', src + LC_FLOAT32_PLANES * Nt * T * 8, 0)
		else:
			put(name, shape, dtype, src, 0)
	return clayout, off, fields


def compact_block(block, layout):
	"""numpy counterpart of ``tp_block_compact``: the full block (uint8) -> the compact block (uint8)."""
	clayout, nbytes, fields = compact_block_layout(layout)
	block = np.asarray(block, dtype='uint8').ravel()
	out = np.zeros(nbytes, dtype='uint8')
	for src, dst, count, kind in fields:
		if kind == 0:
			out[dst:dst + count] = block[src:src + count]
		else:
			out[dst:dst + 4 * count].view('float32')[:] = block[src:src + 8 * count].view('float64').astype('float32')
	return out


def expand_block(cblock, layout, align=256):
	"""A compact block back into the full one (rank 0, after the gather): the float32 planes widened again.  For the planes
	that are float32 sums this is the identity on the original block, bit for bit (``tests/test_distributed_gloo.py``)."""
	clayout, _, _ = compact_block_layout(layout)
	c = unpack_block(cblock, clayout)
	nbytes = max(off + int(np.prod(shape)) * np.dtype(dtype).itemsize for off, shape, dtype in layout.values())
	out = np.zeros(-(-nbytes // align) * align, dtype='uint8')
	full = unpack_block(out, layout)
	for name in layout:
		if name == 'lc':
			full['lc'][:LC_FLOAT32_PLANES] = c['lc32']
			full['lc'][LC_FLOAT32_PLANES:] = c['lc_centroid']
		else:
			full[name][...] = c[name]
	return out


def device_compact_block(ctx, block, out, layout):
	"""``tp_block_compact`` on the context's stream: ``block`` (full, DeviceArray uint8) -> ``out`` (compact, DeviceArray uint8)."""
	_, nbytes, fields = compact_block_layout(layout)
	assert out.nbytes >= nbytes

	class Field(ctypes.Structure):
		_fields_ = [('src_offset', ctypes.c_uint64), ('dst_offset', ctypes.c_uint64), ('count', ctypes.c_uint64), ('kind', ctypes.c_int32), ('reserved', ctypes.c_int32)]
	arr = (Field * len(fields))(*[Field(s_, d_, c_, k_, 0) for s_, d_, c_, k_ in fields])
	ctx._check(ctx.lib.tp_block_compact(ctx.handle, block.ptr, out.ptr, ctypes.cast(arr, ctypes.c_void_p), len(fields)))


def unpack_block(block, layout):
	"""Views of the arrays inside one rank's block (a uint8 array of the size ``packed_block_layout`` returned)."""
	block = np.asarray(block, dtype='uint8').ravel()
	out = {}
	for name, (off, shape, dtype) in layout.items():
		n = int(np.prod(shape)) * np.dtype(dtype).itemsize
		out[name] = block[off:off + n].view(dtype).reshape(shape)
	return out


def assemble_blocks(blocks, layout, sizes, cat_sizes=None):
	"""
	Reassembly on rank 0 of the gathered per-rank blocks (every rank sends the same padded capacity; ``sizes`` = real number
	of targets per rank, ``cat_sizes`` = real number of catalogue rows per rank when the block carries ``cat_in_mask``) into
	arrays in global target order: ``{name: array}``, the target axis is axis 1 of ``lc`` and axis 0 of everything else;
	``cat_in_mask`` is the concatenation of the ranks' catalogue flags (global catalogue row = rank's first row + local row).
	"""
	parts = [unpack_block(b, layout) for b in blocks]
	out = {}
	for name in layout:
		ax = 1 if name == 'lc' else 0
		counts = sizes
		if name == 'cat_in_mask':
			if cat_sizes is None:
				raise ValueError('the blocks carry cat_in_mask: the catalogue rows per rank (cat_sizes) are needed to trim them')
			counts = cat_sizes
		out[name] = np.concatenate([np.take(p[name], np.arange(n), axis=ax) for p, n in zip(parts, counts)], axis=ax)
	return out


def assemble_gathered(blocks, sizes, n_columns=5):
	"""
	Host-side reassembly of gathered light-curve blocks.

	``blocks``: sequence (one per rank) of float64 arrays ``(n_columns, per_rank_capacity, T)`` (every rank
	sends the same padded capacity); ``sizes``: real number of targets per rank.
	Returns ``(n_columns, sum(sizes), T)`` in global target order.
	"""
	parts = [np.asarray(b)[:, :n, :] for b, n in zip(blocks, sizes)]
	return np.concatenate(parts, axis=1) if parts else np.zeros((n_columns, 0, 0))


def replay_skip_targets(starids, tmags, skip_lists, statuses, priorities=None, return_ran=False):
	"""
	The one cross-target dependency of the reference: the ``skip_targets`` bookkeeping the master does on the results it
	receives (AperturePhotometry/photometry.py:269-272 -> taskmanager.py:460-532), replayed for one gathered batch (one
	sector / camera / CCD / cadence / datasource group) in the order the reference's task loop would have run it
	(``get_task``: lowest priority whose status is still NULL, taskmanager.py:402; priority = ascending Tmag, todolist.py:584).

	For the target being saved, the rows of the todo-list named in its skip list are looked up (names that are no targets
	are ignored).  If there are any and the target is strictly brighter than ALL of them, every one of them becomes
	SKIPPED (whatever it was before); otherwise the target itself becomes SKIPPED and the others are left alone.  A target
	that is SKIPPED before its turn is never run, so its own skip list never takes effect.

	``priorities`` defaults to the stable ascending-Tmag order.  Returns the int32 status array; with ``return_ran`` also the
	indices of the targets that ran, in that order.
	Pinned by ``tests/golden/golden_skiptargets.npz`` (the reference's own TaskManager on sqlite todo-lists).
	"""
	tm = np.asarray(tmags, dtype='float64')
	n = len(tm)
	if priorities is None:
		order = np.argsort(tm, kind='stable')
	else:
		order = np.argsort(np.asarray(priorities), kind='stable')
	rows_of = {}
	for i, s in enumerate(starids):
		rows_of.setdefault(int(s), []).append(i)
	SKIPPED = 5
	final = np.zeros(n, dtype='int32') # 0 = status IS NULL
	ran = []
	for i in order:
		if final[i] != 0:
			continue # marked by an earlier result: get_task no longer hands it out
		ran.append(int(i))
		mine = int(statuses[i])
		rows = [j for s in set(int(x) for x in skip_lists[i]) for j in rows_of.get(s, ())]
		if rows:
			if np.all(tm[i] < tm[rows]):
				final[rows] = SKIPPED
			else:
				mine = SKIPPED
		final[i] = mine
	return (final, ran) if return_ran else final


# ---- RCCL ------------------------------------------------------------------------------------
def unique_id():
	lib = _lib.load()
	buf = ctypes.create_string_buffer(128)
	rc = lib.tp_comm_unique_id(buf, 128)
	if rc != 0:
		raise _lib.TessphotError(rc, (lib.tp_last_error(None) or b'').decode())
	return buf.raw


def init(ctx, uid, rank, world):
	ctx._check(ctx.lib.tp_comm_init(ctx.handle, uid, len(uid), int(rank), int(world)))


def init_from_group(ctx, group):
	"""Distribute the RCCL unique id (128 bytes, made on rank 0) through the host group (``hostgroup.SocketGroup`` /
	``TorchGroup``) and create the communicator on ``ctx``."""
	uid, err = b'', None
	if group.rank == 0:
		try:
			uid = unique_id()
		except Exception as e: # noqa: B902 -- the other ranks are waiting in the broadcast: tell them, then raise
			err = e
	uid = group.broadcast_bytes(uid, src=0)
	if err is not None:
		raise err
	if len(uid) != 128:
		raise RuntimeError('rank 0 could not make the RCCL unique id')
	init(ctx, uid, group.rank, group.world)


def init_from_torch(ctx, dist, rank, world):
	"""Distribute the RCCL unique id through an already initialised torch.distributed group."""
	obj = [unique_id() if rank == 0 else None]
	dist.broadcast_object_list(obj, src=0)
	init(ctx, obj[0], rank, world)


def gather(ctx, send, recv, root=0):
	"""``send``: DeviceArray (same size on every rank); ``recv``: DeviceArray on root (world x send), else None."""
	ctx._check(ctx.lib.tp_comm_gather(ctx.handle, send.ptr, None if recv is None else recv.ptr, send.nbytes, int(root)))


def allgather(ctx, send, recv):
	ctx._check(ctx.lib.tp_comm_allgather(ctx.handle, send.ptr, recv.ptr, send.nbytes))
