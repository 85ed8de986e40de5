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


def assemble_gathered(blocks, sizes, n_columns=5):
	"""
	Host-side reassembly of gathered light-curve blocks.

	``blocks``: sequence (one per rank) of float64 arrays ``(n_columns, per_rank_capacity, T)`` (every rank
	sends the same padded capacity); ``sizes``: real number of targets per rank.
	Returns ``(n_columns, sum(sizes), T)`` in global target order.
	"""
	parts = [np.asarray(b)[:, :n, :] for b, n in zip(blocks, sizes)]
	return np.concatenate(parts, axis=1) if parts else np.zeros((n_columns, 0, 0))


def replay_skip_targets(starids, tmags, skip_lists, statuses):
	"""
	The one cross-target dependency of the reference: ``skip_targets`` bookkeeping done by the
	master on gathered results (AperturePhotometry/photometry.py:244-250 ->
	taskmanager.py:460-532).  Results are replayed in priority order (ascending Tmag,
	todolist.py:584): a target that is named in the skip list of an already accepted brighter
	target is marked SKIPPED (5); if the *other* star is the brighter one, the current target is.

	Simplified restatement of TaskManager.save_result's resolution for one batch: returns the new
	int32 status array.
	"""
	order = np.argsort(np.asarray(tmags), kind='stable')
	status = np.array(statuses, dtype='int32', copy=True)
	index_of = {int(s): i for i, s in enumerate(starids)}
	tm = np.asarray(tmags)
	for i in order:
		if status[i] not in (1, 3):
			continue
		for other in skip_lists[i]:
			j = index_of.get(int(other))
			if j is None or status[j] == 5:
				continue
			if tm[j] >= tm[i]:
				status[j] = 5 # the fainter star inside our mask is skipped
			else:
				status[i] = 5 # we sit inside the mask of a brighter star
				break
	return status


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
