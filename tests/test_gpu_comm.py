# -*- coding: utf-8 -*-
"""RCCL plumbing on one GPU: unique id, communicator of size 1, gather / all-gather round trip."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_roundtrip():
	from photometry_amd.device import Context
	from photometry_amd import comm as tpcomm
	with Context(0) as ctx:
		send = ctx.array(np.arange(4096, dtype='float64'))
		# without a communicator: degenerate copies
		recv = ctx.zeros((4096,), 'float64')
		tpcomm.gather(ctx, send, recv, root=0)
		np.testing.assert_array_equal(recv.to_host(), np.arange(4096))
		uid = tpcomm.unique_id()
		assert len(uid) == 128
		tpcomm.init(ctx, uid, 0, 1) # ncclCommInitRank with one rank
		recv2 = ctx.zeros((4096,), 'float64')
		tpcomm.allgather(ctx, send, recv2) # goes through ncclAllGather
		ctx.sync()
		np.testing.assert_array_equal(recv2.to_host(), np.arange(4096))
		recv3 = ctx.zeros((4096,), 'float64')
		tpcomm.gather(ctx, send, recv3, root=0)
		np.testing.assert_array_equal(recv3.to_host(), np.arange(4096))
