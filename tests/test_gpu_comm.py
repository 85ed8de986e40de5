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


def _two_rank_worker(rank, uid_file, out_file):
	import os, sys, time
	sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
	import numpy as np
	from photometry_amd.device import Context
	from photometry_amd import comm as tpcomm
	from photometry_amd._lib import TessphotError
	ctx = Context(0)
	if rank == 0:
		uid = tpcomm.unique_id()
		with open(uid_file + '.tmp', 'wb') as fh:
			fh.write(uid)
		os.replace(uid_file + '.tmp', uid_file)
	else:
		for _ in range(600):
			if os.path.exists(uid_file):
				break
			time.sleep(0.05)
		uid = open(uid_file, 'rb').read()
	try:
		tpcomm.init(ctx, uid, rank, 2)
	except TessphotError as e:
		if rank == 0:
			np.save(out_file, np.array([-1.0]))
			open(out_file + '.msg', 'w').write(str(e))
		return
	n = 1 << 16
	send = ctx.array(np.full(n, rank + 1, dtype='float64') * np.arange(n))
	recv = ctx.zeros((2, n), 'float64') if rank == 0 else None
	for _ in range(3): # repeated, like the per-step gather of bench.py
		tpcomm.gather(ctx, send, recv, root=0)
	ctx.sync()
	if rank == 0:
		got = recv.to_host()
		ok = np.array_equal(got[0], np.arange(n)) and np.array_equal(got[1], 2.0 * np.arange(n))
		np.save(out_file, np.array([1.0 if ok else 0.0]))
	ctx.close()


def test_rccl_two_ranks_gather(tmp_path):
	"""tp_comm_gather itself with TWO RCCL ranks (two processes).  A 1-GPU box can only offer both ranks the same device, which
	RCCL may refuse (duplicate GPU): that refusal is reported as a skip with RCCL's message, anything else must work."""
	import multiprocessing as mp
	uid_file, out_file = str(tmp_path / 'uid.bin'), str(tmp_path / 'ok.npy')
	mpc = mp.get_context('spawn')
	procs = [mpc.Process(target=_two_rank_worker, args=(r, uid_file, out_file)) for r in range(2)]
	for p in procs:
		p.start()
	for p in procs:
		p.join(120)
	for p in procs:
		if p.is_alive():
			p.terminate()
			pytest.fail("two-rank gather hung")
	res = float(np.load(out_file)[0])
	if res < 0:
		pytest.skip("RCCL refused two ranks on one device: " + open(out_file + '.msg').read()[:200])
	assert res == 1.0


def test_compact_block_on_device_equals_numpy_and_loses_nothing():
	"""tp_block_compact on a real step's packed block (aperture + LinPSF, configs[4]'s block in small): equal to comm.compact_block
	byte for byte, and expanding it gives the ORIGINAL block back bit for bit -- flux, flux_err and flux_background are float32
	sums widened on store, so the compact block a rank sends loses nothing."""
	from photometry_amd import simulate, engine, pipeline, psf as hpsf, comm as tpcomm
	from photometry_amd.device import Context
	ctx = Context(0)
	Nt, T, H, W = 300, 203, 15, 15
	scene = simulate.make_scene(Nt, T, H, W, seed=12)
	scene.aperture = None
	cubes = engine.synth_fill(ctx, scene, images=False, images_err=True, backgrounds=False, raw=True)
	batch = pipeline.ApertureBatch(ctx, scene, cubes={'raw': cubes['raw'], 'raw_err': cubes['images_err']})
	work = pipeline.ApertureWork(ctx, batch, packed=True, psf=True, capacity=Nt + 20)
	prf = simulate.synthetic_prf(seed=1)
	lin = pipeline.LinPSFBatch(ctx, scene, hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow']),
		images=cubes['raw'], subtract=work.bkg, work=work)
	pipeline.aperture_step(ctx, batch, work)
	pipeline.linpsf_step(ctx, lin)
	ctx.sync()
	block = work.block.to_host()
	clay, cn, fields = tpcomm.compact_block_layout(work.block_layout)
	assert cn < 0.8 * block.nbytes
	out = ctx.zeros((cn,), 'uint8')
	tpcomm.device_compact_block(ctx, work.block, out, work.block_layout)
	ctx.sync()
	got = out.to_host()
	np.testing.assert_array_equal(got, tpcomm.compact_block(block, work.block_layout))
	np.testing.assert_array_equal(tpcomm.expand_block(got, work.block_layout), block)
	ctx.close()
