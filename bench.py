#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""
bench.py -- whole-job throughput of the per-target photometry hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch of synthetic stamp cubes already resident in HBM, for BASELINE.json
configs[2] ("aperture + background", the configuration the metric is quoted on): 10 000 targets x 1 300 cadences x 15x15
per GPU.  The resident inputs are the RAW flux cube and its error cube; one step =

    B*  per-cadence stamp background (sigma-clipped SExtractor mode)      tp_background_stamp
    B2  time smoothing of the background series (prepare.py:317-335)      tp_smooth_time
    A1 + A2..A5b + A7 + A6: AperturePhotometry.do_photometry for every target, with the background subtracted on the fly
        (B3, prepare.py:419-420) and summed in the aperture                tp_aperture_photometry

N > 1: one process per GPU.  Started as the driver starts it (torch.distributed.run: RANK / WORLD_SIZE in the environment)
or plainly as `python bench.py --gpus N`, in which case this process spawns its N ranks itself before touching the GPU.
Targets are sharded by index (weak scaling: 10 000 targets per GPU); the only data-path exchange is the gather of each
step's output block (light curves + contamination + status + flags + mask, one message per rank) to rank 0 over RCCL,
issued EVERY step on a second stream from the other half of a double-buffered output block, so that it overlaps the next
step's compute; its duration is reported separately.

Rank 0 prints ONE JSON line.  Besides the contract's fields it carries `roofline` (dominant kernel of the timed step),
`rooflines` (every HBM-bound kernel of the step, necessary bytes / time / 8 TB/s -- a fraction above 1 is impossible by
construction), `cpu_baseline`, and at N = 1 the extra legs `aperture_premade_cubes` (the per-target stage alone, SURVEY 8d's
reading), `linpsf` (BASELINE configs[3]) and `end_to_end` (cubes start in pinned host memory: H2D overlapped with compute).
"""

import os
for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS', 'NUMEXPR_NUM_THREADS'):
	os.environ.setdefault(_v, '1') # the CPU baseline runs one process per core: no BLAS / OpenMP oversubscription

import argparse
import json
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E peak (MI355X_MICROARCH.md)
FP64_VALU_TFLOPS = 78.6  # MI355X FP64 vector peak
XGMI_LINK_GBS = 153.0


def parse_args(argv=None):
	p = argparse.ArgumentParser()
	p.add_argument('--gpus', type=int, default=1)
	p.add_argument('--steps', type=int, default=10)
	p.add_argument('--warmup', type=int, default=3)
	p.add_argument('--targets', type=int, default=0, help='targets per GPU (0: 10 000 for workload c2, 12 500 for c4 weak, 100 000 / N for c4 strong)')
	p.add_argument('--workload', choices=('c2', 'c4'), default=None, help='c2: aperture + background (BASELINE configs[2]; default at N = 1); '
		'c4: aperture + PSF, the LinPSF fit in the step and in the gathered block (BASELINE configs[4]; default at N > 1)')
	p.add_argument('--scaling', choices=('weak', 'strong'), default='weak', help='c4: weak = 12 500 targets per GPU, strong = 100 000 targets over the N GPUs')
	p.add_argument('--cadences', type=int, default=1300)
	p.add_argument('--stamp', type=int, default=15)
	p.add_argument('--cpu-sample', type=int, default=4, help='targets per worker process of the CPU baseline (0 = skip)')
	p.add_argument('--cpu-procs', type=int, default=0, help='worker processes of the all-core CPU baseline (0 = physical cores)')
	p.add_argument('--seed', type=int, default=1)
	p.add_argument('--no-gather', action='store_true')
	p.add_argument('--no-extra', action='store_true', help='skip the extra legs (premade cubes, LinPSF, end to end, stages)')
	p.add_argument('--e2e-targets', type=int, default=2048, help='targets of the end-to-end (H2D included) leg (0 = skip)')
	p.add_argument('--frame', type=int, default=1024, help='side of the frame stack of the stamp-cutter stage (0 = skip)')
	p.add_argument('--psf-targets', type=int, default=4096, help='targets of the non-linear PSF photometry leg (0 = skip)')
	p.add_argument('--fullframe-frames', type=int, default=16, help='2048 x 2048 frames of the full-frame background / pixel-flag leg (0 = skip)')
	p.add_argument('--frames-targets', type=int, default=2500, help='targets of the frames-to-results leg on a 512 x 512 stack (0 = skip)')
	return p.parse_args(argv)


# --------------------------------------------------------------------------------------------------
# rank spawning (python bench.py --gpus N without a launcher)
# --------------------------------------------------------------------------------------------------
def spawn_ranks(args):
	"""Start one fresh child process per rank BEFORE anything in this process touches the GPU; relay rank 0's line."""
	with socket.socket() as s:
		s.bind(('127.0.0.1', 0))
		port = s.getsockname()[1]
	procs = []
	for r in range(args.gpus):
		env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
			MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
		procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
			stdout=None if r == 0 else subprocess.DEVNULL))
	rc = 0
	for p in procs:
		rc = max(rc, abs(p.wait()))
	return rc


# --------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (reference-equivalent numpy restatement, per-cadence Python loops like the reference)
# --------------------------------------------------------------------------------------------------
_CPU_JOBS = None


def _cpu_worker(job):
	"""
	One step on a list of targets, as the reference would run it per target: B* per-cadence stamp background (oracle of the
	build-defined estimator), B2 smoothing, B3 subtraction, A1 sum image, K2P2 masks, A6 extraction, A7.
	The downstream stages use the DEVICE's background series so that their results can be compared bit for bit; the
	oracle's own background is compared with it at 1e-6 and its time is counted.  Returns (seconds, results).
	"""
	import numpy as np
	try:
		from threadpoolctl import threadpool_limits
		threadpool_limits(1)
	except Exception: # noqa: B902
		pass
	from oracle import sumimage as osum, aperture as oap, backgrounds as ob
	t0 = time.perf_counter()
	out = []
	for i in range(job.n_targets):
		raw = job.raw[i]
		bkg_raw = ob.background_series(raw)                                   # B*
		bkg = ob.smooth_time(job.dev_bkg_raw[i], job.time_smooth)              # B2 (on the device's series: bit-exact check)
		ob.smooth_time(bkg_raw, job.time_smooth)                               # B2 of the oracle's own series (timed)
		series = job.dev_bkg[i][None, None, :]
		img, err = ob.subtract_background(raw, job.raw_err[i], series)         # B3
		S = osum.sumimage(img, job.quality)                                    # A1
		bcube = np.broadcast_to(series.astype('float32'), img.shape)
		try:
			r = oap.do_photometry(S, img, err, bcube, tuple(job.stamps[i]), job.target_pos_row[i], job.target_pos_column[i],
				job.target_tmag[i], job.target_starid[i], job.catalog_of(i), job.aperture[i])
		except Exception: # noqa: B902 -- tessphot.py:37-49: any exception is STATUS.ERROR
			r = {'status': 2}
		d = {k: r.get(k) for k in ('status', 'flux', 'flux_err', 'flux_background', 'mask')}
		with np.errstate(invalid='ignore', divide='ignore'):
			both = np.isfinite(bkg_raw) & np.isfinite(job.dev_bkg_raw[i])
			d['bkg_nan_equal'] = bool(np.array_equal(np.isnan(bkg_raw), np.isnan(job.dev_bkg_raw[i])))
			d['bkg_max_rel'] = float(np.max(np.abs(bkg_raw[both] / job.dev_bkg_raw[i][both] - 1))) if both.any() else 0.0
		d['smooth_equal'] = bool(np.array_equal(bkg, job.dev_bkg[i], equal_nan=True))
		out.append(d)
	return time.perf_counter() - t0, out


def _cpu_worker_indexed(c):
	return _cpu_worker(_CPU_JOBS[c])


def cgroup_cpu_quota():
	"""CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited."""
	try:
		q, p = open('/sys/fs/cgroup/cpu.max').read().split()
		return None if q == 'max' else float(q) / float(p)
	except Exception: # noqa: B902
		pass
	try:
		q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
		p = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
		return None if q <= 0 else q / p
	except Exception: # noqa: B902
		return None


def physical_cores():
	avail = len(os.sched_getaffinity(0))
	try:
		import psutil
		phys = psutil.cpu_count(logical=False) or avail
	except Exception: # noqa: B902
		phys = avail
	return max(1, min(avail, phys)), avail


def cpu_baseline(ctx, scene, cubes, work, args, T, H, W, time_smooth):
	"""Oracle on a bounded sample of the same device-generated cubes: one core, 16 processes, all physical cores."""
	import numpy as np
	import multiprocessing as mp
	global _CPU_JOBS
	phys, avail = physical_cores()
	nproc = args.cpu_procs if args.cpu_procs > 0 else phys
	per = max(1, args.cpu_sample)
	ns = min(scene.n_targets, nproc * per)
	nproc = max(1, ns // per)
	ns = nproc * per
	sub = scene.subset(slice(0, ns))
	for name, key in (('raw', 'raw'), ('raw_err', 'raw_err')):
		cube = cubes[key]
		host = np.empty((ns, H, W, cube.t_pitch), dtype='float32')
		ctx._check(ctx.lib.tp_memcpy_d2h(ctx.handle, host.ctypes.data, cube.ptr, host.nbytes))
		setattr(sub, name, np.ascontiguousarray(host[..., :T]))
		del host
	sub.dev_bkg_raw = work.bkg_raw.slice0(0, ns).to_host()[:, :T]
	sub.dev_bkg = work.bkg.slice0(0, ns).to_host()[:, :T]
	sub.time_smooth = time_smooth
	sub.aperture = np.ones((ns, H, W), dtype='int32')

	def part(s, sl):
		p = s.subset(sl)
		for k in ('dev_bkg_raw', 'dev_bkg'):
			setattr(p, k, getattr(s, k)[sl])
		p.time_smooth = s.time_smooth
		return p

	# (a) one process, one core: the analogue of one MPI worker of run_tessphot_mpi.py
	n1 = min(ns, max(2, per))
	t1, _ = _cpu_worker(part(sub, slice(0, n1)))
	# (b) worker processes (forked: the sample is shared copy-on-write, nothing is pickled in); rate = targets / slowest
	#     worker's compute time.  All physical cores, and 16 processes as the round-1 reference point.
	rates = {}
	rr_all = None
	for n in sorted({min(16, nproc), nproc}):
		_CPU_JOBS = [part(sub, slice(c, n * per, n)) for c in range(n)]
		with mp.get_context('fork').Pool(n) as pool:
			rr = pool.map(_cpu_worker_indexed, range(n))
		rates[n] = n * per / max(r[0] for r in rr)
		if n == nproc:
			rr_all = rr
	# parity of the sample while we are here
	lc = work.lc.to_host()
	masks = work.mask.to_host()
	status = work.status.to_host()
	bad = bad_bkg = 0
	max_rel = 0.0
	for c, (_, out) in enumerate(rr_all):
		for j, r in enumerate(out):
			i = c + j * nproc
			ok = int(status[i]) == r['status']
			if ok and r.get('mask') is not None:
				ok = np.array_equal(masks[i].astype(bool), r['mask']) and np.array_equal(lc['flux'][i], r['flux'], equal_nan=True) \
					and np.array_equal(lc['flux_err'][i], r['flux_err'], equal_nan=True) \
					and np.array_equal(lc['flux_background'][i], r['flux_background'], equal_nan=True)
			bad += (not ok)
			bad_bkg += (not r['bkg_nan_equal']) or (r['bkg_max_rel'] > 1e-6) or (not r['smooth_equal'])
			max_rel = max(max_rel, r['bkg_max_rel'])
	best = max(rates.values())
	return {
		'value': best, 'unit': 'targets/s', 'cores': max(rates, key=rates.get), 'kind': 'port',
		'sample': f'{ns} of the {scene.n_targets} targets of the same device-generated raw cubes ({per} per worker process); oracle = numpy '
			'restatement of the reference per-cadence loops: stamp background (B*, B2, B3) + sum image + K2P2 + extraction; rate = '
			'targets / slowest worker compute time; BLAS / OpenMP threads pinned to 1',
		'rates_by_process_count': {str(k): v for k, v in rates.items()},
		'single_core_targets_per_s': n1 / t1,
		'calibration': 'dev-container timing of the reference\'s own AperturePhotometry.do_photometry loop (mask given, 15x15x1300) beside this '
			'restatement on the same core (tests/golden/time_reference.py): 0.0916 s/target against 0.0889 -- the port takes 0.97 x the reference\'s time',
		'host_cores': {'physical': phys, 'usable_logical': avail, 'cgroup_cpu_quota': cgroup_cpu_quota()},
	}, {'targets': ns, 'mismatches': int(bad), 'background_mismatches': int(bad_bkg), 'background_max_rel_err': max_rel,
		'what': 'status / mask / flux / flux_err / flux_background bit-exact given the device background; B* within 1e-6 of the oracle, '
			'B2 bit-exact'}


# --------------------------------------------------------------------------------------------------
def kernel_rows(report, n_launch_units, alg=None, necessary=None):
	out = {}
	for name, (n, ms) in report.items():
		avg = ms / n
		k = {'launches': n, 'avg_ms': avg}
		if necessary and name in necessary:
			k['necessary_bytes_per_launch'] = necessary[name]
			k['necessary_GBps'] = necessary[name] / (avg * 1e-3) / 1e9
			k['frac_of_hbm_peak'] = k['necessary_GBps'] / HBM_PEAK_GBS
		if alg and name in alg:
			k['survey_8d_bytes_per_launch'] = alg[name] * n_launch_units
		out[name] = k
	return out


def roofline_of(name, rows, traffic, note=None):
	k = rows[name]
	r = {'kernel': name, 'bound': 'hbm', 'achieved': k['necessary_GBps'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
		'frac': k['necessary_GBps'] / HBM_PEAK_GBS, 'traffic': (traffic or {}).get(name),
		'avg_kernel_ms': k['avg_ms'], 'bytes_per_launch': k['necessary_bytes_per_launch'],
		'bytes': 'necessary bytes per launch (what the kernel cannot avoid reading / writing) = SURVEY 8d per-target figure x targets '
			'per launch, with A6 charged only for the rows of in-mask pixels it needs'}
	if 'survey_8d_bytes_per_launch' in k:
		r['survey_8d_bytes_per_launch_all_rows'] = k['survey_8d_bytes_per_launch']
	if note:
		r['note'] = note
	return r


def main():
	args = parse_args()
	if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
		sys.exit(spawn_ranks(args))
	rank = int(os.environ.get('RANK', '0'))
	local_rank = int(os.environ.get('LOCAL_RANK', '0'))
	world = int(os.environ.get('WORLD_SIZE', '1'))
	args.gpus = world

	# torch is plumbing only, and only for N > 1 (gloo rendezvous, barrier, max over ranks); imported BEFORE the HIP library so
	# that one HIP runtime is shared.  A single-GPU run never loads it.
	dist = None
	torch = None
	if world > 1:
		import torch
		import torch.distributed as dist
		os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
		# gloo announces its connections on stdout (C level): keep stdout for the ONE JSON line
		sys.stdout.flush()
		saved = os.dup(1)
		os.dup2(2, 1)
		try:
			dist.init_process_group(backend='gloo', rank=rank, world_size=world)
		finally:
			os.dup2(saved, 1)
			os.close(saved)

	import numpy as np
	from photometry_amd import simulate, engine, pipeline, _lib
	from photometry_amd.device import Context, DeviceCube
	from photometry_amd import comm as tpcomm
	import ctypes

	# ranks of one node use one GPU each; on a box with fewer GPUs than ranks they share devices (control-flow smoke run)
	ndev = ctypes.c_int(0)
	_lib.load().tp_device_count(ctypes.byref(ndev))
	if ndev.value <= 0:
		raise RuntimeError("bench.py needs a GPU: " + (_lib.load().tp_last_error(None) or b'').decode())
	shared_device = world > ndev.value
	device = local_rank % ndev.value
	ctx = Context(device)
	use_torch_cuda = torch is not None and torch.cuda.is_available()   # N > 1 only
	if use_torch_cuda:
		torch.cuda.set_device(device)

	def device_sync(*ctxs):
		for c in (ctx,) + ctxs:
			c.sync()
		if use_torch_cuda:
			torch.cuda.synchronize(device)

	def barrier():
		if dist is not None:
			dist.barrier()

	workload = args.workload or ('c4' if world > 1 else 'c2')
	psf = workload == 'c4'
	T, H = args.cadences, args.stamp
	if args.targets > 0:
		Nt = args.targets
	elif workload == 'c2':
		Nt = 10000
	elif args.scaling == 'weak':
		Nt = 12500                                            # BASELINE configs[4]: 100 000 targets over 8 GPUs
	else:
		Nt = -(-100000 // world)                              # strong scaling: the 100 000 targets over the N GPUs there are
	W = H
	P = H * W
	scene = simulate.make_scene(Nt, T, H, W, seed=args.seed * 1000 + rank)
	scene.aperture = None
	extras = (world == 1) and (workload == 'c2') and not args.no_extra
	# resident inputs: raw flux + error cubes (2 x 11.8 GB at the default size); the premade-cube leg adds the
	# background-subtracted images and the background cube of the reference's per-target stage
	cubes = engine.synth_fill(ctx, scene, images=extras, images_err=True, backgrounds=extras, raw=True)
	cubes['raw_err'] = cubes['images_err']
	batch = pipeline.ApertureBatch(ctx, scene, cubes={'raw': cubes['raw'], 'raw_err': cubes['raw_err']})
	nbuf = 2 if world > 1 else 1
	works = [pipeline.ApertureWork(ctx, batch, packed=True, psf=psf) for _ in range(nbuf)]
	for w in works[1:]: # the background series are scratch of the step, not outputs: shared
		w.bkg_raw, w.bkg = works[0].bkg_raw, works[0].bkg
	lin = lin_out = None
	if psf:
		# configs[4] "aperture + PSF": the LinPSF fit of every target on the same raw cube, the step's background series subtracted
		# on the fly; its light curve / contamination / status are part of the gathered block
		from photometry_amd import psf as hpsf
		prf = simulate.synthetic_prf(seed=1) # synthetic stand-in for the SPOC PRF file (git-LFS object upstream)
		lin = pipeline.LinPSFBatch(ctx, scene, hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow']),
			images=cubes['raw'], subtract=works[0].bkg, work=works[0])
		lin_out = [lin.out] + [lin.result_for(w) for w in works[1:]]

	# ---- the per-step gather (world > 1): RCCL on a second stream, double-buffered ------------------------------
	do_gather = world > 1 and not args.no_gather
	comm_ctx = None
	gather_mode = 'none (single GPU)' if world == 1 else 'disabled (--no-gather)'
	recv = [None, None]
	block_bytes = works[0].block.nbytes
	if do_gather:
		comm_ctx = Context(device, high_priority=True) # its copy kernels must not queue behind a grid that fills every CU
		ok, note = 1, None
		if shared_device:
			ok, note = 0, f'{world} ranks share {ndev.value} GPU(s): RCCL needs one device per rank'
		else:
			try:
				tpcomm.init_from_torch(comm_ctx, dist, rank, world)
			except Exception as e: # noqa: B902
				ok, note = 0, f'RCCL communicator not created ({e})'
		t = torch.tensor([ok], dtype=torch.int32)
		dist.all_reduce(t, op=dist.ReduceOp.MIN)
		if int(t[0]) == 1:
			gather_mode = 'rccl'
			if rank == 0:
				recv = [ctx.empty((world, block_bytes), 'uint8') for _ in range(2)]
		else:
			# control-flow fallback (never on a real multi-GPU node): the block goes through host memory and gloo
			gather_mode = 'host-gloo fallback: ' + (note or 'RCCL unavailable on another rank')
	ev_done = [ctx.event() for _ in range(nbuf)]
	ev_free = [ctx.event() for _ in range(nbuf)]
	gather_ms = []

	def gather_block(b):
		if gather_mode == 'rccl':
			comm_ctx.wait_event(ev_done[b])
			comm_ctx.timer_start(b)
			tpcomm.gather(comm_ctx, works[b].block, recv[b], root=0)
			comm_ctx.timer_stop(b)
			comm_ctx.record(ev_free[b])
		else:
			ctx.sync()
			t0 = time.perf_counter()
			h = torch.from_numpy(works[b].block.to_host())
			dist.gather(h, [torch.empty_like(h) for _ in range(world)] if rank == 0 else None, dst=0)
			gather_ms.append((time.perf_counter() - t0) * 1e3)

	def run_steps(n, collect=False):
		for s in range(n):
			b = s % nbuf
			if do_gather and gather_mode == 'rccl' and s >= nbuf:
				if collect:
					gather_ms.append(comm_ctx.timer_ms(b)) # waits for gather s - nbuf (long finished)
				ctx.wait_event(ev_free[b]) # block b has left for rank 0: it may be overwritten
			pipeline.aperture_step(ctx, batch, works[b])
			if psf:
				pipeline.linpsf_step(ctx, lin, out=lin_out[b])
			if do_gather:
				ctx.record(ev_done[b])
				gather_block(b)
		if do_gather and gather_mode == 'rccl' and collect:
			for s in range(max(0, n - nbuf), n):
				gather_ms.append(comm_ctx.timer_ms(s % nbuf))

	run_steps(args.warmup)
	device_sync(*([comm_ctx] if comm_ctx else []))
	barrier()
	del gather_ms[:]
	ctx.profile(True)
	ctx.profile_reset()
	t0 = time.perf_counter()
	run_steps(args.steps, collect=True)
	device_sync(*([comm_ctx] if comm_ctx else []))
	barrier()
	elapsed = time.perf_counter() - t0
	ctx.profile(False)
	if dist is not None:
		t = torch.tensor([elapsed], dtype=torch.float64)
		dist.all_reduce(t, op=dist.ReduceOp.MAX)
		elapsed = float(t[0])
	prof = ctx.profile_report()
	work = works[(args.steps - 1) % nbuf] if args.steps > 0 else works[0]
	# the same step without the gather (N > 1): what the overlap has to hide the gather under
	step_alone_ms = None
	if do_gather:
		nalone = max(1, min(args.steps, 5))
		device_sync(comm_ctx)
		barrier()
		t1 = time.perf_counter()
		for _ in range(nalone):
			pipeline.aperture_step(ctx, batch, works[0])
			if psf:
				pipeline.linpsf_step(ctx, lin, out=lin_out[0])
		device_sync()
		step_alone_ms = (time.perf_counter() - t1) / nalone * 1e3
		barrier()

	result = None
	if rank == 0:
		n_mask = float(work.mask.to_host().astype('int64').sum())
		# SURVEY 8d algorithmic bytes per target (A6 charged with every row of its cubes) ...
		alg = {
			'tp_bkg_stamp_kernel': P*T*4 + T*4,
			'tp_bkg_smooth_kernel': 2*T*4,
			'tp_aperture_fused_kernel': (P*T*4 + T*4 + P*8) + (2*P*T*4 + T*4 + P + 5*T*8),
		}
		# ... and the bytes per launch the kernels cannot avoid: A1 needs the whole raw cube, A6 only the rows of in-mask pixels
		necessary = {
			'tp_bkg_stamp_kernel': Nt * (P*T*4 + T*4),
			'tp_bkg_smooth_kernel': Nt * 2*T*4,
			'tp_aperture_fused_kernel': Nt * (P*T*4 + 2*T*4 + P*8) + 2 * n_mask * T * 4 + Nt * (T*4 + P + 5*T*8),
		}
		rows = kernel_rows(prof, Nt, alg, necessary)
		traffic = None
		tfile = os.path.join(ROOT, 'profiles', 'r3_traffic.json')
		if os.path.exists(tfile) and (Nt, T, H) == (10000, 1300, 15):
			traffic = json.load(open(tfile)).get('traffic_bytes_per_launch')
		hbm_kernels = [k for k in rows if k in necessary and k != 'tp_bkg_smooth_kernel']
		dom = max(hbm_kernels, key=lambda k: rows[k]['avg_ms'])
		notes = {
			'tp_bkg_stamp_kernel': 'B*: streams the raw cube once, but is bound by the vector ALUs (a 256-key sorting network per frame '
				'for the sigma-clipped median), not by HBM; traffic = bytes (profiles/).  Its instructions (v_min / v_max / v_med3, DPP '
				'moves, FP64) issue at one wave64 instruction per 4 cycles on gfx950 (tools/lab/valu_rate.hip): ~9 500 cycles per '
				'wavefront of 8 frames, i.e. the vector-ALU time of this launch is ~%.1f ms at 2.4 GHz on 1 024 SIMDs' % (Nt * ((T + 7) // 8) * 9500.0 / 1024 / 2.4e9 * 1e3),
			'tp_aperture_fused_kernel': 'the aperture-sum kernel north_star names (A1 + K2P2 + A6 fused, one wavefront per target)',
		}
		rooflines = [roofline_of(k, rows, traffic, notes.get(k)) for k in sorted(hbm_kernels, key=lambda k: -rows[k]['avg_ms'])]
		step_bytes = sum(necessary[k] for k in rows if k in necessary)
		n_total = Nt * world if not (psf and args.scaling == 'strong' and args.targets == 0) else min(100000, Nt * world)
		if psf:
			nfit = lin.n_fit_stars
			metric = 'targets/sec (whole node), 100k targets x 1300 cad x 15x15, aperture + PSF, target-sharded'
			wl = (f'{Nt} targets/GPU x {T} cadences x {H}x{W} stamps, aperture + PSF (BASELINE configs[4]: 100 000 targets over 8 GPUs = 12 500 per GPU): '
				'raw flux and error cubes resident in HBM; per step the stamp background of every cadence (B*), its time smoothing (B2), '
				'AperturePhotometry.do_photometry of every target with the background subtracted on the fly (B3), and the LinPSF fit of every '
				f'target (linpsf_photometry: P1 table blend + P2-P4, {nfit} fitted stars on this rank) on the same cube; light curves of both '
				'methods in the gathered block')
		else:
			metric = 'targets/sec (whole node), 10k targets x 1300 cad x 15x15, aperture + background'
			wl = (f'{Nt} targets/GPU x {T} cadences x {H}x{W} stamps, aperture + background (BASELINE configs[2]): raw flux and '
				'error cubes resident in HBM; per step the stamp background of every cadence (B*), its time smoothing (B2), and '
				'AperturePhotometry.do_photometry of every target (sum image, K2P2 mask, extraction of flux / error / centroid / '
				'background) with the background subtracted on the fly (B3)')
		result = {
			'metric': metric,
			'value': n_total * args.steps / elapsed, 'unit': 'targets/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
			'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': args.scaling if psf else 'weak', 'vs_baseline': None,
			'dtype': 'f32 (aperture, background) + f64 (PSF fit)' if psf else 'f32', 'data': 'synthetic',
			'config': {'workload': wl, 'baseline_config': 'configs[4]' if psf else 'configs[2]',
				'targets_per_gpu': Nt, 'targets_total': n_total, 'cadences': T, 'stamp': [H, W],
				'parallelism': f'targets sharded over {world} GPU(s), one process per GPU, no data-path collective but the gather of the output block'},
			'roofline': next(r for r in rooflines if r['kernel'] == dom),
			'rooflines': rooflines,
			'step_hbm': {'necessary_bytes_per_step': step_bytes, 'GBps_over_whole_step': step_bytes / (elapsed / args.steps) / 1e9,
				'frac_of_hbm_peak': step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 'mean_mask_pixels': n_mask / Nt},
			'kernels': rows,
			'gather': {'mode': gather_mode, 'bytes_per_rank_per_step': block_bytes if world > 1 else 0,
				'block': 'light curves [5][Nt][T] f64 + contamination f64 + status, flags i32 + mask u8 per target' + (' + LinPSF light curve [Nt][T] f64, contamination f64, status i32' if psf else '') + ', one message per rank (comm.packed_block_layout)',
				'issued': 'every step, second stream, double-buffered output block' if do_gather else None,
				'mean_ms': (sum(gather_ms) / len(gather_ms)) if gather_ms else None,
				'step_ms_without_gather': step_alone_ms,
				'ideal_ms_one_xgmi_link': block_bytes / (XGMI_LINK_GBS * 1e9) * 1e3 if world > 1 else None,
				'xgmi_rate_assumed': f'{XGMI_LINK_GBS} GB/s one way per link: the root receives from its N - 1 peers on N - 1 links at once (direct '
					'send / recv pairs in one RCCL group), so the gather is bound by ONE inbound link per peer; if the 153 GB/s of the guide is the '
					'bidirectional figure the ideal doubles -- mean_ms beside step_ms_without_gather is the measurement that decides'},
		}
		if psf:
			# the LinPSF fit is the longest part of the configs[4] step: its own roofline (FP64 vector ALU, executed flops)
			# the fit launches of the star counts overlap (side streams), so their kernel times do not add up: the fit's share of the
			# step is the step's wall time less the kernels that run alone (plan, coefficient store and the host's look at the plan's
			# totals are then inside it)
			fit_names = ('tp_linpsf_fit_kernel', 'tp_linpsf_fitm_kernel', 'tp_linpsf_plan_kernel', 'tp_linpsf_coef_kernel')
			# (the step WITHOUT the gather where one was timed: a slow gather must not be booked on the fit)
			wall_ms = step_alone_ms if step_alone_ms is not None else elapsed / args.steps * 1e3
			fit_ms = wall_ms - sum(v[1] for k, v in prof.items() if k not in fit_names) / max(args.steps, 1)
			counts = np.diff(lin.star_offsets_h)
			fma = nfit * T * 79 * 24 + float(np.sum(counts * (counts + 1) / 2 + counts)) * T * H * W + nfit * 3 * 79 * 1170
			result['linpsf_roofline'] = {'kernel': 'tp_linpsf_plan_kernel + tp_linpsf_coef_kernel + tp_linpsf_fitm_kernel', 'bound': 'fp64 pipe (matrix + vector instructions share it)',
				'achieved': 2 * fma / (fit_ms * 1e-3) / 1e12, 'peak': FP64_VALU_TFLOPS, 'unit': 'TFLOP/s', 'frac': 2 * fma / (fit_ms * 1e-3) / 1e12 / FP64_VALU_TFLOPS,
				'kernel_ms_per_step': fit_ms, 'fitted_stars': int(nfit)}
		if shared_device:
			result['warning'] = f'{world} ranks shared {ndev.value} GPU(s): a control-flow run, not a scaling measurement'

	# ---- extra legs, N = 1 only -------------------------------------------------------------------------------
	if rank == 0 and extras:
		result['aperture_premade_cubes'] = leg_premade(ctx, scene, cubes, args, Nt, T, H, W, np, engine, pipeline)
		result['stages'] = leg_stages(ctx, scene, cubes, batch, work, args, Nt, T, H, W, np, engine, pipeline)
	if rank == 0 and world == 1 and workload == 'c2' and args.cpu_sample > 0:
		cb, parity = cpu_baseline(ctx, scene, cubes, work, args, T, H, W, batch.time_smooth)
		result['cpu_baseline'] = cb
		result['parity_sample'] = parity
		result['speedup_vs_cpu_baseline'] = result['value'] / cb['value']
		result['speedup_vs_one_core'] = result['value'] / cb['single_core_targets_per_s']
	if rank == 0 and extras:
		if args.e2e_targets > 0:
			result['end_to_end'] = leg_end_to_end(ctx, scene, cubes, args, T, H, W, np, engine, pipeline, Context, DeviceCube)
		for k in ('images', 'backgrounds'):
			cubes[k].free()
		result['linpsf'] = leg_linpsf(ctx, scene, cubes, work, args, Nt, T, H, W, np, engine, pipeline)
		if args.frames_targets > 0 and (T, H) == (1300, 15):
			result['frames_to_results'] = leg_frames(ctx, args, T, np, pipeline)
		for k in ('raw', 'images_err'):
			cubes[k].free()
		if args.psf_targets > 0:
			result['psf_fit'] = leg_psf_fit(ctx, args, np, engine)
		if args.fullframe_frames > 0:
			result['fit_background_frames'] = leg_fullframe(ctx, args, np)

	if rank == 0:
		print(json.dumps(result))
		sys.stdout.flush()
	if dist is not None:
		dist.barrier()
		dist.destroy_process_group()
	if comm_ctx is not None:
		comm_ctx.close()
	ctx.close()


# --------------------------------------------------------------------------------------------------
def leg_premade(ctx, scene, cubes, args, Nt, T, H, W, np, engine, pipeline):
	"""The per-target stage alone on premade cubes (background-subtracted images, errors, background cube): SURVEY 8d's reading
	of configs[2] and the round-1 headline.  Fused kernel only; same HIP-event timing."""
	P = H * W
	batch = pipeline.ApertureBatch(ctx, scene, cubes={k: cubes[k] for k in ('images', 'images_err', 'backgrounds')})
	work = pipeline.ApertureWork(ctx, batch)
	pipeline.aperture_step(ctx, batch, work)
	ctx.sync()
	ctx.profile(True)
	ctx.profile_reset()
	n = max(3, min(args.steps, 10))
	t0 = time.perf_counter()
	for _ in range(n):
		pipeline.aperture_step(ctx, batch, work)
	ctx.sync()
	ms = (time.perf_counter() - t0) / n * 1e3
	ctx.profile(False)
	prof = ctx.profile_report()
	n_mask = float(work.mask.to_host().astype('int64').sum())
	necessary = {'tp_aperture_fused_kernel': Nt * (P*T*4 + T*4 + P*8) + 3 * n_mask * T * 4 + Nt * (P + 5*T*8)}
	alg = {'tp_aperture_fused_kernel': (P*T*4 + T*4 + P*8) + (3*P*T*4 + P + 5*T*8)}
	rows = kernel_rows(prof, Nt, alg, necessary)
	tfile = os.path.join(ROOT, 'profiles', 'r3_traffic.json')
	traffic = json.load(open(tfile)).get('traffic_bytes_per_launch_premade') if os.path.exists(tfile) and (Nt, T, H) == (10000, 1300, 15) else None
	# three stand-alone kernels (A1, K2P2, A6), per-stage durations when a stage owns the GPU
	pipeline.aperture_step(ctx, batch, work, fused=False)
	ctx.sync()
	ctx.profile(True)
	ctx.profile_reset()
	for _ in range(2):
		pipeline.aperture_step(ctx, batch, work, fused=False)
	ctx.sync()
	ctx.profile(False)
	three = {k: {'launches': v[0], 'avg_ms': v[1] / v[0]} for k, v in ctx.profile_report().items()}
	return {'what': 'tp_aperture_photometry on premade cubes (images, errors, background cube: the inputs of the reference per-target stage); '
		'no background estimation in the step', 'targets_per_s': Nt / (ms * 1e-3), 'ms_per_step': ms,
		'roofline': roofline_of('tp_aperture_fused_kernel', rows, traffic), 'kernels': rows, 'three_kernel_path': three}


def leg_stages(ctx, scene, cubes, batch, work, args, Nt, T, H, W, np, engine, pipeline):
	"""Stages beside the step: materialised B3, light-curve diagnostics, stamp cutter."""
	P = H * W
	out = {}
	scratch = cubes['images'] # overwritten: the premade leg is done
	ctx.profile(True)
	ctx.profile_reset()
	for _ in range(3):
		engine.subtract_background(ctx, cubes['raw'], work.bkg, images=scratch)
	ctx.sync()
	r = ctx.profile_report()['tp_bkg_subtract_kernel']
	nb = Nt * (2*P*T*4 + T*4)
	out['subtract_materialised'] = {'what': 'B3 as its own pass (raw cube -> images cube); the step subtracts on the fly instead',
		'kernel': 'tp_bkg_subtract_kernel', 'avg_ms': r[1] / r[0], 'necessary_bytes_per_launch': nb,
		'necessary_GBps': nb / (r[1] / r[0] * 1e-3) / 1e9, 'frac_of_hbm_peak': nb / (r[1] / r[0] * 1e-3) / 1e9 / HBM_PEAK_GBS}
	ctx.profile_reset()
	for _ in range(3):
		pipeline.aperture_diagnostics(ctx, batch, work)
	ctx.sync()
	r = ctx.profile_report()['tp_diagnostics_kernel']
	out['diagnostics'] = {'what': 'light-curve diagnostics of every target (BasePhotometry.py:1343-1407) from the device-resident outputs',
		'avg_ms': r[1] / r[0]}
	if args.frame > 0:
		FR = args.frame
		frames = ctx.zeros((T, FR, FR), 'float32')
		rng = np.random.default_rng(args.seed)
		r0 = rng.integers(0, FR - H, Nt)
		c0 = rng.integers(0, FR - W, Nt)
		cst = ctx.array(np.stack((r0, r0 + H, c0 + 44, c0 + 44 + W), axis=1).astype('int32'))
		engine.cut_stamps(ctx, frames, cst, H, W, 0, 44, out=scratch)
		ctx.profile_reset()
		for _ in range(3):
			engine.cut_stamps(ctx, frames, cst, H, W, 0, 44, out=scratch)
		ctx.sync()
		r = ctx.profile_report()['tp_cut_stamps_kernel']
		# necessary bytes: every frame pixel that lies in some stamp read once + every cube element written once (stamps overlap:
		# SURVEY 8d's 2 P T 4 per target counts a shared pixel once per stamp and is kept as the side figure)
		covered = np.zeros((FR, FR), dtype=bool)
		for a, b in zip(r0, c0):
			covered[a:a + H, b:b + W] = True
		nb = int(covered.sum()) * T * 4 + Nt * P*T*4
		out['cutout'] = {'what': f'stamp cutter: {Nt} stamps cut from a {FR} x {FR} x {T} float32 frame stack resident in HBM '
			'(BasePhotometry._load_cube for the batch), one cube; frame-tile-major: tiles of 2 x 64 pixels x 64 frames through LDS, '
			'stamps served from the tile', 'kernel': 'tp_cut_tiles_kernel', 'timed': 'the three binning passes + the NaN pre-fill + tp_cut_tiles_kernel (profile entry tp_cut_stamps_kernel)', 'avg_ms': r[1] / r[0],
			'necessary_bytes_per_launch': nb, 'necessary_GBps': nb / (r[1] / r[0] * 1e-3) / 1e9, 'frac_of_hbm_peak': nb / (r[1] / r[0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
			'distinct_frame_pixels_in_stamps': int(covered.sum()), 'survey_8d_bytes_per_launch': Nt * 2*P*T*4}
		frames.free()
	ctx.profile(False)
	return out


def leg_end_to_end(ctx, scene, cubes, args, T, H, W, np, engine, pipeline, Context, DeviceCube):
	"""
	SURVEY 8d timing (ii): the cubes start in HOST memory, as a drop-in plugin receives them from BasePhotometry._load_cube.
	Pinned staging buffers, chunks of targets, the upload of chunk i+1 on a second stream while chunk i is processed, the
	light curves copied back asynchronously.  PCIe-bound by construction (2.35 MB in per target, 52 KB out).
	"""
	n = min(args.e2e_targets, scene.n_targets)
	chunk = 256
	n = max(chunk, n // chunk * chunk)
	sub = scene.subset(slice(0, n))
	up = Context(ctx.device, high_priority=False)
	host = {}
	for key in ('raw', 'raw_err'):
		host[key] = ctx.pinned((n, H, W, T), 'float32')
		full = np.empty((chunk, H, W, cubes[key].t_pitch), dtype='float32')
		for a in range(0, n, chunk):
			ctx._check(ctx.lib.tp_memcpy_d2h(ctx.handle, full.ctypes.data, cubes[key].slice0(a, chunk).ptr, full.nbytes))
			host[key].array[a:a + chunk] = full[..., :T]
	out_host = ctx.pinned((n // chunk, 5, chunk, T), 'float64')
	bufs = []
	for _ in range(2):
		dc = {k: DeviceCube(ctx, chunk, T, H, W) for k in ('raw', 'raw_err')}
		for c in dc.values():
			c.data.fill_bytes(0)
		bufs.append(dc)
	batches, works = [], []
	for a in range(0, n, chunk):
		b = pipeline.ApertureBatch(ctx, sub.subset(slice(a, a + chunk)), cubes=bufs[(a // chunk) % 2])
		batches.append(b)
		works.append(pipeline.ApertureWork(ctx, b))
	ev_up = [ctx.event() for _ in range(2)]
	ev_used = [ctx.event() for _ in range(2)]

	def run():
		for i, a in enumerate(range(0, n, chunk)):
			s = i % 2
			if i >= 2:
				up.wait_event(ev_used[s])
			for key in ('raw', 'raw_err'):
				bufs[s][key].upload_async(up, host[key], first_target=a)
			up.record(ev_up[s])
			ctx.wait_event(ev_up[s])
			pipeline.aperture_step(ctx, batches[i], works[i])
			ctx.record(ev_used[s])
			ctx.download_async(out_host, works[i].lc.block, host_offset=i * 5 * chunk * T * 8)
		up.sync()
		ctx.sync()

	run()
	t0 = time.perf_counter()
	reps = 2
	for _ in range(reps):
		run()
	dt = (time.perf_counter() - t0) / reps
	in_bytes = 2 * n * H * W * T * 4
	res = {'what': f'{n} targets whose raw + error cubes start in pinned host memory (reference (H, W, T) layout): chunks of {chunk} targets, '
		'H2D on a second stream overlapped with the step of the previous chunk, light curves copied back; file I/O excluded',
		'targets_per_s': n / dt, 'h2d_GBps': in_bytes / dt / 1e9, 'bound': 'PCIe (2.35 MB in per target)', 'seconds': dt}
	for h in list(host.values()) + [out_host]:
		h.free()
	up.close()
	return res


def leg_frames(ctx, args, T, np, pipeline):
	"""
	The batched drop-in entry as a scheduler would call it: a CCD region's frame stacks (images, errors, backgrounds) resident
	in HBM, ``tessphot_frames`` from target list to per-target results -- default stamps, catalogue selection, stamp cuts on
	the device, the fused pass, the stamp-resize rounds, diagnostics, download and the host-side bookkeeping per target.
	"""
	from photometry_amd import tessphot_frames
	N, FR, Tn = args.frames_targets, 512, 100
	rng = np.random.default_rng(args.seed + 7)
	rows, cols, tmag = rng.uniform(12, FR - 12, N), rng.uniform(12, FR - 12, N), rng.uniform(9.0, 14.0, N)
	img = np.zeros((FR, FR))
	yy, xx = np.mgrid[-4:5, -4:5]
	for r, c, m in zip(rows, cols, tmag):
		ri, ci = int(round(r)), int(round(c))
		img[ri - 4:ri + 5, ci - 4:ci + 5] += 10**(-0.4 * (m - 20.451)) * np.exp(-0.5 * ((yy + ri - r)**2 + (xx + ci - c)**2) / 0.81) / (2 * np.pi * 0.81)
	reps = (T + Tn - 1) // Tn
	base = (img[None] * (1 + 1e-3 * rng.normal(size=Tn))[:, None, None]).astype('float32')
	noise = np.sqrt(np.abs(base) + 200.0).astype('float32')
	images = (base + 30.0 + rng.standard_normal(base.shape).astype('float32') * noise).astype('float32')
	frames = {'images': np.tile(images, (reps, 1, 1))[:T], 'images_err': np.tile(noise, (reps, 1, 1))[:T],
		'backgrounds': np.full((T, FR, FR), 100.0, dtype='float32')}
	del base, noise, images
	tstamp = 1500.0 + np.arange(T) * 1800.0 / 86400.0
	quality = np.zeros(T, dtype='int32')
	cat = {'starid': np.arange(N, dtype='int64') + 1, 'tmag': tmag.astype('float32'), 'row': rows.astype('float32'), 'column': (cols + 44).astype('float32')}
	targets = {'starid': cat['starid'].copy(), 'tmag': tmag, 'row': rows, 'column': cols + 44}
	stack = pipeline.FrameStack(ctx, frames, 0, 44)
	del frames
	ctx.sync()
	tessphot_frames(ctx, stack, {k: v[:128] for k, v in targets.items()}, cat, tstamp, quality)
	t0 = time.perf_counter()
	out = tessphot_frames(ctx, stack, targets, cat, tstamp, quality)
	good = int(np.sum((out.status == 1) | (out.status == 3)))
	resized = int(np.sum(out.stamp_resizes > 0))
	dt = time.perf_counter() - t0
	# every per-target object as well (what the list-based entry of round 2 built unconditionally)
	t1 = time.perf_counter()
	n_obj = sum(1 for b in out if b.status.value in (1, 3))
	dobj = time.perf_counter() - t1
	return {'what': f'tessphot_frames: {N} targets on a {FR} x {FR} x {T} region resident in HBM (three frame stacks) -> columnar results '
		'(status, stamp, resizes, diagnostics, light curves, masks in arrays; per-target objects on demand): stamp cuts, fused pass, '
		'stamp-resize rounds and diagnostics on the device, default stamps / catalogue selection / decisions on the host (one Python process)',
		'targets_per_s': N / dt, 'seconds': dt, 'ok_or_warning': good, 'targets_resized': resized,
		'with_every_per_target_object': {'targets_per_s': N / (dt + dobj), 'seconds_for_the_objects': dobj, 'ok_or_warning': n_obj}}


def linpsf_traffic(Nt, T, H):
	"""HBM bytes per step of the LinPSF fit from the committed PMC passes (profiles/r3_traffic.json), for the default size only."""
	tfile = os.path.join(ROOT, 'profiles', 'r3_traffic.json')
	if os.path.exists(tfile) and (Nt, T, H) == (10000, 1300, 15):
		return json.load(open(tfile)).get('traffic_bytes_per_launch', {}).get('tp_linpsf_fit')
	return None


def leg_linpsf(ctx, scene, cubes, work, args, Nt, T, H, W, np, engine, pipeline):
	"""BASELINE configs[3]: linpsf_photometry PSF-fit path over the same cube (raw cube + on-the-fly background subtraction)."""
	from photometry_amd import simulate, psf as hpsf
	prf = simulate.synthetic_prf(seed=1) # synthetic stand-in for the SPOC PRF file (git-LFS object upstream)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	batch = pipeline.LinPSFBatch(ctx, scene, model, images=cubes['raw'], subtract=work.bkg)
	pipeline.linpsf_step(ctx, batch)
	ctx.sync()
	ctx.profile(True)
	ctx.profile_reset()
	n = max(3, min(args.steps, 5))
	t0 = time.perf_counter()
	for _ in range(n):
		pipeline.linpsf_step(ctx, batch)
	ctx.sync()
	ms = (time.perf_counter() - t0) / n * 1e3
	ctx.profile(False)
	prof = ctx.profile_report()
	kernels = {name: {'launches': c, 'avg_ms': t / c, 'ms_per_step': t / n} for name, (c, t) in prof.items()}
	# the fit = plan + coefficient store + one fit launch per star count.  The fit launches overlap (side streams), so their kernel
	# times do not add up: the fit's share is the step's wall time less the kernels that run alone (P1 blend, finalisation) -- the
	# host's look at the plan's totals is then inside it
	fit_names = ('tp_linpsf_fit_kernel', 'tp_linpsf_fitm_kernel', 'tp_linpsf_plan_kernel', 'tp_linpsf_coef_kernel')
	fit_ms = ms - sum(v['ms_per_step'] for k, v in kernels.items() if k not in fit_names)
	nfit = batch.n_fit_stars
	counts = np.diff(batch.star_offsets_h)
	# ALGORITHMIC flops (what the path needs, the same count as in rounds 1-2): per star-cadence ~79 pixels inside the 5 px
	# cut-off x 24 FMAs of a biquartic; per cadence and finite pixel the normal equations S(S+1)/2 + S FMAs; per (star, visited
	# table origin, pixel) item the 13x13 -> 5x5 contraction (~1 170 FMAs), ~3 origins per star.  The matrix-core fit executes
	# more than that: dense 16 x 16 tiles, 28-52 basis products instead of 24 nested multiplications (PMC: 91.5 M
	# v_mfma_f64_16x16x4_f64 per step on this batch = 1.9e14 flops against 1.1e14 algorithmic)
	fma = nfit * T * 79 * 24 + float(np.sum(counts * (counts + 1) / 2 + counts)) * T * H * W + nfit * 3 * 79 * 1170
	flops = 2.0 * fma
	nbytes = Nt * (H*W*T*4 + T*4) + nfit * T * 16 + Nt * T * 8
	res = {
		'metric': 'targets/sec, 10k targets x 1300 cad x 15x15, linpsf_photometry PSF fit (BASELINE configs[3])',
		'value': Nt / (ms * 1e-3), 'unit': 'targets/s', 'ms_per_step': ms, 'steps': n, 'dtype': 'f64', 'fitted_stars': int(nfit),
		'config': {'workload': f'{Nt} targets x {T} cadences x {H}x{W}, LinPSF fit of {nfit} stars (P1 table blend + P2-P4), raw cube resident, '
			'background series subtracted on the fly'},
		'roofline': {'kernel': 'tp_linpsf_plan_kernel + tp_linpsf_coef_kernel + tp_linpsf_fitm_kernel (matrix-core fit; tp_linpsf_fit_kernel = the vector-ALU fit of the targets that do not qualify)',
			'bound': 'fp64 pipe (not HBM): FP64 matrix and vector instructions share one pipe on this chip and have the same peak',
			'achieved': flops / (fit_ms * 1e-3) / 1e12, 'peak': FP64_VALU_TFLOPS, 'unit': 'TFLOP/s', 'frac': flops / (fit_ms * 1e-3) / 1e12 / FP64_VALU_TFLOPS,
			'flops': 'algorithmic FP64 flops of the path (estimate, see bench.py:leg_linpsf); the matrix-core fit executes ~1.7 x that', 'kernel_ms_per_step': fit_ms, 'kernel_ms_note': 'wall time of the step less the kernels that run alone: the fit launches of the star counts overlap',
			'hbm': {'necessary_bytes_per_step': nbytes, 'GBps': nbytes / (fit_ms * 1e-3) / 1e9, 'frac_of_hbm_peak': nbytes / (fit_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
			'traffic': linpsf_traffic(Nt, T, H)},
		'kernels': kernels,
	}
	if args.cpu_sample > 0:
		# CPU baseline: the oracle loop (reference-equivalent, scipy FITPACK integral per pixel) on a few targets, first cadences
		from oracle import linpsf as olin, psf as opsf
		ns, tsub = 4, min(T, 100)
		host = np.empty((ns, H, W, cubes['raw'].t_pitch), dtype='float32')
		ctx._check(ctx.lib.tp_memcpy_d2h(ctx.handle, host.ctypes.data, cubes['raw'].ptr, host.nbytes))
		bkg = work.bkg.slice0(0, ns).to_host()
		out = batch.out.to_host()
		t1 = time.perf_counter()
		bad = 0
		for i in range(ns):
			cat = scene.catalog_of(i)
			positions = np.empty((tsub, len(cat['starid']), 2))
			positions[:, :, 0] = cat['row_stamp'][None, :] + scene.jitter[:tsub, 1][:, None]
			positions[:, :, 1] = cat['column_stamp'][None, :] + scene.jitter[:tsub, 0][:, None]
			p = opsf.PSF(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'], tuple(scene.stamps[i]))
			p.integrate_to_image = p.integrate_to_image_scipy # literal reference loop (psf.py:136-146)
			img = host[i][:, :, :tsub] - bkg[i][None, None, :tsub]
			ref = olin.do_photometry(img, p, cat, scene.target_starid[i], positions, tuple(scene.stamps[i]),
				scene.target_pos_row[i], scene.target_pos_column[i], np.ones((H, W), dtype='int32'))
			bad += not np.allclose(out['flux'][i][:tsub], ref['flux'], rtol=1e-7, atol=1e-8*np.nanmax(np.abs(ref['flux'])))
		dt = time.perf_counter() - t1
		res['cpu_baseline'] = {'value': ns / (dt * T / tsub), 'unit': 'targets/s', 'cores': 1, 'kind': 'port',
			'sample': f'{ns} targets x first {tsub} cadences, extrapolated linearly to {T} cadences; oracle = literal per-pixel FITPACK loop of the reference',
			'calibration': 'dev-container timing of the reference\'s own LinPSFPhotometry.do_photometry beside this restatement on the same core '
				'(tests/golden/time_reference.py): 3.77 ms/cadence against 4.68 -- the port takes 1.24 x the reference\'s time, i.e. the '
				'reference itself would run about 1.24 x this rate'}
		res['parity_sample'] = {'targets': ns, 'cadences': tsub, 'mismatches': int(bad), 'rtol': 1e-7}
	return res


def tess_like_frames(np, n, R, C, seed):
	"""Synthetic full-frame images: smooth gradient + the corner glow the radial component models + noise + stars."""
	rng = np.random.default_rng(seed)
	yy, xx = np.mgrid[0:R, 0:C]
	r = np.hypot(xx + 44 - 31.0, yy - 2047.0)   # distance from the camera centre of (camera 1, CCD 1)-like geometry
	f = np.empty((n, R, C), dtype='float32')
	for k in range(n):
		img = 120 + 0.02 * xx + 40 * np.exp((r - 2400) / 250.0) + rng.normal(0, 4, r.shape)
		ys, xs = rng.integers(0, R, 400), rng.integers(0, C, 400)
		img[ys, xs] += rng.uniform(500, 60000, 400)
		f[k] = img
	return f


def leg_psf_fit(ctx, args, np, engine):
	"""PSFPhotometry.do_photometry (psf_photometry.py:111-196) for a batch -- per target and cadence a
	Nelder-Mead fit of (row, column, flux) of up to five stars, warm-started along the cadences (tp_psf_fit)."""
	from photometry_amd import simulate, psf as hpsf
	from photometry_amd.device import DeviceCube
	from photometry_amd.plugins import psf_star_selection, mag2flux
	Nt, T, H, W = args.psf_targets, 50, 15, 15
	s = simulate.make_scene(Nt, T, H, W, seed=args.seed * 1000 + 7)
	simulate.fill_cubes(s, nan_fraction=0.001)
	prf = simulate.synthetic_prf(seed=1)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	offs, params, mini = [0], [], []
	for i in range(Nt):
		c = s.catalog_of(i)
		sel = psf_star_selection(c['row_stamp'], c['column_stamp'], c['tmag'], s.target_pos_row[i] - s.stamps[i][0], s.target_pos_column[i] - s.stamps[i][2], s.target_tmag[i])
		params.append(np.column_stack((c['row_stamp'][sel].astype('float64'), c['column_stamp'][sel].astype('float64'), mag2flux(c['tmag'][sel].astype('float64')))))
		offs.append(offs[-1] + len(sel))
		# psf_photometry.py:29-41: the pixels within one pixel of the target position (all pixels collected here)
		jj, ii = np.meshgrid(np.arange(W), np.arange(H))
		mini.append(((np.abs(jj - (s.target_pos_column[i] - s.stamps[i][2])) <= 1) & (np.abs(ii - (s.target_pos_row[i] - s.stamps[i][0])) <= 1)).astype('uint8'))
	coef = engine.linpsf_prf(ctx, ctx.array(model.base_coef), ctx.array(model.weights(s.stamps)))
	a = (DeviceCube.from_host(ctx, s.images), DeviceCube.from_host(ctx, s.backgrounds), coef, ctx.array(model.tx), ctx.array(model.ty),
		ctx.array(np.asarray(offs, dtype='int64')), ctx.array(np.concatenate(params)), ctx.array(np.stack(mini)))
	engine.psf_fit(ctx, *a)
	ctx.sync()
	ctx.profile(True)
	ctx.profile_reset()
	t0 = time.perf_counter()
	res = engine.psf_fit(ctx, *a)
	ctx.sync()
	dt = time.perf_counter() - t0
	ctx.profile(False)
	kern = ctx.profile_report().get('tp_psf_fit_kernel', (1, dt * 1e3))
	kms = kern[1] / kern[0]
	nit = res['nit'].to_host().astype('float64')
	flux = res['flux'].to_host()
	nstars = offs[-1] / Nt
	# executed FP64 work of one simplex iteration (estimate): ~1.6 chi^2 evaluations x stars x 121 pixels of the cached set x
	# (24 Horner FMAs + ~6 for the weights / residual)
	fma = float(nit.sum()) * 1.6 * nstars * 121 * 30
	out = {'what': f'PSFPhotometry.do_photometry for {Nt} targets x {T} cadences x {H}x{W} ({offs[-1]} fitted stars): Nelder-Mead fit of (row, column, flux) '
		'per star and cadence, warm-started from the previous cadence, aperture correction (psf_photometry.py:111-196)',
		'kernel': 'tp_psf_fit_kernel', 'kernel_ms': kms, 'wall_ms': dt * 1e3, 'targets_per_s_at_50_cadences': Nt / dt,
		'targets_per_s_at_1300_cadences': Nt / dt * T / 1300.0, 'mean_simplex_iterations_per_cadence': float(nit.mean()),
		'ns_per_simplex_iteration_chipwide': kms * 1e6 / max(nit.sum(), 1.0), 'finite_fraction': float(np.mean(np.isfinite(flux))),
		'roofline': {'kernel': 'tp_psf_fit_kernel', 'bound': 'latency (a serial chain per target: one workgroup walks the simplex of one target; barriers, '
			'ordering, coefficient rebuilds) -- priced against the FP64 vector peak', 'achieved': 2 * fma / (kms * 1e-3) / 1e12, 'peak': FP64_VALU_TFLOPS,
			'unit': 'TFLOP/s', 'frac': 2 * fma / (kms * 1e-3) / 1e12 / FP64_VALU_TFLOPS, 'traffic': None,
			'flops': 'estimate of the executed FP64 FMAs (see bench.py:leg_psf_fit)'}}
	if args.cpu_sample > 0:
		from oracle import psf as opsf, psf_photometry as opp
		ns, tsub = 2, 3
		t1 = time.perf_counter()
		bad = 0
		for i in range(ns):
			p = opsf.PSF(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'], tuple(s.stamps[i]))
			ref = opp.do_photometry(s.images[i][:, :, :tsub], s.backgrounds[i][:, :, :tsub], p, s.catalog_of(i), tuple(s.stamps[i]),
				s.target_pos_row[i], s.target_pos_column[i], s.target_tmag[i], np.ones((H, W), dtype='int32'))
			ok = ref['success'] & np.isfinite(flux[i][:tsub])
			bad += int(np.sum(~np.isclose(flux[i][:tsub][ok], ref['flux'][ok], rtol=1e-5, atol=0)))
		dc = time.perf_counter() - t1
		out['cpu_baseline'] = {'value': ns / (dc * 1300.0 / tsub), 'unit': 'targets/s', 'cores': 1, 'kind': 'port',
			'sample': f'{ns} targets x first {tsub} cadences (scipy Nelder-Mead on the FITPACK pixel integral, like the reference), extrapolated linearly to 1300 cadences'}
		out['parity_sample'] = {'targets': ns, 'cadences': tsub, 'mismatches': bad, 'rtol': 1e-5}
	return out


def leg_fullframe(ctx, args, np):
	"""backgrounds.fit_background (backgrounds.py:52-211) on full 2048 x 2048 frames, plain and TESS
	branch, and the "background shenanigans" pixel-flag pass (pixel_flags.py:61-79, prepare.py:515-622)."""
	from photometry_amd import prepare
	nf, R, C = args.fullframe_frames, 2048, 2048
	f = tess_like_frames(np, nf, R, C, args.seed + 3)
	d = ctx.array(f)
	geo = prepare.RadialGeometry((R, C), 1, 1)
	out = {'what': f'{nf} frames of {R} x {C} float32 resident in HBM'}
	results = {}
	for name, kw in (('plain', {}), ('tess', dict(geometry=geo))):
		prepare.fit_background_frames(ctx, d, **kw).free()
		ctx.sync()
		ctx.profile(True)
		ctx.profile_reset()
		t0 = time.perf_counter()
		bkg = prepare.fit_background_frames(ctx, d, **kw)
		ctx.sync()
		dt = time.perf_counter() - t0
		ctx.profile(False)
		rep = ctx.profile_report()
		kms = sum(ms for _, ms in rep.values())
		results[name] = bkg.to_host()[0]
		bkg.free()
		passes = 1 if name == 'plain' else 3
		# necessary bytes per frame: the image read once per mesh pass (+ once per ring-mode pass), the background written once
		nb = R * C * 4 * (passes * (2 if name == 'tess' else 1) + 1)
		out[name] = {'wall_ms_per_frame': dt / nf * 1e3, 'kernel_ms_per_frame': kms / nf, 'frames_per_s': nf / dt,
			'kernels_ms_per_frame': {k: ms / nf for k, (_, ms) in rep.items()},
			'roofline': {'kernel': 'tp_bkg_mesh_kernel + tp_bkg_zoom_kernel' + (' + tp_radial_kernels' if name == 'tess' else ''), 'bound': 'hbm',
				'achieved': nb / (kms / nf * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': nb / (kms / nf * 1e-3) / 1e9 / HBM_PEAK_GBS,
				'bytes_per_frame': nb, 'bytes': 'R*C*4 per pass over the image (mesh statistics, ring modes) + the background written once', 'traffic': None}}
	# shenanigans: indicator (15 x 15 median filter of img - SumImage), its robust mean over time, thresholded flags
	ns = min(nf, 25)
	img = ctx.array(f[:ns])
	sumimage = ctx.array(f[:ns].astype('float64').mean(axis=0))
	flags = ctx.zeros((ns, R, C), 'uint8')
	prepare.background_shenanigans(ctx, img, sumimage, flags)
	ctx.sync()
	ctx.profile(True)
	ctx.profile_reset()
	t0 = time.perf_counter()
	prepare.background_shenanigans(ctx, img, sumimage, flags)
	ctx.sync()
	dt = time.perf_counter() - t0
	ctx.profile(False)
	rep = ctx.profile_report()
	kms = sum(ms for _, ms in rep.values())
	nb = R * C * (4 + 4 + 1)   # image read, indicator written (and read back for mean and threshold), flags written
	out['shenanigans'] = {'frames': ns, 'wall_ms_per_frame': dt / ns * 1e3, 'kernel_ms_per_frame': kms / ns,
		'kernels_ms_per_frame': {k: ms / ns for k, (_, ms) in rep.items()},
		'roofline': {'kernel': 'tp_median_filter_kernel', 'bound': 'vector ALU (a 225-key sorting network per pixel), priced against HBM',
			'achieved': nb / (kms / ns * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': nb / (kms / ns * 1e-3) / 1e9 / HBM_PEAK_GBS, 'traffic': None}}
	if args.cpu_sample > 0:
		from oracle import backgrounds as ob
		t1 = time.perf_counter()
		ref_plain = ob.fit_background(f[0])[0]
		t2 = time.perf_counter()
		ref_tess = ob.fit_background_tess(f[0], 1, 1, device_arithmetic=True)[0]
		t3 = time.perf_counter()
		sub = 512
		ob.pixel_background_shenanigans(f[0][:sub, :sub], f[:ns, :sub, :sub].astype('float64').mean(axis=0))
		t4 = time.perf_counter()
		out['plain']['cpu_baseline'] = {'value': 1.0 / (t2 - t1), 'unit': 'frames/s', 'cores': 1, 'kind': 'port', 'sample': 'one 2048 x 2048 frame'}
		out['tess']['cpu_baseline'] = {'value': 1.0 / (t3 - t2), 'unit': 'frames/s', 'cores': 1, 'kind': 'port', 'sample': 'one 2048 x 2048 frame, three rounds'}
		out['shenanigans']['cpu_baseline'] = {'value': 1.0 / ((t4 - t3) * (R * C) / (sub * sub)), 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
			'sample': f'the indicator image (scipy.ndimage.median_filter, size 15) of a {sub} x {sub} corner of one frame, scaled to 2048 x 2048'}
		with np.errstate(invalid='ignore', divide='ignore'):
			out['parity_sample'] = {'frames': 1, 'plain_max_rel_err': float(np.nanmax(np.abs(results['plain'] / ref_plain - 1))),
				'tess_max_rel_err': float(np.nanmax(np.abs(results['tess'] / ref_tess - 1))),
				'what': 'device background of frame 0 against the oracle (TESS branch: the oracle with the device\'s roundings written out)'}
	for a in (d, img, sumimage, flags):
		a.free()
	return out


if __name__ == '__main__':
	main()
