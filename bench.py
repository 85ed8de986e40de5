#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""
bench.py -- whole-job throughput of the per-target photometry hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch of synthetic stamp cubes that is already
resident in HBM: A1 sum image -> A2..A5b K2P2 masks (+A7) -> A6 aperture extraction
(AperturePhotometry.do_photometry for every target).  Workload (BASELINE.json configs[2], the one
the metric is quoted on): 10 000 targets x 1 300 cadences x 15x15 stamps per GPU, aperture +
background cubes.  For N > 1 the driver launches one rank per GPU through torch.distributed.run;
targets are sharded by index (weak scaling: 10 000 targets per GPU), the only data-path exchange
is ONE RCCL gather of the light-curve block at the end of the timed region.

Prints ONE JSON line on rank 0 (see the bench contract in the task description) with the
``roofline`` and ``cpu_baseline`` objects.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0 # MI355X HBM3E peak (MI355X_MICROARCH.md)


def parse_args():
	p = argparse.ArgumentParser()
	p.add_argument('--gpus', type=int, default=1)
	p.add_argument('--steps', type=int, default=10)
	p.add_argument('--warmup', type=int, default=3)
	p.add_argument('--targets', type=int, default=10000, help='targets per GPU')
	p.add_argument('--cadences', type=int, default=1300)
	p.add_argument('--stamp', type=int, default=15)
	p.add_argument('--cpu-sample', type=int, default=384, help='targets in the CPU-baseline sample (0 = skip)')
	p.add_argument('--cpu-procs', type=int, default=16, help='worker processes of the CPU baseline')
	p.add_argument('--seed', type=int, default=1)
	p.add_argument('--no-gather', action='store_true')
	p.add_argument('--workload', choices=['aperture', 'linpsf'], default='aperture',
		help="'aperture' = BASELINE configs[2] (the headline); 'linpsf' = configs[3], the LinPSF fit on the same cube size")
	p.add_argument('--frame', type=int, default=1024, help='side of the synthetic full-frame stack of the stamp-cutter stage (0 = skip)')
	p.add_argument('--unfused', action='store_true', help='time the three stand-alone kernels (A1, K2P2, A6) back to back '
		'instead of the fused per-target kernel')
	return p.parse_args()


def _cpu_worker(job):
	"""
	Oracle (reference-equivalent numpy restatement, per-cadence Python loops like the reference) of one
	step on a list of targets: A1 sum image, K2P2 masks, A6 extraction, A7.  Returns (seconds, results).
	"""
	import numpy as np
	from oracle import sumimage as osum, aperture as oap
	sub = job
	t0 = time.perf_counter()
	out = []
	for i in range(sub.n_targets):
		S = osum.sumimage(sub.images[i], sub.quality)
		r = oap.do_photometry(S, sub.images[i], sub.images_err[i], sub.backgrounds[i], tuple(sub.stamps[i]),
			sub.target_pos_row[i], sub.target_pos_column[i], sub.target_tmag[i], sub.target_starid[i],
			sub.catalog_of(i), sub.aperture[i])
		out.append({k: r.get(k) for k in ('status', 'flux', 'flux_err', 'flux_background', 'pos_centroid', 'mask', 'contamination')})
	return time.perf_counter() - t0, out


_CPU_JOBS = None


def main_linpsf(args, ctx, rank, world, dist, torch, device_sync, barrier):
	"""BASELINE configs[3]: linpsf_photometry PSF-fit path over the same cube size (images cube resident)."""
	import numpy as np
	from photometry_amd import simulate, engine, pipeline, psf as hpsf
	Nt, T, H = args.targets, args.cadences, args.stamp
	W = H
	scene = simulate.make_scene(Nt, T, H, W, seed=args.seed * 1000 + rank)
	cubes = engine.synth_fill(ctx, scene, images=True, images_err=False, backgrounds=False)
	prf = simulate.synthetic_prf(seed=1) # synthetic stand-in for the SPOC PRF file (git-LFS object upstream)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	batch = pipeline.LinPSFBatch(ctx, scene, model, images=cubes['images'])
	for _ in range(args.warmup):
		pipeline.linpsf_step(ctx, batch)
	device_sync(); barrier()
	ctx.profile(True); ctx.profile_reset()
	t0 = time.perf_counter()
	for _ in range(args.steps):
		pipeline.linpsf_step(ctx, batch)
	device_sync(); barrier()
	elapsed = time.perf_counter() - t0
	ctx.profile(False)
	if dist is not None:
		t = torch.tensor([elapsed], dtype=torch.float64)
		dist.all_reduce(t, op=dist.ReduceOp.MAX)
		elapsed = float(t[0])
	prof = ctx.profile_report()
	if rank == 0:
		nfit = batch.n_fit_stars
		# ALGORITHMIC FP64 flops of the fit (the reference's direct contraction): per fitted star-cadence ~79 pixels inside
		# the 5 px cut-off x (169 + 13) multiply-adds.  The kernel's polynomial path executes about 8x fewer.
		flops = nfit * T * 79 * 182 * 2.0
		kernels = {name: {'launches': n, 'avg_ms': ms / n, 'ms_per_step': ms / args.steps} for name, (n, ms) in prof.items()}
		fit = kernels['tp_linpsf_fit_kernel']  # one launch per star-count class: the step's fit time is their sum
		fit['fp64_TFLOPs'] = flops / (fit['ms_per_step'] * 1e-3) / 1e12
		result = {
			'metric': 'targets/sec (whole node), 10k targets x 1300 cad x 15x15, linpsf_photometry PSF fit',
			'value': Nt * world * args.steps / elapsed, 'unit': 'targets/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
			'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
			'dtype': 'f64', 'data': 'synthetic',
			'config': {'workload': f'{Nt} targets/GPU x {T} cadences x {H}x{W} stamps, LinPSF fit of {nfit} stars (P1 table blend + P2-P4), '
				'image cube resident in HBM', 'fitted_stars': int(nfit)},
			'roofline': {'kernel': 'tp_linpsf_fit_kernel', 'bound': 'fp64-valu', 'achieved': fit['fp64_TFLOPs'], 'peak': 78.6, 'unit': 'TFLOP/s',
				'frac': fit['fp64_TFLOPs'] / 78.6, 'traffic': None, 'kernel_ms_per_step': fit['ms_per_step'],
				'note': 'achieved = algorithmic flops of the direct 13x13 contraction / time; the polynomial fast path does ~8x fewer'},
			'kernels': kernels,
		}
		if world == 1 and args.cpu_sample > 0:
			# CPU baseline: the oracle loop (reference-equivalent, scipy FITPACK integral per pixel) on a few targets
			from oracle import linpsf as olin, psf as opsf
			ns = min(Nt, max(1, args.cpu_sample // 96)) # 4 targets at the default --cpu-sample 384
			tsub = min(T, 100)
			host = np.empty((ns, H, W, cubes['images'].t_pitch), dtype='float32')
			ctx._check(ctx.lib.tp_memcpy_d2h(ctx.handle, host.ctypes.data, cubes['images'].ptr, host.nbytes))
			res = batch.out.to_host()
			t1 = time.perf_counter()
			bad = 0
			for i in range(ns):
				cat = scene.catalog_of(i)
				positions = np.empty((tsub, len(cat['starid']), 2))
				positions[:, :, 0] = cat['row_stamp'][None, :] + scene.jitter[:tsub, 1][:, None]
				positions[:, :, 1] = cat['column_stamp'][None, :] + scene.jitter[:tsub, 0][:, None]
				p = opsf.PSF(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'], tuple(scene.stamps[i]))
				p.integrate_to_image = p.integrate_to_image_scipy # literal reference loop (psf.py:136-146)
				ref = olin.do_photometry(host[i][:, :, :tsub], p, cat, scene.target_starid[i], positions, tuple(scene.stamps[i]),
					scene.target_pos_row[i], scene.target_pos_column[i], np.ones((H, W), dtype='int32'))
				bad += not np.allclose(res['flux'][i][:tsub], ref['flux'], rtol=1e-7, atol=1e-8*np.nanmax(np.abs(ref['flux'])))
			dt = time.perf_counter() - t1
			result['cpu_baseline'] = {'value': ns / (dt * T / tsub), 'unit': 'targets/s', 'cores': 1, 'kind': 'port',
				'sample': f'{ns} targets x first {tsub} cadences, extrapolated linearly to {T} cadences; oracle = literal per-pixel FITPACK loop of the reference'}
			result['parity_sample'] = {'targets': ns, 'cadences': tsub, 'mismatches': int(bad)}
		print(json.dumps(result))
	if dist is not None:
		dist.barrier()
		dist.destroy_process_group()
	ctx.close()


def _cpu_worker_indexed(c):
	return _cpu_worker(_CPU_JOBS[c])


def main():
	args = parse_args()
	rank = int(os.environ.get('RANK', '0'))
	local_rank = int(os.environ.get('LOCAL_RANK', '0'))
	if os.environ.get('TP_BENCH_FORCE_DEVICE'): # smoke-testing the multi-rank control flow on a 1-GPU box (use with --no-gather)
		local_rank = int(os.environ['TP_BENCH_FORCE_DEVICE'])
	world = int(os.environ.get('WORLD_SIZE', '1'))
	if world != args.gpus and world > 1:
		args.gpus = world

	# torch is plumbing only (rendezvous, barrier, device sync); import it BEFORE the HIP library
	# so that a single HIP runtime is shared by both.
	dist = None
	torch = None
	try:
		if os.environ.get('TP_BENCH_NO_TORCH') and world == 1:
			raise ImportError
		import torch
		if world > 1:
			import torch.distributed as dist
			os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
			dist.init_process_group(backend='gloo', rank=rank, world_size=world)
	except ImportError:
		torch = None

	import numpy as np
	from photometry_amd import simulate, engine, pipeline
	from photometry_amd.device import Context
	from photometry_amd import comm as tpcomm

	def device_sync():
		ctx.sync()
		if torch is not None and torch.cuda.is_available():
			torch.cuda.synchronize(local_rank)

	def barrier():
		if dist is not None:
			dist.barrier()

	ctx = Context(local_rank)
	if torch is not None and torch.cuda.is_available():
		torch.cuda.set_device(local_rank)

	Nt, T, H = args.targets, args.cadences, args.stamp
	W = H
	if args.workload == 'linpsf':
		return main_linpsf(args, ctx, rank, world, dist, torch, device_sync, barrier)
	# every rank gets its own contiguous shard of the global target list (weak scaling)
	scene = simulate.make_scene(Nt, T, H, W, seed=args.seed * 1000 + rank)
	scene.aperture = None
	# resident inputs: background-subtracted images, errors and the background cube
	# (3 x 11.7 GB at the default size) + the raw cube for the separately timed background stage
	cubes = engine.synth_fill(ctx, scene, raw=True)
	batch = pipeline.ApertureBatch(ctx, scene, cubes={k: cubes[k] for k in ('images', 'images_err', 'backgrounds')})
	work = pipeline.ApertureWork(ctx, batch)

	# The only data-path exchange: one RCCL gather of the light-curve block at the end of the timed region.  If the
	# communicator cannot be created the ranks agree (over gloo) to run without it and the JSON line says so: the hot
	# path itself has no collective.
	do_gather = world > 1 and not args.no_gather
	comm_note = None
	if do_gather:
		ok = 1
		try:
			tpcomm.init_from_torch(ctx, dist, rank, world)
		except Exception as e: # noqa: B902
			ok, comm_note = 0, f'RCCL communicator not created ({e}); light curves left on their ranks'
		t = torch.tensor([ok], dtype=torch.int32)
		dist.all_reduce(t, op=dist.ReduceOp.MIN)
		if int(t[0]) == 0:
			do_gather = False
			comm_note = comm_note or 'RCCL communicator not created on another rank; light curves left on their ranks'
	gather_buf = None
	lc_bytes = work.lc.block.nbytes
	if do_gather and rank == 0:
		gather_buf = ctx.empty((world, lc_bytes // 8), 'float64')

	def do_step():
		pipeline.aperture_step(ctx, batch, work, fused=not args.unfused)

	def profiling(on):
		ctx.profile(on)
		if on:
			ctx.profile_reset()

	for _ in range(args.warmup):
		do_step()
	device_sync()
	barrier()
	profiling(True)
	t0 = time.perf_counter()
	for _ in range(args.steps):
		do_step()
	if do_gather:
		tpcomm.gather(ctx, work.lc.block, gather_buf, root=0)
	device_sync()
	barrier()
	elapsed = time.perf_counter() - t0
	profiling(False)
	if dist is not None:
		t = torch.tensor([elapsed], dtype=torch.float64)
		dist.all_reduce(t, op=dist.ReduceOp.MAX)
		elapsed = float(t[0])

	prof = ctx.profile_report()

	# ---- the same arithmetic as three stand-alone kernels back to back (A1, K2P2, A6): per-stage durations when a
	# stage owns the GPU.  Not part of `value`.
	prof_serial = None
	if not args.unfused and rank == 0:
		pipeline.aperture_step(ctx, batch, work, fused=False)
		device_sync()
		ctx.profile(True)
		ctx.profile_reset()
		ts0 = time.perf_counter()
		for _ in range(2):
			pipeline.aperture_step(ctx, batch, work, fused=False)
		device_sync()
		serial_ms = (time.perf_counter() - ts0) / 2 * 1e3
		ctx.profile(False)
		prof_serial = ctx.profile_report()

	# ---- the stamp-level background stage (B*, B2, B3), timed the same way right after the headline region:
	# raw cube -> per-cadence sigma-clipped background series -> time smoothing -> subtraction (in place)
	ctx.profile(True)
	ctx.profile_reset()
	bkg_raw = ctx.zeros((Nt, cubes['raw'].t_pitch), 'float32')
	bkg_s = ctx.zeros((Nt, cubes['raw'].t_pitch), 'float32')
	device_sync()
	tb0 = time.perf_counter()
	nb = max(1, min(args.steps, 3))
	for _ in range(nb):
		engine.background_stamp(ctx, cubes['raw'], out=bkg_raw)
		engine.smooth_time(ctx, bkg_raw, T, batch.time_smooth, out=bkg_s)
	engine.subtract_background(ctx, cubes['raw'], bkg_s, images=cubes['raw'])
	device_sync()
	bkg_stage_ms = (time.perf_counter() - tb0) / nb * 1e3
	ctx.profile(False)
	prof_bkg = ctx.profile_report()

	# ---- the light-curve diagnostics of the batch, timed the same way; not part of `value`
	ctx.profile(True)
	ctx.profile_reset()
	pipeline.aperture_diagnostics(ctx, batch, work)
	device_sync()
	td0 = time.perf_counter()
	for _ in range(3):
		pipeline.aperture_diagnostics(ctx, batch, work)
	device_sync()
	diag_stage_ms = (time.perf_counter() - td0) / 3 * 1e3
	ctx.profile(False)
	prof_bkg.update(ctx.profile_report())

	# ---- the stamp cutter: the same 10k stamps cut from a 1024 x 1024 x T frame stack in HBM into the
	# (no longer needed) raw cube; timed the same way, not part of `value`
	cut_stage_ms = None
	if args.frame > 0:
		FR = args.frame
		frames = ctx.zeros((T, FR, FR), 'float32')
		rng = np.random.default_rng(args.seed)
		r0 = rng.integers(0, FR - H, Nt)
		c0 = rng.integers(0, FR - W, Nt)
		cstamps = ctx.array(np.stack((r0, r0 + H, c0 + 44, c0 + 44 + W), axis=1).astype('int32'))
		ctx.profile(True)
		ctx.profile_reset()
		engine.cut_stamps(ctx, frames, cstamps, H, W, 0, 44, out=cubes['raw'])
		device_sync()
		tc0 = time.perf_counter()
		for _ in range(3):
			engine.cut_stamps(ctx, frames, cstamps, H, W, 0, 44, out=cubes['raw'])
		device_sync()
		cut_stage_ms = (time.perf_counter() - tc0) / 3 * 1e3
		ctx.profile(False)
		prof_bkg.update(ctx.profile_report())
		frames.free()

	result = None
	if rank == 0:
		total_targets = Nt * world * args.steps
		value = total_targets / elapsed
		P = H * W
		# algorithmic bytes per target
		alg = {
			'tp_sumimage_kernel': P*T*4 + T*4 + P*8,             # A1
			'tp_aperture_kernel': 3*P*T*4 + P + 5*T*8,           # A6 (three cubes)
			# A1 + A6 in one kernel: the same reads and writes minus nothing (the sum image is still written, the mask too)
			'tp_aperture_fused_kernel': (P*T*4 + T*4 + P*8) + (3*P*T*4 + P + 5*T*8),
			'tp_bkg_stamp_kernel': P*T*4 + T*4,                  # B*
			'tp_bkg_smooth_kernel': 2*T*4,                       # B2
			'tp_bkg_subtract_kernel': 2*P*T*4 + T*4,             # B3 (materialised)
			'tp_cut_stamps_kernel': 2*P*T*4,                     # stamp cutter: read the stamp pixels, write the cube
		}
		def kernel_table(report, targets_per_launch):
			out = {}
			for name, (n, ms) in report.items():
				avg = ms / n
				k = {'launches': n, 'avg_ms': avg}
				if name in alg:
					k['algorithmic_bytes_per_launch'] = alg[name] * targets_per_launch
					k['achieved_GBps'] = alg[name] * targets_per_launch / (avg * 1e-3) / 1e9
				out[name] = k
			return out
		kernels = kernel_table(prof, Nt)
		kernels.update(kernel_table(prof_bkg, Nt))
		# the dominant kernel of the TIMED step (the background-stage kernels are reported in `kernels` only)
		dom = max((k for k in prof if True), key=lambda k: kernels[k]['avg_ms'])
		if dom not in alg: # a latency-bound kernel (K2P2) dominates: report the largest HBM-bound one and say so
			dom_hbm = max((k for k in prof if k in alg), key=lambda k: kernels[k]['avg_ms'])
		else:
			dom_hbm = dom
		traffic = None
		tfile = os.path.join(ROOT, 'profiles', 'r1_traffic.json')
		if os.path.exists(tfile) and (Nt, T, H) == (10000, 1300, 15):
			# HBM bytes per launch from rocprofv3 PMC passes of this same command (profiles/run_profile.sh),
			# FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE
			traffic = json.load(open(tfile)).get(dom_hbm)
		# bytes this implementation cannot avoid: A1 needs every pixel of the images cube, A6 only the rows of the
		# pixels that ended up in a mask (three cubes), plus the outputs -- the SURVEY 8d figure charges A6 with all rows
		n_mask = float(work.mask.to_host().astype('int64').sum())
		necessary = {
			'tp_sumimage_kernel': Nt * (P*T*4 + T*4 + P*8),
			'tp_aperture_kernel': 3 * n_mask * T * 4 + Nt * (P + 5*T*8),
		}
		necessary['tp_aperture_fused_kernel'] = necessary['tp_sumimage_kernel'] + necessary['tp_aperture_kernel']
		roofline = {
			'kernel': dom_hbm, 'bound': 'hbm', 'achieved': kernels[dom_hbm]['achieved_GBps'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
			'frac': kernels[dom_hbm]['achieved_GBps'] / HBM_PEAK_GBS, 'traffic': traffic,
			'avg_kernel_ms': kernels[dom_hbm]['avg_ms'], 'longest_kernel_of_step': dom,
			'targets_per_launch': Nt,
		}
		if dom_hbm in necessary:
			nb = necessary[dom_hbm] / (kernels[dom_hbm]['avg_ms'] * 1e-3) / 1e9
			roofline['necessary'] = {'what': 'bytes per launch the kernel cannot avoid (A1: whole images cube; A6: only the rows of the '
				'in-mask pixels of the three cubes; outputs) -- `achieved` uses the SURVEY 8d figure, which charges A6 with every row',
				'bytes_per_launch': necessary[dom_hbm], 'achieved': nb, 'frac': nb / HBM_PEAK_GBS, 'mean_mask_pixels': n_mask / Nt}
		result = {
			'metric': 'targets/sec (whole node), 10k targets x 1300 cad x 15x15, aperture + background',
			'value': value, 'unit': 'targets/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
			'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
			'dtype': 'f32', 'data': 'synthetic',
			'config': {'workload': f'{Nt} targets/GPU x {T} cadences x {H}x{W} stamps, aperture + background (BASELINE configs[2]): '
				'sum image + K2P2 masks + extraction of flux / error / centroid / background from the images, error and background '
				'cubes resident in HBM',
				'targets_per_gpu': Nt, 'cadences': T, 'stamp': [H, W], 'parallelism': f'targets sharded over {world} GPU(s)'},
			'roofline': roofline,
			'kernels': kernels,
			'gather': ('none (single GPU)' if world == 1 else ('one RCCL gather of the light-curve block, inside the timed region' if do_gather else (comm_note or 'disabled (--no-gather)'))),
			'background_stage': {'what': 'B* per-cadence stamp background + B2 time smoothing (+ one B3 subtraction) on the raw cube, '
				'timed right after the headline region; not part of `value`', 'ms_per_pass': bkg_stage_ms,
				'targets_per_s_including_it': Nt * world / (elapsed / args.steps + bkg_stage_ms * 1e-3)},
			'cutout_stage': {'what': f'stamp cutter: the {Nt} stamps cut from a {args.frame} x {args.frame} x {T} float32 frame stack resident in HBM '
				'(BasePhotometry._load_cube for the batch); one cube; not part of `value`', 'ms_per_cube': cut_stage_ms},
			'diagnostics_stage': {'what': 'light-curve diagnostics of every target (mean flux, variance, rms_hour, ptp, centroid, variability, '
				'mask size, edge flux: BasePhotometry.py:1343-1407) from the device-resident outputs; not part of `value`', 'ms_per_pass': diag_stage_ms},
		}

		if prof_serial is not None:
			result['three_kernel_path'] = kernel_table(prof_serial, Nt)
			result['three_kernel_path']['ms_per_step'] = serial_ms

	# ---- CPU baseline (rank 0, N = 1 only): the oracle on a bounded sample of the same cubes -----
	if rank == 0 and world == 1 and args.cpu_sample > 0:
		cores_avail = len(os.sched_getaffinity(0))
		nproc = max(1, min(cores_avail, args.cpu_procs))
		ns = min(Nt, max(args.cpu_sample, nproc * 24) // nproc * nproc)
		sub = scene.subset(slice(0, ns))
		for name in ('images', 'images_err', 'backgrounds'):
			cube = cubes[name]
			host = np.empty((ns, H, W, cube.t_pitch), dtype='float32')
			ctx._check(ctx.lib.tp_memcpy_d2h(ctx.handle, host.ctypes.data, cube.ptr, host.nbytes))
			setattr(sub, name, np.ascontiguousarray(host[..., :T]))
			del host
		sub.aperture = np.ones((ns, H, W), dtype='int32')
		# (a) one process, one core -- the analogue of one MPI worker of run_tessphot_mpi.py
		n1 = min(ns, 32)
		t1, _ = _cpu_worker(sub.subset(slice(0, n1)))
		# (b) nproc worker processes (forked: the sample is shared copy-on-write, nothing is pickled in);
		#     throughput = targets / slowest worker's compute time (process start-up excluded)
		import multiprocessing as mp
		global _CPU_JOBS
		_CPU_JOBS = [sub.subset(slice(c, ns, nproc)) for c in range(nproc)]
		with mp.get_context('fork').Pool(nproc) as pool:
			rr = pool.map(_cpu_worker_indexed, range(nproc))
		tmax = max(r[0] for r in rr)
		result['cpu_baseline'] = {
			'value': ns / tmax, 'unit': 'targets/s', 'cores': nproc, 'kind': 'port',
			'sample': f'{ns} of the {Nt} targets of the same device-generated cubes ({ns // nproc} per worker process, {nproc} processes '
				f'on a host with {cores_avail} usable cores); oracle = numpy restatement of the reference per-cadence loop '
				'(sum image + K2P2 + extraction); rate = targets / slowest worker compute time',
			'single_core_targets_per_s': n1 / t1,
			'host_cores_available': cores_avail,
		}
		result['speedup_vs_cpu_baseline'] = result['value'] / (ns / tmax)
		result['speedup_vs_one_core'] = result['value'] / (n1 / t1)
		# parity of the sample while we are here (masks / statuses / float32 sums bit-exact)
		lc = work.lc.to_host()
		masks = work.mask.to_host()
		status = work.status.to_host()
		bad = 0
		for c, (_, out) in enumerate(rr):
			for j, r in enumerate(out):
				i = c + j * nproc
				ok = int(status[i]) == r['status']
				if ok and r['mask'] is not None:
					ok = np.array_equal(masks[i].astype(bool), r['mask']) and np.array_equal(lc['flux'][i], r['flux'], equal_nan=True) \
						and np.array_equal(lc['flux_err'][i], r['flux_err'], equal_nan=True) \
						and np.array_equal(lc['flux_background'][i], r['flux_background'], equal_nan=True)
				bad += (not ok)
		result['parity_sample'] = {'targets': ns, 'mismatches': int(bad)}

	if rank == 0:
		print(json.dumps(result))
	if dist is not None:
		dist.barrier()
		dist.destroy_process_group()
	ctx.close()


if __name__ == '__main__':
	main()
