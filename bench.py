#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""
bench.py -- whole-job throughput of the per-target photometry hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch of synthetic stamp cubes already resident in HBM, for BASELINE.json
configs[2] ("aperture + background", the configuration the metric is quoted on): 10 000 targets x 1 300 cadences x 15x15
per GPU.  The resident inputs are the RAW flux cube and its error cube; one step reads the raw cube ONCE:

    B* + B2 + A1  per-cadence stamp background (sigma-clipped SExtractor mode), its time smoothing (prepare.py:317-335) and
        the sum image of raw - background (prepare.py:419-421, 450-459) in one pass        tp_background_sumimage
    A2..A5b + A7 + A6: the K2P2 mask and the extraction of AperturePhotometry.do_photometry for every target, the background
        subtracted on the fly (B3) and summed in the aperture; in-mask pixel rows only     tp_aperture_photometry_from_sumimage

N > 1: one process per GPU through photometry_amd.sharded (rank spawning, sharding, per-step double-buffered gather,
reassembly: the package's entry, not bench code).  Started as the driver starts it (torch.distributed.run: RANK / WORLD_SIZE
in the environment) or plainly as `python bench.py --gpus N`, in which case the ranks are spawned before anything touches the GPU.
Targets are sharded by index (weak scaling: 10 000 targets per GPU); the only data-path exchange is the gather of each
step's output block (light curves + contamination + status + flags + mask, one message per rank) to rank 0 over RCCL,
issued EVERY step on a second stream from the other half of a double-buffered output block, so that it overlaps the next
step's compute; its duration is reported separately.

The legs live in benchlib/ (one module per leg); this file parses the arguments, runs the timed region and assembles the line.
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
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)

from benchlib.common import HBM_PEAK_GBS, FP64_VALU_TFLOPS, XGMI_LINK_GBS, kernel_rows, roofline_of, committed_traffic  # noqa: E402


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
	p.add_argument('--no-bind', action='store_true', help="leave the CPU affinity alone (default: the process is bound to the cores of its GPU's NUMA node)")
	p.add_argument('--gather-when', choices=('auto', 'step', 'final'), default='final', help='N > 1: gather the output block once after the last step '
		"(default: north_star's final light-curve gather), every step under the next step's compute, or (auto) whichever the warm-up's measurement favours")
	p.add_argument('--no-compact', action='store_true', help='N > 1: gather the full float64 block instead of the compact one (flux, flux_err, flux_background as float32)')
	p.add_argument('--host-group', choices=('socket', 'gloo'), default='socket', help='N > 1: the host-side group (rendezvous, barriers, RCCL id): '
		'plain TCP sockets (no PyTorch) or torch.distributed gloo')
	p.add_argument('--no-extra', action='store_true', help='skip the extra legs (premade cubes, LinPSF, end to end, stages)')
	p.add_argument('--e2e-targets', type=int, default=2048, help='targets of the end-to-end (H2D included) leg (0 = skip)')
	p.add_argument('--frame', type=int, default=1024, help='side of the frame stack of the stamp-cutter stage (0 = skip)')
	p.add_argument('--psf-targets', type=int, default=4096, help='targets of the non-linear PSF photometry leg (0 = skip)')
	p.add_argument('--fullframe-frames', type=int, default=16, help='2048 x 2048 frames of the full-frame background / pixel-flag leg (0 = skip)')
	p.add_argument('--linpsf-drift', type=int, default=1, help='1: the LinPSF leg also runs the scene with a pointing drift (0 = skip: the profiling passes, whose per-kernel means it would mix)')
	p.add_argument('--frames-large', type=int, default=10000, help='targets of the second frames-to-results leg, on a 1024 x 1024 stack (0 = skip)')
	p.add_argument('--frames-targets', type=int, default=2500, help='targets of the frames-to-results leg on a 512 x 512 stack (0 = skip)')
	return p.parse_args(argv)



def main():
	args = parse_args()
	from photometry_amd import sharded
	if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
		sys.exit(sharded.spawn_ranks(__file__, sys.argv[1:], args.gpus))
	rank, local_rank, world = sharded.rank_environment()
	args.gpus = world
	# the host-side group of a multi-rank run (rendezvous, barrier, max over ranks, the RCCL id): plain TCP sockets by default --
	# no PyTorch is imported; --host-group gloo takes torch.distributed's gloo group instead (imported BEFORE the HIP library)
	group = sharded.init_host_group(rank, world, kind=args.host_group)

	import numpy as np
	from photometry_amd import simulate, engine, pipeline, _lib
	from photometry_amd.device import Context, DeviceCube
	import ctypes

	# ranks of one node use one GPU each; on a box with fewer GPUs than ranks they share devices (control-flow smoke run)
	ndev = ctypes.c_int(0)
	_lib.load().tp_device_count(ctypes.byref(ndev))
	if ndev.value <= 0:
		raise RuntimeError("bench.py needs a GPU: " + (_lib.load().tp_last_error(None) or b'').decode())
	shared_device = world > ndev.value
	device = local_rank % ndev.value
	from photometry_amd.device import bind_host_to_device
	all_cpus = os.sched_getaffinity(0)
	numa_node = None if args.no_bind else bind_host_to_device(device)   # this rank's threads on the cores next to its GPU
	ctx = Context(device)

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
	worker = sharded.DeviceShardWorker(ctx, scene, capacity=Nt, psf=psf, nbuf=2 if world > 1 else 1, extras=extras)
	run = sharded.ShardedRun(worker, Nt * world, rank=rank, world=world, group=group,
		gather='none' if args.no_gather else 'auto', when='step', shared_device=shared_device, compact=not args.no_compact)
	cubes, batch, lin = worker.cubes, worker.batch, worker.lin

	def device_sync():
		run.sync()

	# warm-up, with the gather of every step on the second stream: its measured duration beside the step without a gather
	# decides whether the timed region gathers every step (it hides under the next step) or once at the end
	run.run_steps(args.warmup, collect=True)
	step_alone_ms = None
	if run.gathers:
		if len(run.gather_ms) < 2:
			run.run_steps(2 * run.nbuf, collect=True)
		nalone = max(1, min(args.steps, 5))
		device_sync()
		run.barrier()
		t1 = time.perf_counter()
		for _ in range(nalone):
			worker.step(0)
		device_sync()
		step_alone_ms = run.max_over_ranks((time.perf_counter() - t1) / nalone * 1e3)
		if args.gather_when == 'auto':
			run.choose_when(step_alone_ms)
		else:
			run.when = args.gather_when
	per_step_gather_ms = list(run.gather_ms)
	del run.gather_ms[:]
	device_sync()
	run.barrier()
	ctx.profile(True)
	ctx.profile_reset()
	t0 = time.perf_counter()
	run.run_steps(args.steps, collect=True)
	device_sync()
	run.barrier()
	elapsed = run.max_over_ranks(time.perf_counter() - t0)
	ctx.profile(False)
	prof = ctx.profile_report()
	work = worker.works[run.last_buffer if run.last_buffer is not None else 0]
	per_step_gather_ms = run.gather_ms or per_step_gather_ms

	result = None
	if rank == 0:
		n_mask = float(work.mask.to_host().astype('int64').sum())
		# SURVEY 8d algorithmic bytes per target (A6 charged with every row of its cubes) ...
		alg = {
			'tp_bkg_stamp_sum_kernel': (P*T*4 + T*4) + 2*T*4 + (T*4 + P*8),
			'tp_aperture_fused_kernel': P*8 + (2*P*T*4 + T*4 + P + 5*T*8),
		}
		# ... and the bytes per launch the kernels cannot avoid: the background + sum-image pass needs the whole raw cube ONCE (and
		# writes two series and the sum image); the mask + extraction launch reads the sum image, the in-mask rows of two cubes and
		# the background series
		necessary = {
			'tp_bkg_stamp_sum_kernel': Nt * (P*T*4 + T*4 + 2*T*4 + P*8),
			'tp_aperture_fused_kernel': Nt * P*8 + 2 * n_mask * T * 4 + Nt * (T*4 + P + 5*T*8),
		}
		rows = kernel_rows(prof, Nt, alg, necessary)
		traffic = committed_traffic(default_size=(Nt, T, H) == (10000, 1300, 15))
		hbm_kernels = [k for k in rows if k in necessary]
		dom = max(hbm_kernels, key=lambda k: rows[k]['avg_ms'])
		nblocks = (T + 31) // 32
		notes = {
			'tp_bkg_stamp_sum_kernel': 'B* + B2 + A1 in one pass: streams the raw cube once, but is bound by the vector ALUs (a 256-key sorting network '
				'per frame for the sigma-clipped median), not by HBM.  Its instructions (v_min / v_max / v_med3, DPP moves, FP64) issue at one '
				'wave64 instruction per 4 cycles on gfx950 (tools/lab/valu_rate.hip): ~9 500 cycles per wavefront of 8 frames for B*, i.e. the '
				'vector-ALU time of the B* part of this launch is ~%.1f ms at 2.4 GHz on 1 024 SIMDs; the smoothing and the sum image add ~200 '
				'instructions per wavefront and block and one split-phase workgroup barrier per block' % (Nt * nblocks * 4 * 9500.0 / 1024 / 2.4e9 * 1e3),
			'tp_aperture_fused_kernel': 'the aperture-sum kernel north_star names, from the K2P2 mask on (the sum image comes from the background pass): '
				'one wavefront per target, in-mask pixel rows only; bound by the vector ALUs, not by HBM: 38 700 vector instructions per target '
				'(counters: profiles/r6_step_kernel_counters.txt), about half the mask builder (scipy bracket / Brent / Powell replayed, DBSCAN, '
				'watershed) and half the extraction, which takes the same time with its rows resident in L2 (DESIGN.md section 3)',
		}
		rooflines = [roofline_of(k, rows, traffic, notes.get(k)) for k in sorted(hbm_kernels, key=lambda k: -rows[k]['avg_ms'])]
		step_bytes = sum(necessary[k] for k in rows if k in necessary)
		n_total = Nt * world if not (psf and args.scaling == 'strong' and args.targets == 0) else min(100000, Nt * world)
		if psf:
			nfit = lin.n_fit_stars
			metric = 'targets/sec (whole node), 100k targets x 1300 cad x 15x15, aperture + PSF, target-sharded'
			wl = (f'{Nt} targets/GPU x {T} cadences x {H}x{W} stamps, aperture + PSF (BASELINE configs[4]: 100 000 targets over 8 GPUs = 12 500 per GPU): '
				'raw flux and error cubes resident in HBM; per step the stamp background of every cadence (B*), its time smoothing (B2) and the sum '
				'image in one pass over the raw cube, AperturePhotometry.do_photometry of every target with the background subtracted on the fly (B3), '
				f'and the LinPSF fit of every target (linpsf_photometry: P1 table blend + P2-P4, {nfit} fitted stars on this rank) on the same cube; '
				'light curves of both methods in the gathered block')
		else:
			metric = 'targets/sec (whole node), 10k targets x 1300 cad x 15x15, aperture + background'
			wl = (f'{Nt} targets/GPU x {T} cadences x {H}x{W} stamps, aperture + background (BASELINE configs[2]): raw flux and '
				'error cubes resident in HBM; per step the stamp background of every cadence (B*), its time smoothing (B2) and the sum image '
				'(A1) in ONE pass over the raw cube, then AperturePhotometry.do_photometry of every target from the mask on (K2P2 mask, '
				'extraction of flux / error / centroid / background) with the background subtracted on the fly (B3)')
		wl_short = (f'configs[4]: {Nt} targets/GPU x {T} cad x {H}x{W}, aperture + background + LinPSF' if psf
			else f'configs[2]: {Nt} targets x {T} cad x {H}x{W}, aperture + background')
		gms = per_step_gather_ms
		result = {
			'metric': metric,
			'value': n_total * args.steps / elapsed, 'unit': 'targets/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
			'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': args.scaling if psf else 'weak', 'vs_baseline': None,
			'dtype': 'f32 (aperture, background) + f64 (PSF fit)' if psf else 'f32', 'data': 'synthetic',
			'config': {'workload': wl, 'workload_short': wl_short, 'baseline_config': 'configs[4]' if psf else 'configs[2]',
				'targets_per_gpu': Nt, 'targets_total': n_total, 'cadences': T, 'stamp': [H, W], 'host_numa_node': numa_node,
				'parallelism': f'targets sharded over {world} GPU(s), one process per GPU (photometry_amd.sharded), no data-path collective but the gather of the output block',
				'parallelism_short': f'targets sharded over {world} GPU(s), 1 process/GPU, output gather only'},
			'roofline': next(r for r in rooflines if r['kernel'] == dom),
			'rooflines': rooflines,
			'step_hbm': {'necessary_bytes_per_step': step_bytes, 'GBps_over_whole_step': step_bytes / (elapsed / args.steps) / 1e9,
				'frac_of_hbm_peak': step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 'mean_mask_pixels': n_mask / Nt,
				'note': 'the raw cube is read once per step (rounds 1-3: twice -- the background kernel, then the sum-image phase of the fused kernel)'},
			'kernels': rows,
			'gather': {'mode': run.mode, 'when': run.when if run.gathers else None, 'bytes_per_rank_per_step': run.send_nbytes if world > 1 else 0,
				'bytes_per_rank_full_block': worker.block_nbytes if world > 1 else 0, 'compact': bool(run.compact),
				'block': 'light curves [5][Nt][T] f64 + contamination f64 + status, flags i32 + mask u8 per target' + (' + LinPSF light curve [Nt][T] f64, contamination f64, status i32' if psf else '') + ', one message per rank (comm.packed_block_layout); sent compact: flux, flux_err, flux_background as the float32 values they are (comm.compact_block_layout)',
				'issued': ('every step, second stream, double-buffered output block' if run.when == 'step' else 'once, after the last step (inside the timed region)') if run.gathers else None,
				'issued_short': run.when if run.gathers else None,
				'mean_ms': (sum(gms) / len(gms)) if gms else None,
				'mean_ms_from': ('the timed region' if run.gather_ms else 'the warm-up steps (per-step gather, before the timed region chose "final")') if gms else None,
				'final_ms': (sum(run.final_ms) / len(run.final_ms)) if run.final_ms else None,
				'step_ms_without_gather': step_alone_ms,
				'ideal_ms_one_xgmi_link': run.send_nbytes / (XGMI_LINK_GBS * 1e9) * 1e3 if world > 1 else None,
				'measured_8gpu': False,
				'measured_on': 'never on an 8-GPU node so far (this pool gives one GPU per call): the N > 1 lines come from the driver\'s node, if it has one; no 8-GPU scaling curve exists',
				'host_group': type(group).__name__,
				'xgmi_rate_assumed': f'{XGMI_LINK_GBS} GB/s one way per link: the root receives from its N - 1 peers on N - 1 links at once (direct '
					'send / recv pairs in one RCCL group), so the gather is bound by ONE inbound link per peer; if the 153 GB/s of the guide is the '
					'bidirectional figure the ideal doubles -- mean_ms beside step_ms_without_gather is the measurement that decides'},
		}
		if psf:
			# the LinPSF fit is the longest part of the configs[4] step: its own roofline (FP64 pipe, algorithmic flops).  The fit launches of
			# the star counts overlap (side streams), so their kernel times do not add up: the fit's share of the step is the step's wall
			# time less the kernels that run alone
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
		from benchlib.legs.premade import leg_premade
		from benchlib.legs.stages import leg_stages
		result['aperture_premade_cubes'] = leg_premade(ctx, scene, cubes, args, Nt, T, H, W, np, engine, pipeline)
		result['stages'] = leg_stages(ctx, scene, cubes, batch, work, args, Nt, T, H, W, np, engine, pipeline)
	if rank == 0 and world == 1 and workload == 'c2' and args.cpu_sample > 0:
		from benchlib.cpu import cpu_baseline
		bound = os.sched_getaffinity(0)
		os.sched_setaffinity(0, all_cpus)     # the CPU baseline gets every core the box gives, not only the GPU's NUMA node
		try:
			cb, parity = cpu_baseline(ctx, scene, cubes, work, args, T, H, W, batch.time_smooth)
		finally:
			os.sched_setaffinity(0, bound)
		result['cpu_baseline'] = cb
		result['parity_sample'] = parity
		result['speedup_vs_cpu_baseline'] = result['value'] / cb['value']
		result['speedup_vs_one_core'] = result['value'] / cb['single_core_targets_per_s']
	if rank == 0 and extras:
		from benchlib.legs.end_to_end import leg_end_to_end
		from benchlib.legs.linpsf import leg_linpsf
		from benchlib.legs.frames import leg_frames, leg_psf_frames
		from benchlib.legs.psf_fit import leg_psf_fit
		from benchlib.legs.fullframe import leg_fullframe
		if args.e2e_targets > 0:
			result['end_to_end'] = leg_end_to_end(ctx, scene, cubes, args, T, H, W, np, engine, pipeline, Context, DeviceCube)
		for k in ('images', 'backgrounds'):
			cubes[k].free()
		result['linpsf'] = leg_linpsf(ctx, scene, cubes, work, args, Nt, T, H, W, np, engine, pipeline)
		if args.frames_targets > 0 and (T, H) == (1300, 15):
			result['frames_to_results'] = leg_frames(ctx, args, T, np, pipeline)
			result['psf_frames_to_results'] = leg_psf_frames(ctx, args, T, np, pipeline)
		for k in ('raw', 'images_err'):
			cubes[k].free()
		if args.psf_targets > 0:
			result['psf_fit'] = leg_psf_fit(ctx, args, np, engine)
		if args.fullframe_frames > 0:
			result['fit_background_frames'] = leg_fullframe(ctx, args, np)
		if args.frames_targets > 0 and args.frames_large > 0 and (T, H) == (1300, 15):
			# the batched entry once more, on a batch of the headline's size: four times the region (same star density), four times the
			# targets.  Last: four such jobs in flight leave ~100 GB in the engine's allocation caches, and a leg that sizes a buffer by
			# the free memory (PSFPhotometry's coefficient store) ran a quarter slower behind it
			result['frames_to_results_large_batch'] = leg_frames(ctx, args, T, np, pipeline, N=args.frames_large, FR=1024, NB=6, runs=7)

	if rank == 0:
		# the full result (every leg with its notes) goes to bench_legs.json and stderr; stdout gets ONE short, self-checked line
		from benchlib.line import emit
		line = emit(result, ROOT, stream=sys.stderr)
		sys.stdout.flush()
		print(line)
		sys.stdout.flush()
	group.barrier()
	run.close()
	ctx.close()
	group.close()


if __name__ == '__main__':
	main()
