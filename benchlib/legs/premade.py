# -*- coding: utf-8 -*-
import time
from ..common import kernel_rows, roofline_of, committed_traffic


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
	traffic = committed_traffic('traffic_bytes_per_launch_premade', (Nt, T, H) == (10000, 1300, 15))
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
