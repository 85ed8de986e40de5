# -*- coding: utf-8 -*-
from ..common import HBM_PEAK_GBS


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
