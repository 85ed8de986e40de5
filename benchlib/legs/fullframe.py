# -*- coding: utf-8 -*-
import time
from ..common import HBM_PEAK_GBS, TRAFFIC_FILE, committed_traffic


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


def leg_fullframe(ctx, args, np):
	"""backgrounds.fit_background (backgrounds.py:52-211) on full 2048 x 2048 frames, plain and TESS
	branch, and the "background shenanigans" pixel-flag pass (pixel_flags.py:61-79, prepare.py:515-622)."""
	from photometry_amd import prepare
	nf, R, C = args.fullframe_frames, 2048, 2048
	f = tess_like_frames(np, nf, R, C, args.seed + 3)
	d = ctx.array(f)
	geo = prepare.RadialGeometry((R, C), 1, 1)
	out = {'what': f'{nf} frames of {R} x {C} float32 resident in HBM'}
	# HBM bytes per frame from the committed PMC passes (profiles/run_profile.sh runs this leg on a few frames): the mean over a kernel's
	# dispatches divided by the frames of a dispatch; the mesh / zoom / radial kernels run once per round (3 in the TESS branch)
	tj = committed_traffic('traffic_bytes_per_launch', (R, C) == (2048, 2048)) or {}
	ff = committed_traffic('fullframe_frames', (R, C) == (2048, 2048)) or 0
	per_frame = {k: v / ff for k, v in tj.items() if ff and k.startswith(('tp_bkg_mesh', 'tp_mesh_finish', 'tp_bkg_zoom', 'tp_radial', 'tp_median_filter', 'tp_block_median', 'tp_threshold'))}
	b1 = committed_traffic('b1_traffic_bytes_per_frame', (R, C) == (2048, 2048)) or {}
	def leg_traffic(names, rounds, branch=None):
		if branch in b1:     # every dispatch of the branch summed (profiles/run_profile.sh, tools/radial_time.py MODE=branch)
			return b1[branch]
		got = [per_frame[k] for k in per_frame if k.startswith(names)]
		return rounds * sum(got) if got else None
	src = (TRAFFIC_FILE + ' (committed rocprofv3 PMC passes; not measured in this run)') if (per_frame or b1) else None
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
				'bytes_per_frame': nb, 'bytes': 'R*C*4 per pass over the image (mesh statistics, ring modes) + the background written once',
				'traffic': leg_traffic(('tp_bkg_mesh', 'tp_mesh_finish', 'tp_bkg_zoom') + (('tp_radial',) if name == 'tess' else ()), passes, name), 'traffic_unit': 'bytes per frame', 'traffic_source': src}}
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
		'roofline': {'kernel': 'tp_median_filter_kernel', 'bound': 'vector ALU (tp_median15_quad_kernel: per four adjacent pixels one sort of the 180 values their windows share + four sorts of 45 + a rank selection from the two sorted lists), priced against HBM',
			'achieved': nb / (kms / ns * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': nb / (kms / ns * 1e-3) / 1e9 / HBM_PEAK_GBS,
			'traffic': leg_traffic(('tp_median_filter', 'tp_block_median', 'tp_threshold'), 1), 'traffic_unit': 'bytes per frame', 'traffic_source': src}}
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
