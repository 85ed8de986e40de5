# -*- coding: utf-8 -*-
import time
from ..common import HBM_PEAK_GBS, FP64_VALU_TFLOPS, TRAFFIC_FILE, committed_traffic


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
	by_path = engine.linpsf_last_counts(ctx)
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
	# v_mfma_f64_16x16x4_f64 per step on this batch = 1.9e11 flops against 1.3e11 algorithmic)
	fma = nfit * T * 79 * 24 + float(np.sum(counts * (counts + 1) / 2 + counts)) * T * H * W + nfit * 3 * 79 * 1170
	flops = 2.0 * fma
	nbytes = Nt * (H*W*T*4 + T*4) + nfit * T * 16 + Nt * T * 8
	traffic = (committed_traffic(default_size=(Nt, T, H) == (10000, 1300, 15)) or {}).get('tp_linpsf_fit')
	res = {
		'metric': 'targets/sec, 10k targets x 1300 cad x 15x15, linpsf_photometry PSF fit (BASELINE configs[3])',
		'value': Nt / (ms * 1e-3), 'unit': 'targets/s', 'ms_per_step': ms, 'steps': n, 'dtype': 'f64', 'fitted_stars': int(nfit),
		'config': {'workload': f'{Nt} targets x {T} cadences x {H}x{W}, LinPSF fit of {nfit} stars (P1 table blend + P2-P4), raw cube resident, '
			'background series subtracted on the fly'},
		'roofline': {'kernel': 'tp_linpsf_plan_kernel + tp_linpsf_coef_kernel + tp_linpsf_fitm_kernel (matrix-core fit; tp_linpsf_fit_kernel = the vector-ALU fit of the targets that do not qualify)',
			'bound': 'fp64 pipe (not HBM): FP64 matrix and vector instructions share one pipe on this chip and have the same peak',
			'achieved': flops / (fit_ms * 1e-3) / 1e12, 'peak': FP64_VALU_TFLOPS, 'unit': 'TFLOP/s', 'frac': flops / (fit_ms * 1e-3) / 1e12 / FP64_VALU_TFLOPS,
			'flops': 'algorithmic FP64 flops of the path (estimate, see benchlib/legs/linpsf.py); the matrix-core fit executes ~1.7 x that.  SURVEY 8d\'s banded figure '
				'(0.19 GFLOP per target: FITPACK\'s 13 x 13 contraction per star, pixel and cadence) is NOT a floor for this formulation -- the quartic-spline form '
				'evaluates 25-49 basis products per star and pixel instead -- and would give a fraction above 1; it is not used', 'kernel_ms_per_step': fit_ms, 'kernel_ms_note': 'wall time of the step less the kernels that run alone: the fit launches of the star counts overlap',
			'hbm': {'necessary_bytes_per_step': nbytes, 'GBps': nbytes / (fit_ms * 1e-3) / 1e9, 'frac_of_hbm_peak': nbytes / (fit_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
			'traffic': traffic, 'traffic_source': (TRAFFIC_FILE + ' (committed rocprofv3 PMC passes; not measured in this run)') if traffic is not None else None},
		'kernels': kernels,
		'targets_by_path': by_path,
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
	if getattr(args, 'linpsf_drift', 1):
		res['drift'] = leg_linpsf_drift(ctx, scene, model, prf, work, args, Nt, T, H, W, np, engine, pipeline, ms)
	return res


def leg_linpsf_drift(ctx, scene, model, prf, work, args, Nt, T, H, W, np, engine, pipeline, ms_no_drift):
	"""The same fit on a sector with a pointing drift: every star moves by half a pixel (column) / 0.35 px (row) linearly over the
	series on top of the jitter -- 4.5 knot intervals of the PRF grid, where round 3 sent the target to the vector-ALU kernels."""
	import copy
	sd = copy.copy(scene)
	ramp = np.linspace(0.0, 1.0, T)
	sd.jitter = scene.jitter + np.stack((0.5 * ramp, -0.35 * ramp), axis=1)
	raw = engine.synth_fill(ctx, sd, images=False, images_err=False, backgrounds=False, raw=True)['raw']
	batch = pipeline.LinPSFBatch(ctx, sd, model, images=raw, subtract=work.bkg)
	pipeline.linpsf_step(ctx, batch)
	ctx.sync()
	n = max(3, min(args.steps, 5))
	ctx.profile(True)
	ctx.profile_reset()
	t0 = time.perf_counter()
	for _ in range(n):
		pipeline.linpsf_step(ctx, batch)
	ctx.sync()
	ms = (time.perf_counter() - t0) / n * 1e3
	ctx.profile(False)
	kernels = {name: {'launches': c, 'ms_per_step': t / n} for name, (c, t) in ctx.profile_report().items()}
	# the same scene on the vector-ALU kernels (where round 3 sent every target whose stars leave their three knot intervals)
	engine.linpsf_set_path(ctx, 0)
	try:
		pipeline.linpsf_step(ctx, batch)
		ctx.sync()
		t1 = time.perf_counter()
		for _ in range(2):
			pipeline.linpsf_step(ctx, batch)
		ctx.sync()
		ms_valu = (time.perf_counter() - t1) / 2 * 1e3
	finally:
		engine.linpsf_set_path(ctx, 1)
	pipeline.linpsf_step(ctx, batch)   # (the results compared below are the matrix-core ones)
	ctx.sync()
	out = {'what': 'linear drift of 0.5 px (column) and -0.35 px (row) over the series added to every star\'s positions (and to the synthetic cube), jitter as before',
		'value': Nt / (ms * 1e-3), 'unit': 'targets/s', 'ms_per_step': ms, 'ratio_to_no_drift': ms / ms_no_drift, 'ms_per_step_on_the_vector_alu_kernels': ms_valu, 'targets_by_path': engine.linpsf_last_counts(ctx), 'kernels': kernels}
	if args.cpu_sample > 0:
		from oracle import linpsf as olin, psf as opsf
		ns, tsub = 2, min(T, 60)
		host = np.empty((ns, H, W, raw.t_pitch), dtype='float32')
		ctx._check(ctx.lib.tp_memcpy_d2h(ctx.handle, host.ctypes.data, raw.ptr, host.nbytes))
		bkg = work.bkg.slice0(0, ns).to_host()
		res = batch.out.to_host()
		bad = 0
		for i in range(ns):
			cat = sd.catalog_of(i)
			# the last cadences: where the drift has carried the stars farthest from the first segment's knot intervals
			ks = np.arange(T - tsub, T)
			positions = np.empty((tsub, len(cat['starid']), 2))
			positions[:, :, 0] = cat['row_stamp'][None, :] + sd.jitter[ks, 1][:, None]
			positions[:, :, 1] = cat['column_stamp'][None, :] + sd.jitter[ks, 0][:, None]
			p = opsf.PSF(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'], tuple(sd.stamps[i]))
			img = host[i][:, :, ks] - bkg[i][None, None, ks]
			ref = olin.do_photometry(img, p, cat, sd.target_starid[i], positions, tuple(sd.stamps[i]), sd.target_pos_row[i], sd.target_pos_column[i],
				np.ones((H, W), dtype='int32'))
			bad += not np.allclose(res['flux'][i][ks], ref['flux'], rtol=1e-7, atol=1e-8 * np.nanmax(np.abs(ref['flux'])))
		out['parity_sample'] = {'targets': ns, 'cadences': tsub, 'which': 'the last cadences of the series', 'mismatches': int(bad), 'rtol': 1e-7}
	raw.free()
	return out
