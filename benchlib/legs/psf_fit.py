# -*- coding: utf-8 -*-
import time
from ..common import FP64_VALU_TFLOPS


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
