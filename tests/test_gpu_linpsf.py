# -*- coding: utf-8 -*-
"""
GPU parity of the LinPSF path (P1 PRF blend, P2 pixel-integrated PRF, P3 least squares, P4 loop)
against the oracle and against the golden vectors produced by the reference's own
``LinPSFPhotometry.do_photometry``.  Float64 throughout; north_star tolerance on flux is 1e-5
relative, asserted here at 1e-8.
"""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
	from photometry_amd.device import Context
	c = Context(0)
	yield c
	c.close()


def _positions(s, sel, T):
	"""catalog_attime stand-in: reference position + per-cadence jitter (BasePhotometry.py:1224-1258)."""
	rs = s.catalog['row_stamp'][sel].astype('float64')
	cs = s.catalog['column_stamp'][sel].astype('float64')
	pos_row = rs[:, None] + s.jitter[None, :, 1]
	pos_col = cs[:, None] + s.jitter[None, :, 0]
	return np.ascontiguousarray(pos_row), np.ascontiguousarray(pos_col)


def test_prf_blend_matches_reference_construction(ctx):
	"""P1: device blend of per-sample spline tables == spline of the blended PRF (psf.py:101-119)."""
	from photometry_amd import engine, psf as hpsf
	from oracle import psf as opsf
	prf = opsf.synthetic_prf(seed=3)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	stamps = np.array([[100, 115, 300, 315], [0, 15, 44, 59], [2030, 2045, 2070, 2085], [1000, 1011, 1200, 1213]])
	w = model.weights(stamps)
	coef = engine.linpsf_prf(ctx, ctx.array(model.base_coef), ctx.array(w)).to_host()
	for i, st in enumerate(stamps):
		ref = opsf.PSF(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'], tuple(st))
		np.testing.assert_allclose(coef[i].reshape(model.n, model.n), ref.coeffs, rtol=1e-9, atol=1e-13*np.abs(ref.coeffs).max())
		np.testing.assert_array_equal(model.tx, ref.tx)


# jitter scale 1: sigma 0.02 px (a few table origins per star: the polynomial path, cadences sorted by origin);
# 6: sigma 0.12 px (dozens of origins per star, several per wavefront even after sorting); 30: sigma 0.6 px (more origins
# than the plan allows: the target is flagged and redone by the general direct kernel); 1300+ cadences: several workgroups per
# target; 8300 cadences: beyond the LDS sort of the plan kernel (natural cadence order)
@pytest.mark.parametrize("max_neigh,T,H,W,jit", [(3, 40, 11, 11, 1), (1, 70, 15, 15, 1), (6, 16, 13, 12, 1),
	(3, 40, 11, 11, 6), (2, 33, 15, 15, 30), (1, 1301, 9, 9, 1), (3, 1100, 9, 9, 2), (1, 8300, 7, 7, 1),
	(13, 12, 11, 11, 1)])     # more than 8 fitted stars: the run-time sized kernel
@pytest.mark.parametrize("path", [0, 1])   # 0: vector-ALU kernels for every target; 1: matrix-core fit where the target qualifies
def test_linpsf_matches_oracle(ctx, path, max_neigh, T, H, W, jit):
	from photometry_amd import simulate, engine, psf as hpsf
	from photometry_amd.device import DeviceCube
	from oracle import psf as opsf, linpsf as olin
	nt = 6 if T < 1000 else 2
	s = simulate.make_scene(nt, T, H, W, seed=50 + max_neigh, max_neighbours=max_neigh, neighbour_tmag_range=(9.0, 17.0))
	s.jitter = s.jitter * jit
	simulate.fill_cubes(s, nan_fraction=0.01)
	s.images[0, :, :, 3] = np.nan       # a frame without a single good pixel -> flux 0 (pinv of a zero matrix)
	prf = opsf.synthetic_prf(seed=7)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	sel, star_offsets, target_index = hpsf.select_stars(s.catalog, s.cat_offsets, s.target_starid)
	pos_row, pos_col = _positions(s, sel, T)
	max_stars = int(np.diff(star_offsets).max())
	assert max_stars > 8 or max_neigh < 13
	coef = engine.linpsf_prf(ctx, ctx.array(model.base_coef), ctx.array(model.weights(s.stamps)))
	engine.linpsf_set_path(ctx, path)
	try:
		res = engine.linpsf_fit(ctx, DeviceCube.from_host(ctx, s.images), coef, ctx.array(model.tx), ctx.array(model.ty),
			ctx.array(star_offsets), ctx.array(target_index), ctx.array(pos_row), ctx.array(pos_col), max_stars).to_host()
	finally:
		engine.linpsf_set_path(ctx, 1)
	for i in range(s.n_targets):
		cat = s.catalog_of(i)
		ncat = len(cat['starid'])
		positions = np.empty((T, ncat, 2))
		positions[:, :, 0] = cat['row_stamp'][None, :] + s.jitter[:, 1][:, None]
		positions[:, :, 1] = cat['column_stamp'][None, :] + s.jitter[:, 0][:, None]
		p = opsf.PSF(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'], tuple(s.stamps[i]))
		ref = olin.do_photometry(s.images[i], p, cat, s.target_starid[i], positions, tuple(s.stamps[i]),
			s.target_pos_row[i], s.target_pos_column[i], s.aperture[i])
		assert ref['nstars'] == star_offsets[i+1] - star_offsets[i] and ref['staridx'] == target_index[i]
		scale = np.nanmax(np.abs(ref['flux']))
		np.testing.assert_allclose(res['flux'][i], ref['flux'], rtol=1e-8, atol=1e-9*scale)
		assert np.all(np.isnan(res['flux_err'][i]))
		assert int(res['status'][i]) == ref['status']
		np.testing.assert_allclose(res['contamination'][i], ref['contamination'], rtol=1e-7, atol=1e-11)
		np.testing.assert_allclose(res['fluxes_mean'][star_offsets[i]:star_offsets[i+1]], ref['fluxes_mean'], rtol=1e-8, atol=1e-9*scale)
	assert res['flux'][0, 3] == 0.0


@pytest.mark.parametrize("path", [0, 1])
def test_linpsf_golden(ctx, golden_dir, path):
	"""The reference's own LinPSFPhotometry.do_photometry (golden) through the device kernels (both mappings of the fit)."""
	from photometry_amd import engine, psf as hpsf
	from photometry_amd.device import DeviceCube
	from scipy.interpolate import RectBivariateSpline
	g = np.load(os.path.join(golden_dir, 'golden_linpsf.npz'))
	x = g['prf_x']
	model = hpsf.PRFModel.from_spline(RectBivariateSpline(x, x, g['prf_img']))
	Nt, H, W, T = g['images'].shape
	catalog = {k[4:]: g[k] for k in g.files if k.startswith('cat_') and k != 'cat_offsets'}
	sel, star_offsets, target_index = hpsf.select_stars(catalog, g['cat_offsets'], g['target_starid'])
	pos_row = np.concatenate([g[f'lp{i}_positions'][:, :, 0].T for i in range(Nt)])[sel]
	pos_col = np.concatenate([g[f'lp{i}_positions'][:, :, 1].T for i in range(Nt)])[sel]
	coef = engine.linpsf_prf(ctx, ctx.array(model.base_coef), ctx.array(model.weights(g['stamps'])))
	engine.linpsf_set_path(ctx, path)
	try:
		res = engine.linpsf_fit(ctx, DeviceCube.from_host(ctx, g['images']), coef, ctx.array(model.tx), ctx.array(model.ty),
			ctx.array(star_offsets), ctx.array(target_index), ctx.array(np.ascontiguousarray(pos_row)), ctx.array(np.ascontiguousarray(pos_col)),
			int(np.diff(star_offsets).max())).to_host()
	finally:
		engine.linpsf_set_path(ctx, 1)
	for i in range(int(g['n_linpsf'])):
		ref = g[f'lp{i}_flux']
		np.testing.assert_allclose(res['flux'][i], ref, rtol=1e-8, atol=1e-9*np.abs(ref).max())
		assert int(res['status'][i]) == int(g[f'lp{i}_status'])
		np.testing.assert_allclose(res['contamination'][i], float(g[f'lp{i}_contamination']), rtol=1e-7, atol=1e-11)


def test_linpsf_argument_checks(ctx):
	from photometry_amd import engine
	from photometry_amd.device import DeviceCube
	from photometry_amd._lib import TessphotError
	img = DeviceCube(ctx, 1, 8, 5, 5)
	z = ctx.zeros((200,), 'float64')
	for bad in (0.0, -1.0, float('nan')):
		with pytest.raises(TessphotError) as e:
			engine.linpsf_fit(ctx, img, z, ctx.zeros((121,), 'float64'), ctx.zeros((121,), 'float64'), ctx.zeros((2,), 'int64'),
				ctx.zeros((1,), 'int32'), ctx.zeros((1, 8), 'float64'), ctx.zeros((1, 8), 'float64'), 1, cutoff_radius=bad)
		assert 'cutoff_radius' in str(e.value)


# include/tessphot_hip.h (tp_linpsf_fit): grids other than the SPOC layout and cut-off radii beyond its evenly spaced knots -- or none
# at all (psf.py:142 ``cutoff_radius is None``) -- are fitted by the general kernels with the FITPACK box integral itself
@pytest.mark.parametrize("kind,cutoff,max_neigh", [('warped', 5, 3), ('nsub7', 5, 2), ('spoc', 7.5, 3), ('spoc', None, 2), ('coarse', None, 3), ('rect', 5, 3), ('rect', None, 2),
	('warped', 6.0, 13), ('spoc', 5.3, 1), ('nsub7', 6.5, -2)])      # (-2: two neighbours at most, 300 cadences: several workgroups per target)
def test_linpsf_any_grid_any_cutoff(ctx, kind, cutoff, max_neigh):
	from photometry_amd import simulate, engine, psf as hpsf
	from photometry_amd.device import DeviceCube
	from oracle import psf as opsf, linpsf as olin
	nt, T, H, W = 5, 37, 13, 12
	if max_neigh < 0:
		nt, T, max_neigh = 3, 300, -max_neigh
	s = simulate.make_scene(nt, T, H, W, seed=70 + max_neigh, max_neighbours=max_neigh, neighbour_tmag_range=(9.0, 17.0))
	s.jitter = s.jitter * 3
	simulate.fill_cubes(s, nan_fraction=0.01)
	from prf_common import general_prf
	prf = general_prf(kind)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	sel, star_offsets, target_index = hpsf.select_stars(s.catalog, s.cat_offsets, s.target_starid)
	pos_row, pos_col = _positions(s, sel, T)
	max_stars = int(np.diff(star_offsets).max())
	coef = engine.linpsf_prf(ctx, ctx.array(model.base_coef), ctx.array(model.weights(s.stamps)))
	res = engine.linpsf_fit(ctx, DeviceCube.from_host(ctx, s.images), coef, ctx.array(model.tx), ctx.array(model.ty),
		ctx.array(star_offsets), ctx.array(target_index), ctx.array(pos_row), ctx.array(pos_col), max_stars, cutoff_radius=cutoff).to_host()
	counts = engine.linpsf_last_counts(ctx)
	assert counts['any_grid_targets'] == nt      # every target went through the general kernels ...
	assert counts['matrix_core_targets'] + counts['vector_alu_polynomial_targets'] + counts['vector_alu_general_targets'] + counts['many_star_targets'] == 0
	for i in range(s.n_targets):
		cat = s.catalog_of(i)
		ncat = len(cat['starid'])
		positions = np.empty((T, ncat, 2))
		positions[:, :, 0] = cat['row_stamp'][None, :] + s.jitter[:, 1][:, None]
		positions[:, :, 1] = cat['column_stamp'][None, :] + s.jitter[:, 0][:, None]
		p = opsf.PSF(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'], tuple(s.stamps[i]))
		ref = olin.do_photometry(s.images[i], p, cat, s.target_starid[i], positions, tuple(s.stamps[i]),
			s.target_pos_row[i], s.target_pos_column[i], s.aperture[i], cutoff_radius=cutoff)
		scale = np.nanmax(np.abs(ref['flux']))
		np.testing.assert_allclose(res['flux'][i], ref['flux'], rtol=1e-8, atol=1e-9*scale)
		assert int(res['status'][i]) == ref['status']
		np.testing.assert_allclose(res['contamination'][i], ref['contamination'], rtol=1e-7, atol=1e-11)
		np.testing.assert_allclose(res['fluxes_mean'][star_offsets[i]:star_offsets[i+1]], ref['fluxes_mean'], rtol=1e-8, atol=1e-9*scale)


def test_spoc_grid_and_default_cutoff_stay_on_the_fast_kernels(ctx):
	"""The decision is made from the knots on the device: the SPOC layout with the plugin's cut-off (linpsf_photometry.py:63) never
	reaches the general kernels."""
	from photometry_amd import simulate, engine, psf as hpsf
	from photometry_amd.device import DeviceCube
	from oracle import psf as opsf
	s = simulate.make_scene(4, 20, 11, 11, seed=5, max_neighbours=2)
	simulate.fill_cubes(s)
	prf = opsf.synthetic_prf(seed=7)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	sel, star_offsets, target_index = hpsf.select_stars(s.catalog, s.cat_offsets, s.target_starid)
	pos_row, pos_col = _positions(s, sel, 20)
	coef = engine.linpsf_prf(ctx, ctx.array(model.base_coef), ctx.array(model.weights(s.stamps)))
	for cutoff, general in ((5, False), (5.25, False), (5.26, True)):
		engine.linpsf_fit(ctx, DeviceCube.from_host(ctx, s.images), coef, ctx.array(model.tx), ctx.array(model.ty),
			ctx.array(star_offsets), ctx.array(target_index), ctx.array(pos_row), ctx.array(pos_col), int(np.diff(star_offsets).max()), cutoff_radius=cutoff)
		counts = engine.linpsf_last_counts(ctx)
		fast = counts['matrix_core_targets'] + counts['vector_alu_polynomial_targets'] + counts['vector_alu_general_targets']
		assert counts['any_grid_targets'] == (4 if general else 0) and fast == (0 if general else 4)


def test_matrix_core_fit_takes_the_qualifying_targets(ctx):
	"""With the default mapping a batch of targets with up to four fitted stars and ordinary jitter is fitted by the matrix-core
	kernel alone (no vector-ALU fit launch); with path 0, and for stars that wander over more than three knot intervals, by the
	vector-ALU kernels -- same light curves either way."""
	from photometry_amd import simulate, engine, psf as hpsf
	from photometry_amd.device import DeviceCube
	from oracle import psf as opsf
	prf = opsf.synthetic_prf(seed=7)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	out = {}
	for jit in (1, 8):
		s = simulate.make_scene(24, 64, 13, 13, seed=77, max_neighbours=2, neighbour_tmag_range=(9.0, 16.0))
		s.jitter = s.jitter * jit
		simulate.fill_cubes(s, nan_fraction=0.002)
		sel, star_offsets, target_index = hpsf.select_stars(s.catalog, s.cat_offsets, s.target_starid)
		pos_row, pos_col = _positions(s, sel, 64)
		coef = engine.linpsf_prf(ctx, ctx.array(model.base_coef), ctx.array(model.weights(s.stamps)))
		cube = DeviceCube.from_host(ctx, s.images)
		for path in (1, 0):
			engine.linpsf_set_path(ctx, path)
			ctx.profile(True)
			ctx.profile_reset()
			try:
				res = engine.linpsf_fit(ctx, cube, coef, ctx.array(model.tx), ctx.array(model.ty), ctx.array(star_offsets), ctx.array(target_index),
					ctx.array(pos_row), ctx.array(pos_col), int(np.diff(star_offsets).max())).to_host()
				ctx.sync()
				rep = ctx.profile_report()
			finally:
				ctx.profile(False)
				engine.linpsf_set_path(ctx, 1)
			out[(jit, path)] = (res, set(rep))
	# ordinary jitter: matrix cores only / vector ALUs only
	assert 'tp_linpsf_fitm_kernel' in out[(1, 1)][1] and 'tp_linpsf_fit_kernel' not in out[(1, 1)][1]
	assert 'tp_linpsf_fitm_kernel' not in out[(1, 0)][1] and 'tp_linpsf_fit_kernel' in out[(1, 0)][1]
	# wide jitter: more than 3 x 3 knot intervals inside 16 cadences -> the plan leaves (most of) the targets to the vector-ALU kernels
	assert 'tp_linpsf_fit_kernel' in out[(8, 1)][1] or 'tp_linpsf_fit_direct_kernel' in out[(8, 1)][1]
	for jit in (1, 8):
		a, b = out[(jit, 1)][0], out[(jit, 0)][0]
		scale = np.nanmax(np.abs(a['flux']))
		np.testing.assert_allclose(a['flux'], b['flux'], rtol=1e-9, atol=1e-10 * scale)
		np.testing.assert_array_equal(a['status'], b['status'])


def _oracle_linpsf(s, prf, i, pos_row_of, pos_col_of, T):
	"""Oracle light curve of target i with per-star position series given by index into the catalogue of the target."""
	from oracle import psf as opsf, linpsf as olin
	cat = s.catalog_of(i)
	ncat = len(cat['starid'])
	positions = np.empty((T, ncat, 2))
	positions[:, :, 0] = pos_row_of(i, cat)
	positions[:, :, 1] = pos_col_of(i, cat)
	p = opsf.PSF(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'], tuple(s.stamps[i]))
	return olin.do_photometry(s.images[i], p, cat, s.target_starid[i], positions, tuple(s.stamps[i]),
		s.target_pos_row[i], s.target_pos_column[i], s.aperture[i])


@pytest.mark.parametrize("T,drift", [(200, 0.5), (1300, 0.8), (333, -0.9)])
def test_drifting_stars_stay_on_the_matrix_cores(ctx, T, drift):
	"""A pointing drift over the series (half a pixel and more = 4.5+ knot intervals of the PRF grid, linear in time, on top of
	the jitter): the series is cut into segments, each with the spline of the intervals visited THEN, and every target is fitted
	by the matrix-core kernel -- no vector-ALU fit launch -- within 1e-8 of the oracle and of the vector-ALU mapping."""
	from photometry_amd import simulate, engine, psf as hpsf
	from photometry_amd.device import DeviceCube
	from oracle import psf as opsf
	nt = 8 if T < 1000 else 3
	s = simulate.make_scene(nt, T, 13, 13, seed=91, max_neighbours=2, neighbour_tmag_range=(9.0, 16.0))
	ramp = np.linspace(0.0, 1.0, T)
	s.jitter = s.jitter + np.stack((drift * ramp, -0.7 * drift * ramp), axis=1)     # (column, row) drift
	simulate.fill_cubes(s, nan_fraction=0.003)
	prf = opsf.synthetic_prf(seed=7)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	sel, star_offsets, target_index = hpsf.select_stars(s.catalog, s.cat_offsets, s.target_starid)
	pos_row, pos_col = _positions(s, sel, T)
	coef = engine.linpsf_prf(ctx, ctx.array(model.base_coef), ctx.array(model.weights(s.stamps)))
	cube = DeviceCube.from_host(ctx, s.images)
	res, counts, kernels = {}, {}, {}
	for path in (1, 0):
		engine.linpsf_set_path(ctx, path)
		ctx.profile(True)
		ctx.profile_reset()
		try:
			res[path] = engine.linpsf_fit(ctx, cube, coef, ctx.array(model.tx), ctx.array(model.ty), ctx.array(star_offsets), ctx.array(target_index),
				ctx.array(pos_row), ctx.array(pos_col), int(np.diff(star_offsets).max())).to_host()
			ctx.sync()
			kernels[path] = set(ctx.profile_report())
			counts[path] = engine.linpsf_last_counts(ctx)
		finally:
			ctx.profile(False)
			engine.linpsf_set_path(ctx, 1)
	c = counts[1]
	assert c['matrix_core_targets'] == nt and c['vector_alu_polynomial_targets'] == 0 and c['vector_alu_general_targets'] == 0, c
	assert c['matrix_core_segments'] > nt, c                        # the drift needs more than one spline per star
	assert 'tp_linpsf_fitm_kernel' in kernels[1] and 'tp_linpsf_fit_kernel' not in kernels[1] and 'tp_linpsf_fit_direct_kernel' not in kernels[1]
	assert counts[0]['matrix_core_targets'] == 0
	scale = np.nanmax(np.abs(res[1]['flux']))
	np.testing.assert_allclose(res[1]['flux'], res[0]['flux'], rtol=1e-9, atol=1e-10 * scale)
	np.testing.assert_array_equal(res[1]['status'], res[0]['status'])
	np.testing.assert_allclose(res[1]['contamination'], res[0]['contamination'], rtol=1e-7, atol=1e-11)
	for i in range(min(nt, 3)):
		ref = _oracle_linpsf(s, prf, i, lambda i, cat: cat['row_stamp'][None, :] + s.jitter[:, 1][:, None],
			lambda i, cat: cat['column_stamp'][None, :] + s.jitter[:, 0][:, None], T)
		sc = np.nanmax(np.abs(ref['flux']))
		np.testing.assert_allclose(res[1]['flux'][i], ref['flux'], rtol=1e-8, atol=1e-9 * sc)
		assert int(res[1]['status'][i]) == ref['status']
		np.testing.assert_allclose(res[1]['contamination'][i], ref['contamination'], rtol=1e-7, atol=1e-11)


@pytest.mark.parametrize("path", [0, 1])
def test_target_without_a_fitted_star(ctx, path):
	"""A target whose own catalogue entry has a NaN magnitude loses it in the star selection (linpsf_photometry.py:93-101) and is
	left with no fitted star: its flux is all NaN and the plugin's status ERROR ('All target flux values are NaN'), and the
	targets beside it are fitted exactly as without it (round 3's plan kernel wrote such a target's list entry before the lists)."""
	from photometry_amd import simulate, engine, psf as hpsf
	from photometry_amd.device import DeviceCube
	from oracle import psf as opsf
	T = 48
	s = simulate.make_scene(5, T, 11, 11, seed=23, max_neighbours=0)
	simulate.fill_cubes(s, nan_fraction=0.002)
	prf = opsf.synthetic_prf(seed=7)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	sel, star_offsets, target_index = hpsf.select_stars(s.catalog, s.cat_offsets, s.target_starid)
	pos_row, pos_col = _positions(s, sel, T)
	coef = engine.linpsf_prf(ctx, ctx.array(model.base_coef), ctx.array(model.weights(s.stamps)))
	cube = DeviceCube.from_host(ctx, s.images)

	def fit(offsets, pr, pc, tidx):
		engine.linpsf_set_path(ctx, path)
		try:
			return engine.linpsf_fit(ctx, cube, coef, ctx.array(model.tx), ctx.array(model.ty), ctx.array(offsets), ctx.array(tidx),
				ctx.array(np.ascontiguousarray(pr)), ctx.array(np.ascontiguousarray(pc)), 1).to_host()
		finally:
			engine.linpsf_set_path(ctx, 1)
	full = fit(star_offsets, pos_row, pos_col, target_index)
	assert np.all(np.diff(star_offsets) == 1)
	# drop the star of target 2: offsets 0 1 2 2 3 4
	keep = np.ones(len(pos_row), dtype=bool)
	keep[star_offsets[2]] = False
	offs = np.concatenate(([0], np.cumsum(np.diff(star_offsets) - (np.arange(5) == 2)))).astype('int64')
	got = fit(offs, pos_row[keep], pos_col[keep], target_index)
	assert int(got['status'][2]) == 2 and np.all(np.isnan(got['flux'][2][:T]))          # STATUS.ERROR
	for i in (0, 1, 3, 4):
		np.testing.assert_array_equal(got['flux'][i][:T], full['flux'][i][:T])
		assert int(got['status'][i]) == int(full['status'][i])
		np.testing.assert_array_equal(got['contamination'][i], full['contamination'][i])


def test_star_positions_on_the_device():
	"""tp_star_positions: float64(base[s] + shift[k]) with the sum in float32 -- bit for bit the host expression the batched LinPSF
	entry used to build (and the plugin's catalogue after catalog_attime for a translation); ragged sizes, a pitch beyond T."""
	from photometry_amd import engine
	from photometry_amd.device import Context
	ctx = Context(0)
	rng = np.random.default_rng(3)
	for n, T in ((1, 1), (7, 1300), (70001, 33), (5, 257)):
		base = rng.uniform(0, 2048, n).astype('float32')
		shift = rng.normal(0, 0.3, T).astype('float32')
		got = engine.star_positions(ctx, ctx.array(base), ctx.array(shift)).to_host()
		np.testing.assert_array_equal(got, (base[:, None] + shift[None, :]).astype('float64'))
	ctx.close()
