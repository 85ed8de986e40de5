# -*- coding: utf-8 -*-
"""
Pin the oracle (CPU restatement) against the golden vectors produced by executing the
reference's own code (tests/golden/make_golden.py).  CPU only.
"""
import os
import numpy as np
import pytest
from oracle import quality, utilities, sumimage, aperture, psf as opsf, linpsf, k2p2
from scipy.interpolate import RectBivariateSpline


def _load(golden_dir, name):
	return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_misc(golden_dir):
	g = _load(golden_dir, 'golden_misc.npz')
	assert quality.TESS_DEFAULT_BITMASK == int(g['quality_default_bitmask']) == 4335
	np.testing.assert_array_equal(quality.tess_filter(g['quality_in']), g['quality_filter'])
	np.testing.assert_array_equal(quality.pixel_filter(g['pixelflags_in']), g['pixelflags_filter'])
	np.testing.assert_array_equal(utilities.mag2flux(g['mag_in']), g['mag2flux'])
	assert utilities.mad_to_sigma == float(g['mad_to_sigma'])
	# reference's own known answers: tests/test_utilities.py:24-35, utilities.py:121-126
	np.testing.assert_allclose(utilities.move_median_central(g['mmc_in'], 3), [3, 2, 2, 0, 0, 0, 1, 2, 2, 3])
	np.testing.assert_allclose(utilities.move_median_central(g['mmc_in'], 3), g['mmc_out'])
	X, Y = np.meshgrid(np.arange(-1, 2), np.arange(-1, 2))
	np.testing.assert_array_equal(utilities.integratedGaussian(X, Y, 10, 0, 0), g['ig_out'])
	np.testing.assert_allclose(g['ig_out'], [[0.58433556, 0.92564571, 0.58433556],
		[0.92564571, 1.46631496, 0.92564571], [0.58433556, 0.92564571, 0.58433556]], rtol=1e-7)


def test_sumimage(golden_dir):
	g = _load(golden_dir, 'golden_sumimage.npz')
	S = sumimage.sumimage_batch(g['images'], g['quality'])
	np.testing.assert_array_equal(S, g['sumimage']) # bit-exact incl. NaN positions
	assert np.isnan(S).sum() == 2


def _catalog(g, i, prefix='cat_'):
	a, b = g['cat_offsets'][i], g['cat_offsets'][i+1]
	return {k[len(prefix):]: g[k][a:b] for k in g.files if k.startswith(prefix) and k != 'cat_offsets'}


def test_aperture(golden_dir):
	g = _load(golden_dir, 'golden_aperture.npz')
	for n in range(int(g['n_cases'])):
		i = int(g[f'case{n}_target'])
		kind = str(g[f'case{n}_kind'])
		cat = _catalog(g, i)
		if f'case{n}_cat_row' in g.files:
			cat['row'] = g[f'case{n}_cat_row']
			cat['column'] = g[f'case{n}_cat_column']
		masks = g[f'case{n}_masks']
		masks = None if (kind in ('none', 'nostars') or masks.shape[0] == 0) else masks.astype(bool)
		res = aperture.do_photometry(g['sumimage'][i], g['images'][i], g['images_err'][i], g['backgrounds'][i],
			tuple(g['stamps'][i]), g['target_pos_row'][i], g['target_pos_column'][i], g['target_tmag'][i],
			g['target_starid'][i], cat, g['aperture'][i], masks=masks)
		assert res['status'] == int(g[f'case{n}_status']), f"case {n}"
		if bool(g[f'case{n}_has_mask']):
			np.testing.assert_array_equal(res['mask'], g[f'case{n}_final_mask'])
			# float32 np.sum / sqrt: bit-exact
			np.testing.assert_array_equal(res['flux'], g[f'case{n}_flux'])
			np.testing.assert_array_equal(res['flux_err'], g[f'case{n}_flux_err'])
			np.testing.assert_allclose(res['pos_centroid'], g[f'case{n}_pos_centroid'], rtol=1e-14)
			np.testing.assert_array_equal(res['flux_background'], g[f"case{n}_flux_background"]) # np.nansum (photometry.py:201)
			cont = float(g[f'case{n}_contamination'])
			if np.isnan(cont):
				assert np.isnan(res['contamination'])
			else:
				assert abs(res['contamination'] - cont) < 1e-6
			np.testing.assert_array_equal(np.asarray(res['skip_targets'], dtype='int64'), g[f'case{n}_skip_targets'])
	# special frames of case 4 behaved as designed
	f = g['case4_flux']
	assert np.isnan(f[2]) and np.isnan(f[3]) and np.isfinite(f[4]) and np.isnan(f[7])
	assert np.all(np.isnan(g['case4_pos_centroid'][4]))
	assert np.isnan(g['case4_flux_background'][5]) and np.isfinite(g['case4_flux_background'][6])


def test_k2p2_reference_control_flow(golden_dir):
	"""The reference's own k2p2FixFromSum (real scipy/sklearn + stand-ins) vs the oracle restatement."""
	g = _load(golden_dir, 'golden_k2p2.npz')
	settings = dict(thresh=0.8, min_no_pixels_in_mask=4, min_for_cluster=4,
		cluster_radius=np.sqrt(2) + np.finfo(np.float64).eps, segmentation=True, ws_blur=0.5, ws_thres=0,
		ws_footprint=3, extend_overflow=True)
	nmulti = 0
	for n in range(int(g['n_cases'])):
		S = g[f'k{n}_sumimage']
		cat = g[f'k{n}_catalog']
		ref = g[f'k{n}_masks']
		assert str(g[f'k{n}_err']) == ''
		# golden was made with the scipy installed here (bracket validation on):
		masks, bw = k2p2.k2p2FixFromSum(S, catalog=cat, validate_bracket=True, **settings)
		assert bw == float(g[f'k{n}_bw'])
		if ref.shape[0] == 0:
			assert masks is None
			continue
		assert masks.shape == ref.shape
		nmulti += ref.shape[0] > 1
		# order of equally-sized masks is unspecified in the reference (unstable argsort): compare as sets
		a = sorted(m.astype(bool).tobytes() for m in masks)
		b = sorted(m.astype(bool).tobytes() for m in ref)
		assert a == b, f"case {n}"
	assert nmulti >= 5


def test_psf_integrate(golden_dir):
	g = _load(golden_dir, 'golden_psf.npz')
	x = g['prf_x']
	spline = RectBivariateSpline(x, x, g['prf_img'])
	for n in range(int(g['n_cases'])):
		shape = tuple(int(v) for v in g[f'p{n}_shape'])
		cutoff = float(g[f'p{n}_cutoff'])
		cutoff = None if np.isnan(cutoff) else cutoff
		p = opsf.PSF.from_spline(spline, shape)
		img = p.integrate_to_image(g[f'p{n}_params'], cutoff_radius=cutoff)
		ref = g[f'p{n}_img']
		np.testing.assert_allclose(img, ref, rtol=1e-11, atol=1e-15*np.abs(ref).max())
		np.testing.assert_array_equal(img == 0, ref == 0)
	# reference test tests/test_psf.py:37-63: unit star at a pixel centre peaks at that pixel
	p = opsf.PSF.from_spline(spline, (15, 15))
	img = p.integrate_to_image(np.array([[7.0, 7.0, 1.0]]))
	assert np.unravel_index(np.argmax(img), img.shape) == (7, 7)


def test_linpsf(golden_dir):
	g = _load(golden_dir, 'golden_linpsf.npz')
	for n in range(int(g['n_lsfit'])):
		x = linpsf.lsfit(g[f'ls{n}_A'], g[f'ls{n}_b'])
		np.testing.assert_array_equal(x, g[f'ls{n}_x'])
	x = g['prf_x']
	spline = RectBivariateSpline(x, x, g['prf_img'])
	for n in range(int(g['n_linpsf'])):
		i = int(g[f'lp{n}_target'])
		cat = _catalog(g, i)
		H, W = g['images'].shape[1:3]
		p = opsf.PSF.from_spline(spline, (H, W))
		res = linpsf.do_photometry(g['images'][i], p, cat, g['target_starid'][i], g[f'lp{n}_positions'],
			tuple(g['stamps'][i]), g['target_pos_row'][i], g['target_pos_column'][i], g['aperture'][i])
		assert res['status'] == int(g[f'lp{n}_status'])
		np.testing.assert_allclose(res['flux'], g[f'lp{n}_flux'], rtol=1e-9)
		assert np.all(np.isnan(res['flux_err']))
		np.testing.assert_allclose(res['contamination'], float(g[f'lp{n}_contamination']), rtol=1e-8, atol=1e-12)


def test_diagnostics_golden(golden_dir):
	"""oracle.diagnostics vs the reference's own BasePhotometry.photometry() (BasePhotometry.py:1343-1407)."""
	from oracle import diagnostics as odiag
	g = np.load(os.path.join(golden_dir, 'golden_diagnostics.npz'))
	for i in range(int(g['n_cases'])):
		assert int(g[f'case{i}_status']) in (1, 3)
		d = odiag.diagnostics(g['time'], g['quality'], g[f'case{i}_flux'], g[f'case{i}_flux_err'], g[f'case{i}_pos_centroid'],
			sumimage=g['sumimage'][i], mask=g[f'case{i}_mask'])
		assert d['flags'] == 0
		for key in ('mean_flux', 'ptp', 'mask_size', 'edge_flux'):
			assert d[key] == float(g[f'case{i}_{key}']), key                      # selections / integer / same summation: exact
		for key in ('variance', 'rms_hour', 'variability'):
			np.testing.assert_allclose(d[key], float(g[f'case{i}_{key}']), rtol=1e-12, err_msg=key)
		np.testing.assert_array_equal([d['pos_centroid_col'], d['pos_centroid_row']], g[f'case{i}_det_pos_centroid'])


def test_cutout_golden(golden_dir):
	"""oracle.cutout.load_cube vs the reference's own BasePhotometry._load_cube (BasePhotometry.py:720-742)."""
	from oracle import cutout
	g = np.load(os.path.join(golden_dir, 'golden_cutout.npz'))
	for st, ref in zip(g['stamps'], g['cubes']):
		got = cutout.load_cube(g['frames'], tuple(st), *g['offsets'])
		np.testing.assert_array_equal(got, ref)


def test_psf_photometry_golden(golden_dir):
	"""oracle.psf_photometry vs the reference's own PSFPhotometry.do_photometry (psf_photometry.py:111-196, real scipy
	Nelder-Mead); and the oracle's restatement of scipy's Nelder-Mead gives the same fits as scipy itself."""
	from oracle import psf_photometry as opp
	g = _load(golden_dir, 'golden_psfphot.npz')
	x = g['prf_x']
	spline = RectBivariateSpline(x, x, g['prf_img'])
	H, W = g['images'].shape[1:3]
	for n in range(int(g['n_psfphot'])):
		i = int(g[f'pp{n}_target'])
		cat = _catalog(g, i)
		p = opsf.PSF.from_spline(spline, (H, W))
		for use_scipy in (True, False):
			res = opp.do_photometry(g['images'][i], g['backgrounds'][i], p, cat, tuple(g['stamps'][i]), g['target_pos_row'][i],
				g['target_pos_column'][i], g['target_tmag'][i], g['aperture'][i], use_scipy=use_scipy)
			assert res['status'] == int(g[f'pp{n}_status'])
			# the simplex search is a chain of comparisons: last-bit differences of the PRF integral (own FITPACK restatement vs
			# scipy's) may change single steps, both runs end within the optimiser's own tolerances (xatol = fatol = 1e-4)
			np.testing.assert_allclose(res['flux'], g[f'pp{n}_flux'], rtol=2e-5)
			np.testing.assert_allclose(res['pos_centroid'], g[f'pp{n}_pos_centroid'], atol=2e-4)
			assert np.all(np.isnan(res['flux_err']))


def test_pixel_manual_exclude_golden(golden_dir):
	"""pixel_flags.pixel_manual_exclude (pixel_flags.py:13-58): the oracle's restatement and the product's host rule against the
	reference's own function on 19 header / data cases (golden_pixelflags.npz)."""
	from oracle import backgrounds as ob
	from photometry_amd import prepare
	g = _load(golden_dir, 'golden_pixelflags.npz')
	assert len(g['camera']) == 19 and set(g['first_excluded_column']) == {0, 1536, 2048}
	for i in range(len(g['camera'])):
		is_tess, zero = bool(g['is_tess'][i]), bool(g['zero'][i])
		ffi = None if g['ffiindex'][i] < 0 else int(g['ffiindex'][i])
		data = np.zeros((4, 2048), dtype='float32') if zero else np.ones((4, 2048), dtype='float32')
		m = ob.pixel_manual_exclude(data, is_tess, int(g['camera'][i]), int(g['ccd'][i]), ffi, float(g['tstart'][i]), float(g['tstop'][i]))
		cols = m.all(axis=0)
		assert np.array_equal(m, np.broadcast_to(cols, m.shape))
		first = int(np.argmax(cols)) if cols.any() else 2048
		assert first == g['first_excluded_column'][i], i
		rule = prepare.manual_exclude_columns(1, 2048, is_tess, int(g['camera'][i]), int(g['ccd'][i]), None if ffi is None else [ffi],
			[float(g['tstart'][i])], [float(g['tstop'][i])])[0]
		if zero and is_tess: # the zero-image rule is applied on the device (tp_frames_pixel_flags)
			rule = 0
		assert rule == g['first_excluded_column'][i], i




def test_shenanigans_mean_golden(golden_dir):
	"""The block-median mean of the shenanigans indicator (prepare.py:557-571 executed by the generator), including frame counts
	that are not a multiple of 25: the short last block's median also runs over the frames the previous block left in the
	reference's buffer."""
	from oracle import backgrounds as ob
	g = _load(golden_dir, 'golden_shenanigans.npz')
	for c in range(int(g['n_cases'])):
		ind = g[f's{c}_indicator']                      # (T, R, C) float32
		T = ind.shape[0]
		indicies = list(range(T))
		np.random.seed(0)
		np.random.shuffle(indicies)
		mean = np.zeros(ind.shape[1:])
		for k in range(0, T, 25):
			blockdata = np.stack([ind[i].astype('float64') for i in ob.shenanigans_block_frames(indicies, k)], axis=2)
			with np.errstate(all='ignore'):
				import warnings
				with warnings.catch_warnings():
					warnings.simplefilter('ignore')
					b = np.nanmedian(blockdata, axis=2)
			b[np.isnan(b)] = 0
			mean += b
		mean /= np.ceil(T / 25)
		np.testing.assert_array_equal(mean, g[f's{c}_mean'])
	# a last block of 10 frames after a full one: 15 stale frames take part
	assert len(ob.shenanigans_block_frames(list(range(60)), 50)) == 25 and len(ob.shenanigans_block_frames(list(range(60)), 25)) == 25
	assert len(ob.shenanigans_block_frames(list(range(10)), 0)) == 10


def test_psf_distribution_fixture_belongs_to_the_seeded_scene(golden_dir):
	"""tests/golden/golden_psf_distribution.npz (the oracle's Nelder-Mead fit of 120 targets x 20 cadences, 21 minutes to make) holds
	outputs only; the GPU test rebuilds the scene from the recorded seeds: the rebuilt scene must be the one that was fitted."""
	import sys
	sys.path.insert(0, golden_dir)
	import make_psf_distribution as mk
	g = np.load(os.path.join(golden_dir, 'golden_psf_distribution.npz'))
	assert tuple(int(v) for v in g['shape']) == (mk.NT, mk.T, mk.H, mk.W)
	s, _prf = mk.build_scene()
	assert float(np.nansum(s.images.astype('float64'))) == float(g['images_checksum'][0])
	assert g['flux'].shape == (mk.NT, mk.T) and g['nit'].max() == 1500 and 0.4 < np.isfinite(g['flux']).mean() < 0.9   # two- and three-star fits mostly run into maxiter: NaN by the reference's rule
	assert (g['status'] == 1).all()      # the plugin returns OK; NaN fluxes are only logged (psf_photometry.py:190-194)
