# -*- coding: utf-8 -*-
"""Shared helpers of the K2P2 parity tests (host-sim on CPU, HIP kernel on GPU)."""
import numpy as np
from oracle import aperture as oap, k2p2 as ok2p2, sumimage as osum


def make_cases(kind, seed):
	"""Seeded scenes that exercise the different K2P2 branches."""
	from photometry_amd import simulate
	if kind == 'faint15':
		s = simulate.make_scene(48, 120, 15, 15, seed=seed)
	elif kind == 'small11':
		s = simulate.make_scene(32, 60, 11, 11, seed=seed)
	elif kind == 'crowded':
		s = simulate.make_scene(32, 100, 21, 17, seed=seed, max_neighbours=5, neighbour_tmag_range=(8.0, 13.0))
	elif kind == 'bright':
		s = simulate.make_scene(24, 150, 15, 15, seed=seed, tmag_range=(4.0, 7.5), neighbour_tmag_range=(6.0, 9.0))
	elif kind == 'tiny':
		s = simulate.make_scene(12, 40, 6, 7, seed=seed, max_neighbours=1)
	elif kind == 'wide':
		# a resized stamp of a crowded field: a dozen clusters, several of them at the stamp's edges and corners -- the per-cluster
		# passes of the mask builder run over windows around each cluster (k2p2_core.h: struct Win)
		s = simulate.make_scene(20, 60, 29, 33, seed=seed, max_neighbours=14, neighbour_tmag_range=(7.5, 13.5))
	elif kind == 'huge':
		# blended very bright stars on a 100 x 96 stamp: ONE cluster of several thousand pixels split by the watershed -- more ranks
		# than the three summary words of the flood's bit set hold (3 072): the scanning path of k2p2_core.h::watershed
		s = simulate.make_scene(3, 8, 100, 96, seed=seed, tmag_range=(1.0, 2.0), max_neighbours=2, neighbour_tmag_range=(1.0, 2.5), sigma_psf=7.0)
	elif kind == 'large':
		# beyond the LDS-resident mask builder (about 54 x 54 pixels): the work arrays live in HBM
		s = simulate.make_scene(5, 24, 62, 58, seed=seed, tmag_range=(4.0, 6.0), max_neighbours=6, neighbour_tmag_range=(6.0, 10.0), sigma_psf=1.6)
	else:
		raise ValueError(kind)
	simulate.fill_cubes(s, nan_fraction=0.002)
	S = osum.sumimage_batch(s.images, s.quality)
	rng = np.random.default_rng(seed)
	if kind == 'bright':
		# emulate saturated bleed columns
		for i in range(s.n_targets):
			c = int(round(s.star_params[i, 0, 1])); r = int(round(s.star_params[i, 0, 0]))
			sat = 0.6*np.nanmax(S[i])
			for rr in range(max(r-5, 0), min(r+6, s.height)):
				S[i, rr, c] = sat*(1 + 0.001*np.sin(rr))
	# a few degenerate sum images
	if s.n_targets >= 12:
		S[1][:] = np.nan                       # no flux at all -> K2P2NoFlux -> ERROR
		S[2][:] = -np.abs(S[2])                # no positive flux -> ERROR
		S[3][:] = 5.0                          # constant -> bandwidth 0 -> ERROR
		S[4][S[4] > np.nanpercentile(S[4], 30)] = np.nan   # most pixels never observed
		S[5][2, 3] = np.nan
		S[6][:] = np.abs(rng.normal(10, 1e-3, S[6].shape))  # pure noise, no star
		# target far from stamp centre (still inside), and on the edge
		s.target_pos_row[7] = s.stamps[7, 0] + 0.4
		s.target_pos_column[7] = s.stamps[7, 2] + s.width - 1.2
		# catalog empty for target 8
	return s, S


def oracle_batch(s, S, cut_override=None):
	out = []
	for i in range(s.n_targets):
		cat = s.catalog_of(i)
		kw = {}
		try:
			if cut_override is not None:
				c = np.column_stack((cat['column_stamp'], cat['row_stamp'], cat['tmag']))
				try:
					mm, info = ok2p2.k2p2FixFromSum(S[i], catalog=c, cut_override=cut_override[i], full_output=True, **oap.K2P2_SETTINGS)
					mm = None if mm is None else np.asarray(mm, dtype=bool)
				except ok2p2.K2P2NoStars:
					mm = None
				kw['masks'] = mm
			r = oap.do_photometry(S[i], s.images[i], s.images_err[i], s.backgrounds[i], tuple(s.stamps[i]),
				s.target_pos_row[i], s.target_pos_column[i], s.target_tmag[i], s.target_starid[i], cat,
				# bit 1 of BasePhotometry.aperture = finite sum image (BasePhotometry.py:1043)
				s.aperture[i] & np.isfinite(S[i]).astype(s.aperture.dtype), **kw)
		except Exception as e: # noqa: B902  -- the reference turns any exception into STATUS.ERROR (tessphot.py:37-49)
			r = {'status': oap.STATUS_ERROR, 'exception': repr(e)}
		if cut_override is None:
			try:
				r['thr'] = ok2p2.threshold(S[i], 0.8, full_output=True)
			except Exception: # noqa: B902
				r['thr'] = None
		out.append(r)
	return out


def compare(s, S, got, ref, check_cut=True):
	"""``got``: dict of arrays mask,status,flags,contamination,diag,cat_in_mask.  Returns stats dict."""
	n_exact = 0
	n_error_agree = 0
	n_razor = 0
	n_tie = 0
	margins = []
	dcuts = []
	for i in range(s.n_targets):
		r = ref[i]
		razor = False
		if check_cut and r.get('thr') is not None and np.isfinite(r['thr']['CUT']):
			# The threshold comes out of a Brent line search that stops at a 1e-2 relative step
			# (scipy Powell, xtol*100): last-bit differences of exp()/summation order in the KDE are
			# amplified to ~1e-8 in CUT -- numpy's own SIMD exp differs between CPUs by as much.  So CUT is
			# compared to 2e-6 relative, and mask bit-exactness is asserted whenever no pixel lies within
			# 4*|dCUT| of the oracle's CUT (the complementary test feeds the oracle's CUT to the kernel).
			thr = r['thr']
			d = got['diag'][i]
			dcut = abs(d[0] - thr['CUT'])
			assert abs(d[3] - thr['bandwidth']) <= 1e-12*abs(thr['bandwidth'])
			assert int(d[5]) == thr['nflux']
			# Two samples give a KDE with two maxima of EQUAL height: which one the first argmax of the FFT density lands on is
			# decided by the transform's rounding noise (numpy's pocketfft there, a radix-2 FFT here) -- seen once in 5 936 fuzz
			# targets, a sum image with two positive pixels.  The threshold is then not comparable; the outputs below still are.
			tie = thr.get('nflux_cut', 3) <= 2 and dcut > 2e-6*max(1.0, abs(thr['CUT']))
			if tie:
				n_tie += 1
			else:
				# the KDE's argmax (the start of the Powell search): the same grid point.  (The search is forgiving -- a start one grid
				# step off usually ends in the same mode -- which hid a linear binning that went wrong above 512 samples until round 6.)
				assert abs(d[4] - thr['max_guess']) <= 1e-9*max(1.0, abs(thr['max_guess'])), f"target {i}: KDE argmax {d[4]} vs {thr['max_guess']}"
				assert dcut <= 2e-6*max(1.0, abs(thr['CUT'])), f"target {i}: CUT {d[0]} vs {thr['CUT']}"
				with np.errstate(invalid='ignore'):
					margin = np.nanmin(np.abs(S[i] - thr['CUT']))
				margins.append(margin)
				dcuts.append(dcut)
				razor = margin <= 4*dcut
		if razor:
			n_razor += 1
			continue
		assert int(got['status'][i]) == r['status'], f"target {i}: status {got['status'][i]} vs {r['status']} ({r.get('errors')}, {r.get('exception')}) flags={got['flags'][i]:#x}"
		if 'mask' in r and r['status'] != oap.STATUS_ERROR:
			np.testing.assert_array_equal(got['mask'][i].astype(bool), r['mask'], err_msg=f"target {i} mask")
			assert bool(got['flags'][i] & 1) == bool(r['using_minimum_mask']), f"target {i} min-aperture flag"
			c = r['contamination']
			if np.isnan(c):
				assert np.isnan(got['contamination'][i])
			else:
				assert abs(got['contamination'][i] - c) < 2e-6, f"target {i}: contamination {got['contamination'][i]} vs {c}"
			a, b = s.cat_offsets[i], s.cat_offsets[i+1]
			inm = np.zeros(b - a, dtype=bool)
			inm[r['target_in_mask']] = True
			np.testing.assert_array_equal(got['cat_in_mask'][a:b].astype(bool), inm)
			e = r.get('edge', {})
			fl = int(got['flags'][i])
			assert bool(fl & 2) == ('down' in e) and bool(fl & 4) == ('up' in e) and bool(fl & 8) == ('left' in e) and bool(fl & 16) == ('right' in e)
			n_exact += 1
		else:
			n_error_agree += 1   # no mask on the reference's side (ERROR): the statuses were asserted equal above
	return {'n_exact': n_exact, 'n_error_agree': n_error_agree, 'n_razor': n_razor, 'n_tie': n_tie, 'min_margin': float(np.min(margins)) if margins else np.nan,
		'max_dcut': float(np.max(dcuts)) if dcuts else 0.0}


def own_chain_check(S_dev, images, images_err, backgrounds, quality, stamp, pos_row, pos_col, tmag, starid, catalog, aperture, got_mask, got_status,
	got_flux=None, got_flux_err=None, got_flux_background=None):
	"""
	The parity chain closed at one target: the ORACLE's own sum image (its restatement of prepare.py:450-459 /
	BasePhotometry.py:1008-1019 on the same cube) -> the oracle's own mask and light curve, against the device's mask (built from
	the device's sum image).  The two sum images agree to ~1e-16 relative (float64 sums of float32 values in another order), so
	the masks can differ only where a pixel of the sum image sits within that distance (times the K2P2 threshold's sensitivity,
	see :func:`compare`) of the threshold: such a target is reported as a razor case, anything else must be equal.
	Returns ``'exact'``, ``'razor'`` or raises.
	"""
	S_or = osum.sumimage(images, quality)
	with np.errstate(invalid='ignore'):
		np.testing.assert_allclose(S_dev, S_or, rtol=1e-12, atol=0, equal_nan=True)
	try:
		ref = oap.do_photometry(S_or, images, images_err, backgrounds, stamp, pos_row, pos_col, tmag, starid, catalog, aperture)
	except Exception as e: # noqa: B902 -- any exception in the plugin is STATUS.ERROR upstream (tessphot.py:37-49)
		ref = {'status': oap.STATUS_ERROR, 'exception': repr(e)}
	same = int(got_status) == ref['status'] and (ref['status'] == oap.STATUS_ERROR or 'mask' not in ref or np.array_equal(np.asarray(got_mask).astype(bool), ref['mask']))
	if not same:
		# razor: a pixel of the sum image within the reach of the threshold's own uncertainty (2e-6 relative, see compare())
		thr = ok2p2.threshold(S_or, 0.8, full_output=True)
		with np.errstate(invalid='ignore'):
			margin = np.nanmin(np.abs(S_or - thr['CUT']))
		assert margin <= 8e-6 * max(1.0, abs(thr['CUT'])), f"masks differ off the razor's edge: margin {margin}, CUT {thr['CUT']}, status {got_status} vs {ref['status']}"
		return 'razor'
	if ref['status'] != oap.STATUS_ERROR and 'mask' in ref and got_flux is not None:
		np.testing.assert_array_equal(got_flux, ref['flux'])
		np.testing.assert_array_equal(got_flux_err, ref['flux_err'])
		if got_flux_background is not None and backgrounds is not None:
			np.testing.assert_array_equal(got_flux_background, ref['flux_background'])
	return 'exact'
