# -*- coding: utf-8 -*-
"""CPU tests of the background oracle (B*, B2, B3), incl. the reference's own known answer."""
import numpy as np
from oracle import backgrounds as ob


def test_constant_image_known_answer():
	"""tests/test_background.py:36-54 of the reference: constant 1000 -> background 1000, nothing masked."""
	for shape in ((15, 15), (11, 11), (64, 64)):
		fakeimg = np.full(shape, 1000, dtype='float32')
		bck, mask = ob.fit_background_stamp(fakeimg)
		assert mask.shape == fakeimg.shape and mask.dtype == bool
		assert not np.any(mask), "Nothing should be masked out"
		np.testing.assert_allclose(bck, 1000)


def test_mask_rules():
	img = np.full((15, 15), 50.0, dtype='float32')
	img[0, 0] = np.nan; img[0, 1] = np.inf; img[0, 2] = 9e4; img[0, 3] = -1.0
	bck, mask = ob.fit_background_stamp(img)
	assert mask.sum() == 4 and mask[0, :4].all()          # backgrounds.py:91-94
	np.testing.assert_allclose(bck, 50.0)
	img[:8, :] = np.nan                                    # > 50 % masked -> no estimate
	assert np.isnan(ob.fit_background_stamp(img)[0])
	assert np.isnan(ob.fit_background_stamp(np.full((5, 5), np.nan, dtype='float32'))[0])


def test_sigma_clip_and_mode_estimator():
	rng = np.random.default_rng(0)
	img = rng.normal(100, 3, (15, 15)).astype('float32')
	img[7, 7] = 5000; img[7, 8] = 3000; img[6, 7] = 800   # a star: must be clipped away
	bck, _ = ob.fit_background_stamp(img)
	assert abs(bck - 100) < 1.0
	d = ob.sigma_clip(img.ravel())
	assert d.size <= 222 and d.max() < 200
	# SExtractor formula branches
	x = np.array([1, 1, 1, 1.0])
	assert ob.sextractor_background(x) == 1.0              # std == 0 -> mean
	x = np.array([0, 0, 0, 0, 10.0])                       # |mean-med|/std = 2/4 >= 0.3 -> median
	assert ob.sextractor_background(x) == 0.0
	x = np.array([9., 10., 11., 10., 10.4])                # small skew -> 2.5 med - 1.5 mean
	np.testing.assert_allclose(ob.sextractor_background(x), 2.5*10 - 1.5*x.mean())


def test_smooth_time_matches_definition():
	"""prepare.py:317-335"""
	rng = np.random.default_rng(1)
	x = rng.normal(100, 5, (3, 40)).astype('float32')
	x[0, 5] = np.nan; x[1, :3] = np.nan; x[2, 10:19] = np.nan
	for ts in (3, 9):
		y = ob.smooth_time(x, ts)
		w = ts // 2
		for k in range(40):
			blk = x[:, max(k - w, 0):min(k + w + 1, 40)]
			with np.errstate(invalid='ignore'):
				ref = np.nanmean(blk.astype('float64'), axis=1) if True else None
			ok = ~np.isnan(ref)
			np.testing.assert_allclose(y[ok, k], ref[ok], rtol=1e-6)
			assert np.all(np.isnan(y[~ok, k]))
	assert ob.time_smooth_width(1800) == 3 and ob.time_smooth_width(600) == 9


def test_subtract_background():
	"""prepare.py:419-425"""
	rng = np.random.default_rng(2)
	raw = rng.normal(500, 10, (4, 5, 6)).astype('float32')
	err = np.sqrt(raw).astype('float32')
	bkg = rng.normal(100, 1, 6).astype('float32')
	flags = np.zeros((4, 5, 6), dtype='uint8'); flags[1, 2, 3] = 2; flags[0, 0, 0] = 1
	img, e = ob.subtract_background(raw, err, bkg, flags)
	assert np.isnan(img[1, 2, 3]) and np.isnan(e[1, 2, 3]) and np.isfinite(img[0, 0, 0])
	ok = flags != 2
	np.testing.assert_array_equal(img[ok], (raw - bkg)[ok])
	np.testing.assert_array_equal(e[ok], err[ok])


def test_b2_b3_against_the_reference_statements(golden_dir):
	"""golden_background.npz = prepare.py:317-335 and :419-425 EXECUTED (tests/golden/make_golden.py reads those lines from the
	reference and runs them on dict-backed HDF5 stand-ins): smoothing windows, edge clipping, float32 block, NaN handling,
	BACKAPP switch, manual-exclude masking."""
	import os
	g = np.load(os.path.join(golden_dir, 'golden_background.npz'))
	frames = g['b2_frames']                                   # (N, H, W): one background image per cadence
	for ts in (3, 9):
		got = ob.smooth_time(np.moveaxis(frames, 0, -1), ts)   # (H, W, N), smoothing along the last axis
		np.testing.assert_array_equal(np.moveaxis(got, -1, 0), g[f'b2_ts{ts}_smoothed'])
	bk = g['b2_ts3_smoothed']
	for backapp in (0, 1):
		img, err = ob.subtract_background(g['b3_raw'], g['b3_raw_err'], bk, g['b3_flags'], backapp=bool(backapp))
		np.testing.assert_array_equal(img, g[f'b3_images_backapp{backapp}'])
		np.testing.assert_array_equal(err, g[f'b3_errors_backapp{backapp}'])
	assert np.isnan(g['b3_images_backapp0'][3, 1, 1]) and np.isfinite(g['b3_images_backapp0'][3, 2, 2])   # only ManualExclude (2) masks
