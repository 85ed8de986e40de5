# -*- coding: utf-8 -*-
"""
The whole data path of a CCD region without leaving HBM: raw frames -> prepare stage (pixel flags, backgrounds B1, smoothing B2,
subtraction B3, sum image A1) -> aperture photometry of every target with its stamp-resize loop (A1..A7, stamps cut on the
device from the prepare stage's own output arrays) -- against the oracle's restatement of the same chain
(prepare.py:265-459 -> BasePhotometry._load_cube -> AperturePhotometry.do_photometry).

The background estimator itself is compared at 1e-6 elsewhere (test_gpu_fullframe.py); here the oracle continues from the
device's unsmoothed backgrounds, so that everything downstream -- smoothing, subtraction, manual excludes, sum images, masks,
float32 aperture sums -- must come out bit for bit.
"""
import numpy as np
import pytest
from scipy.special import erf

pytestmark = pytest.mark.gpu


def _raw_region(seed=21, R=160, C=176, T=20):
	rng = np.random.default_rng(seed)
	row0, col0 = 512, 700
	stars = [(row0 + 40.3, col0 + 35.6, 10.5), (row0 + 42.9, col0 + 38.4, 12.3), (row0 + 100.4, col0 + 60.7, 9.2), (row0 + 70.2, col0 + 120.4, 11.4),
		(row0 + 125.7, col0 + 140.1, 8.6), (row0 + 20.1, col0 + 150.3, 12.9), (row0 + 140.5, col0 + 30.2, 10.9), (row0 + 80.0, col0 + 10.9, 11.8)]
	rr, cc = np.arange(R) + row0, np.arange(C) + col0
	img = np.zeros((R, C))
	for (r, c, tmag) in stars:
		flux = 10**(-0.4 * (tmag - 20.451))
		pr = 0.5 * (erf((rr + 0.5 - r) / (np.sqrt(2) * 0.9)) - erf((rr - 0.5 - r) / (np.sqrt(2) * 0.9)))
		pc = 0.5 * (erf((cc + 0.5 - c) / (np.sqrt(2) * 0.9)) - erf((cc - 0.5 - c) / (np.sqrt(2) * 0.9)))
		img += flux * np.outer(pr, pc)
	yy, xx = np.mgrid[0:R, 0:C]
	raw = np.empty((T, R, C), dtype='float32')
	err = np.empty((T, R, C), dtype='float32')
	for k in range(T):
		sky = (150 + 0.05 * xx + 0.02 * yy) * (1 + 0.03 * np.sin(k / 3.0))
		sig = img * (1 + 1e-3 * rng.normal()) + sky
		noise = np.sqrt(sig + 100.0)
		raw[k] = sig + 25.0 + rng.normal(size=sig.shape) * noise     # +25: a residual the smoothed background does not remove
		err[k] = noise
	raw[rng.random(raw.shape) < 3e-4] = np.nan
	time = 1400.0 + np.arange(T) * 1800.0 / 86400.0
	quality = np.zeros(T, dtype='int32'); quality[7] = 32
	cat = {'starid': np.arange(len(stars), dtype='int64') + 501, 'tmag': np.array([s[2] for s in stars], dtype='float32'),
		'row': np.array([s[0] for s in stars], dtype='float32'), 'column': np.array([s[1] for s in stars], dtype='float32')}
	targets = {'starid': cat['starid'].copy(), 'tmag': np.array([s[2] for s in stars]), 'row': np.array([s[0] for s in stars]),
		'column': np.array([s[1] for s in stars])}
	return raw, err, row0, col0, time, quality, cat, targets


def test_raw_frames_to_light_curves():
	from photometry_amd import prepare, pipeline, tessphot_frames, STATUS
	from photometry_amd.device import Context
	from oracle import backgrounds as ob, aperture as oap, sumimage as osum
	raw, err, row0, col0, time, quality, cat, targets = _raw_region()
	T, R, C = raw.shape
	flags = np.zeros((T, R, C), dtype='uint8')
	flags[3, 38:42, 100:130] = 2           # a manually excluded strip in one frame (crosses nobody's mask centre)
	flags[11, :, 170:] = 2                 # and the last columns of another
	ctx = Context(0)
	d_raw, d_flags = ctx.array(raw), ctx.array(flags)
	out = prepare.prepare_frames(ctx, d_raw, ctx.array(err), quality, pixel_flags=d_flags)

	# ---- the oracle's prepare stage, from the device's unsmoothed backgrounds
	us = prepare.fit_background_frames(ctx, d_raw, exclude=d_flags).to_host()
	for k in (0, 11):
		ref_bkg, _ = ob.fit_background(raw[k], exclude=flags[k] != 0)
		np.testing.assert_allclose(us[k], ref_bkg, rtol=1e-6)
	bkg = np.moveaxis(ob.smooth_time(np.moveaxis(us, 0, -1), 3), -1, 0)
	img, img_err = ob.subtract_background(raw, err, bkg, flags)
	np.testing.assert_array_equal(out['backgrounds'].to_host(), bkg)
	np.testing.assert_array_equal(out['images'].to_host(), img)
	np.testing.assert_array_equal(out['images_err'].to_host(), img_err)
	np.testing.assert_array_equal(out['sumimage'].to_host(), osum.sumimage(np.moveaxis(img, 0, -1), quality))
	assert np.all(np.isnan(img[3, 38:42, 100:130])) and np.all(np.isnan(img[11, :, 170:]))

	# ---- photometry straight from the prepare stage's device arrays
	stack = pipeline.FrameStack(ctx, {k: out[k] for k in ('images', 'images_err', 'backgrounds')}, row0, col0)
	batch = tessphot_frames(ctx, stack, targets, cat, time, quality)
	frames = {'images': np.moveaxis(img, 0, 2), 'images_err': np.moveaxis(img_err, 0, 2), 'backgrounds': np.moveaxis(bkg, 0, 2)}
	n_ok = 0
	for i in range(len(targets['starid'])):
		b = batch[i]
		o = oap.photometry_on_frames(oap.FrameTarget(frames, row0, col0, quality, cat, int(targets['starid'][i]), float(targets['tmag'][i]),
			float(targets['row'][i]), float(targets['column'][i])))
		assert b.status.value == o['status'], (i, b.status, o['status'], b._details.get('errors'), o['errors'])
		assert tuple(b._details['stamp']) == tuple(o['stamp'])
		assert b._details['stamp_resizes'] == o['stamp_resizes']
		if 'mask' in o:
			np.testing.assert_array_equal(b.final_phot_mask, o['mask'])
			np.testing.assert_array_equal(b.lightcurve['flux'], o['flux'])
			np.testing.assert_array_equal(b.lightcurve['flux_err'], o['flux_err'])
			np.testing.assert_array_equal(b.lightcurve['flux_background'], o['flux_background'])
			np.testing.assert_allclose(b.lightcurve['pos_centroid'], o['pos_centroid'], rtol=1e-12, equal_nan=True)
			assert b._details.get('skip_targets', []) == o['skip_targets']
		n_ok += b.status in (STATUS.OK, STATUS.WARNING)
	assert n_ok >= 6
	# the close pair (3.8 pixels apart) is split by the watershed: two masks, neither swallows the other target
	assert batch[0]._details['mask_size'] > 0 and batch[1]._details['mask_size'] > 0
	assert not batch[0]._details.get('skip_targets') and not batch[1]._details.get('skip_targets')
	# the light curve of the brightest target follows its flux (sanity of the whole chain: sky gone, star kept)
	f = batch[4].lightcurve['flux']
	expect = 10**(-0.4 * (8.6 - 20.451))
	assert abs(np.nanmedian(f) / expect - 1) < 0.1
	ctx.close()
