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


def _bstar_scalar(x, flux_cutoff=8e4, frac=0.5):
	"""B*'s definition (oracle/backgrounds.py header) once more, one frame, plain Python loops: a second, independent writing of
	the same text against which the vectorised :func:`oracle.backgrounds.bstar_frames` is checked."""
	f64 = np.float64
	x = np.asarray(x, dtype='float32').ravel()
	P = len(x)
	kept = [v for v in x if (v >= 0) and (v <= np.float32(flux_cutoff))]
	n = len(kept)
	if n == 0 or np.float32(P - n) > np.float32(frac) * np.float32(P):
		return np.float32(np.nan)
	tree = lambda a: ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))
	a1, a2 = [f64(0)] * 8, [f64(0)] * 8
	for p in range(P):
		v = x[p]
		if (v >= 0) and (v <= np.float32(flux_cutoff)):
			a1[p % 8] = a1[p % 8] + f64(v)
			a2[p % 8] = a2[p % 8] + f64(v) * f64(v)
	s1, s2 = tree(a1), tree(a2)
	k = sorted(f64(v) for v in kept)
	lo, hi = 0, n
	for it in range(6):
		m = hi - lo
		m1 = lo + m // 2
		m0 = m1 if m % 2 else m1 - 1
		med = (k[m0] + k[m1]) * f64(0.5)
		if it == 5:
			break
		mm = f64(m)
		q9 = f64(9) * (mm * s2 - s1 * s1)
		if not q9 > 0:
			q9 = f64(0)
		na = 0
		while na < m and (k[hi - 1 - na] - med) * mm > 0 and ((k[hi - 1 - na] - med) * mm)**2 > q9:
			na += 1
		nb = 0
		while nb < m and (k[lo + nb] - med) * mm < 0 and ((k[lo + nb] - med) * mm)**2 > q9:
			nb += 1
		if na == 0 and nb == 0:
			break
		r1, r2 = [f64(0)] * 8, [f64(0)] * 8
		s = 0
		while 8 * s < max(na, nb):
			for g in range(8):
				t = 8 * s + g
				if t < na:
					r1[g] = r1[g] + k[hi - 1 - t]; r2[g] = r2[g] + k[hi - 1 - t] * k[hi - 1 - t]
				if t < nb:
					r1[g] = r1[g] + k[lo + t]; r2[g] = r2[g] + k[lo + t] * k[lo + t]
			s += 1
		s1, s2 = s1 - tree(r1), s2 - tree(r2)
		lo, hi = lo + nb, hi - na
	mm = f64(hi - lo)
	q = mm * s2 - s1 * s1
	e = s1 - mm * med
	mean = s1 / mm
	if not q > 0:
		return np.float32(mean)
	if e * e < f64(0.09) * q:
		return np.float32(f64(2.5) * med - f64(1.5) * mean)
	return np.float32(med)


def _frames(rng, F, P):
	X = rng.normal(100.0, 3.0, (F, P)).astype('float32')
	for f in range(F):
		k = f % 6
		if k == 0:
			idx = rng.choice(P, size=min(P, 25), replace=False); X[f, idx] += rng.uniform(50, 60000, len(idx)).astype('float32')
		elif k == 1:
			idx = rng.choice(P, size=min(P, 12), replace=False); X[f, idx] = rng.uniform(0, 20, len(idx)).astype('float32')
			idx = rng.choice(P, size=min(P, 12), replace=False); X[f, idx] += rng.uniform(100, 3000, len(idx)).astype('float32')
		elif k == 2:
			X[f] = np.round(X[f])
		elif k == 3:
			X[f, rng.choice(P, size=max(0, min(P, P // 2 + (f % 3) - 1)), replace=False)] = np.nan
		elif k == 4:
			X[f] = np.exp(rng.uniform(-20, 11, P)).astype('float32')
	return X


def test_bstar_definition_two_writings_agree():
	"""The vectorised definition against the plain-loop writing of the same text, bit for bit; both ends clipping, more than eight
	values a pass, ties, frames around the 50 % rule, stamps from 1 to 441 pixels."""
	rng = np.random.default_rng(7)
	for P in (225, 121, 256, 441, 30, 1, 9):
		X = _frames(rng, 60, P)
		got, st = ob.bstar_frames(X, full=True)
		ref = np.array([_bstar_scalar(X[f]) for f in range(len(X))], dtype='float32')
		np.testing.assert_array_equal(got, ref)
		if P >= 121:
			assert (st['passes'] >= 2).any() and (st['lo'] > 8).any() and (st['n'] - st['hi'] > 8).any()


def test_bstar_definition_is_the_literal_estimator_up_to_rounding():
	"""The defined arithmetic against the literal astropy / photutils statements with numpy's own summation orders: equal to
	float32 rounding on frames where no decision sits within rounding of its threshold (all of these)."""
	rng = np.random.default_rng(8)
	X = _frames(rng, 300, 225)
	got = ob.bstar_frames(X)
	lit = np.array([ob.fit_background_stamp_literal(X[f].reshape(15, 15))[0] for f in range(len(X))])
	np.testing.assert_array_equal(np.isnan(got), np.isnan(lit))
	np.testing.assert_allclose(got, lit.astype('float32'), rtol=2.4e-7, equal_nan=True)   # 2 ulp of float32


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
