# -*- coding: utf-8 -*-
"""
GPU parity of B* (stamp background), B2 (time smoothing), B3 (subtraction) and of the on-the-fly
subtraction inside A1 / A6.

B* is build-defined and defined down to the last bit (oracle/backgrounds.py, module header: which float64 operations in
which order); the kernels implement that arithmetic, so B* is asserted BIT FOR BIT like B2 / B3.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
	from photometry_amd.device import Context
	c = Context(0)
	yield c
	c.close()


def _scene(nt, T, H, W, seed):
	from photometry_amd import simulate
	s = simulate.make_scene(nt, T, H, W, seed=seed)
	simulate.fill_cubes(s, nan_fraction=0.004, with_raw=True)
	return s


@pytest.mark.parametrize("nt,T,H,W", [(6, 70, 15, 15), (5, 33, 11, 11), (3, 9, 16, 16), (2, 12, 20, 21), (4, 40, 6, 5), (2, 7, 2, 2), (3, 37, 25, 25)])
def test_background_stamp_parity(ctx, nt, T, H, W):
	from photometry_amd import engine
	from photometry_amd.device import DeviceCube
	from oracle import backgrounds as ob
	s = _scene(nt, T, H, W, seed=200 + H)
	raw = s.raw.copy()
	raw[0, :, :, 1] = 1000.0                      # constant frame -> exactly 1000 (reference known answer)
	raw[0, :, :, 2] = np.nan                      # all masked -> NaN
	raw[0, :H//2 + 1, :, 3] = -5.0                # > 50 % masked -> NaN
	raw[0, 0, 0, 4] = 9e4; raw[0, 0, 1, 4] = np.inf; raw[0, 1, 0, 4] = -1.0
	bkg = engine.background_stamp(ctx, DeviceCube.from_host(ctx, raw)).to_host()[:, :T]
	ref = np.stack([ob.background_series(raw[i]) for i in range(nt)])
	np.testing.assert_array_equal(bkg, ref)       # bit for bit, NaN positions included
	assert bkg[0, 1] == 1000.0 and np.isnan(bkg[0, 2]) and np.isnan(bkg[0, 3])
	# the estimate tracks the injected background (level +-5 % sinusoid), stars clipped away
	truth = s.backgrounds[:, 0, 0, :]
	ok = np.isfinite(bkg) & (np.arange(T)[None, :] > 4)
	if H * W >= 121:
		assert np.nanmedian(np.abs(bkg[ok] / truth[ok] - 1)) < 0.03


def _hard_frames(rng, F, P):
	"""Frames that exercise every branch of the clipping: deep clips at both ends (more than eight values a pass: several steps of
	the walk), values on both sides of the thresholds within a few ulp, ties, constant frames, frames near the 50 % rule."""
	X = rng.normal(100.0, 3.0, (F, P)).astype('float32')
	kind = rng.integers(0, 8, F)
	for f in range(F):
		k = kind[f]
		if k == 0:      # a bright star: 10-40 pixels far above the sky
			idx = rng.choice(P, size=min(P, int(rng.integers(10, 41))), replace=False)
			X[f, idx] += rng.uniform(50, 60000, len(idx)).astype('float32')
		elif k == 1:    # low outliers as well (cold pixels): both ends clip, many at a time
			idx = rng.choice(P, size=min(P, int(rng.integers(9, 30))), replace=False)
			X[f, idx] = rng.uniform(0, 20, len(idx)).astype('float32')
			idx = rng.choice(P, size=min(P, int(rng.integers(9, 30))), replace=False)
			X[f, idx] += rng.uniform(100, 3000, len(idx)).astype('float32')
		elif k == 2:    # heavy ties: quantised values
			X[f] = np.round(X[f])
		elif k == 3:    # constant but for a few
			X[f] = np.float32(rng.uniform(1, 7e4))
			X[f, rng.choice(P, size=min(P, 3), replace=False)] *= np.float32(1.0001)
		elif k == 4:    # masked pixels close to the 50 % rule
			nm = min(P, P // 2 + int(rng.integers(-2, 3)))
			if nm > 0:
				X[f, rng.choice(P, size=nm, replace=False)] = [np.nan, -1.0, 9e4, np.inf][int(rng.integers(0, 4))]
		elif k == 5:    # a value a few ulp either side of med + 3 std of the first pass
			x64 = X[f].astype('float64')
			thr = np.median(x64) + 3 * x64.std()
			X[f, 0] = np.nextafter(np.float32(thr), np.float32(np.inf if rng.integers(0, 2) else -np.inf))
		elif k == 6:    # wide dynamic range
			X[f] = np.exp(rng.uniform(-20, 11, P)).astype('float32')
	return X


@pytest.mark.parametrize("H,W", [(15, 15), (11, 11), (16, 16), (3, 3), (20, 21)])
def test_background_stamp_bit_exact_on_hard_frames(ctx, H, W):
	"""B* bit for bit on 4 x 1300 frames built to hit every decision (see _hard_frames); 20 x 21 goes through the generic kernel."""
	from photometry_amd import engine
	from photometry_amd.device import DeviceCube
	from oracle import backgrounds as ob
	rng = np.random.default_rng(1000 + H * W)
	nt, T, P = 4, 1300, H * W
	X = _hard_frames(rng, nt * T, P)
	raw = np.ascontiguousarray(np.moveaxis(X.reshape(nt, T, H, W), 1, 3))
	bkg = engine.background_stamp(ctx, DeviceCube.from_host(ctx, raw)).to_host()[:, :T]
	ref, st = ob.bstar_frames(X, full=True)
	np.testing.assert_array_equal(bkg.reshape(-1), ref)
	if P >= 121:
		assert (st['passes'] >= 3).sum() > 100 and ((st['lo'] > 8) & (st['n'] - st['hi'] > 8)).sum() > 20   # the branches were taken
	# and the one-pass kernel's series is the same series
	q = ctx.array(np.zeros(T, dtype='int32'))
	b_raw, _, _ = engine.background_sumimage(ctx, DeviceCube.from_host(ctx, raw), q, 3)
	np.testing.assert_array_equal(b_raw.to_host()[:, :T].reshape(-1), ref)


@pytest.mark.parametrize("nt,T,H,W", [(6, 70, 15, 15), (5, 33, 11, 11), (3, 9, 16, 16), (4, 40, 6, 5), (3, 32, 15, 15), (2, 64, 13, 9), (2, 200, 17, 17)])
@pytest.mark.parametrize("time_smooth", [1, 3, 9, 17, 19])
def test_background_sumimage_one_pass(ctx, nt, T, H, W, time_smooth):
	"""tp_background_sumimage: the two series bit for bit those of tp_background_stamp + tp_smooth_time (series shorter than the
	window, one-block series, blocks whose last cadences wait for the next block, stamps above 256 pixels and windows above 17
	through the three-entry route), the sum image within rounding of tp_sumimage and 1e-12 of the oracle."""
	from photometry_amd import engine
	from photometry_amd.device import DeviceCube
	from oracle import sumimage as osum, backgrounds as ob
	s = _scene(nt, T, H, W, seed=300 + H + T)
	raw = s.raw.copy()
	raw[0, :, :, 1] = np.nan                       # a cadence without estimate inside the windows of its neighbours
	if T > 40:
		raw[1, :, :, 30:34] = np.nan               # ... across a block boundary
	q = s.quality.astype('int32').copy()
	q[T // 2] |= 32
	q[T - 1] |= 4
	cube = DeviceCube.from_host(ctx, raw)
	dq = ctx.array(q)
	b_raw, b_smooth, S = engine.background_sumimage(ctx, cube, dq, time_smooth)
	ref_raw = engine.background_stamp(ctx, cube)
	ref_smooth = engine.smooth_time(ctx, ref_raw, T, time_smooth)
	ref_S = engine.sumimage(ctx, cube, dq, subtract=ref_smooth)
	np.testing.assert_array_equal(b_raw.to_host()[:, :T], ref_raw.to_host()[:, :T])
	np.testing.assert_array_equal(b_smooth.to_host()[:, :T], ref_smooth.to_host()[:, :T])
	np.testing.assert_array_equal(ref_smooth.to_host()[:, :T], ob.smooth_time(ref_raw.to_host()[:, :T], time_smooth))
	got = S.to_host()
	np.testing.assert_allclose(got, ref_S.to_host(), rtol=1e-13, atol=0, equal_nan=True)
	diff = raw - b_smooth.to_host()[:, None, None, :T]
	np.testing.assert_allclose(got, osum.sumimage_batch(diff, q), rtol=1e-12, atol=0, equal_nan=True)
	# run to run: fixed summation order, no atomics
	_, _, S2 = engine.background_sumimage(ctx, cube, dq, time_smooth)
	np.testing.assert_array_equal(S2.to_host(), got)


def test_smooth_and_subtract(ctx):
	from photometry_amd import engine
	from photometry_amd.device import DeviceCube
	from oracle import backgrounds as ob
	s = _scene(5, 61, 9, 10, seed=31)
	rng = np.random.default_rng(3)
	T = 61
	series = rng.normal(100, 3, (5, 64)).astype('float32')
	series[0, 7] = np.nan; series[1, :4] = np.nan; series[2, 20:31] = np.nan
	d = ctx.array(series)
	for ts in (3, 9):
		out = engine.smooth_time(ctx, d, T, ts).to_host()[:, :T]
		ref = ob.smooth_time(series[:, :T], ts)
		np.testing.assert_array_equal(out, ref) # float32, bit-exact incl. NaN positions
	# B3, with manual-exclude flags, out of place and in place
	flags = np.zeros((5, 9, 10, T), dtype='uint8')
	flags[1, 2, 3, 5] = 2; flags[0, 0, 0, 0] = 1; flags[4, 8, 9, 60] = 3
	raw, err = DeviceCube.from_host(ctx, s.raw), DeviceCube.from_host(ctx, s.raw_err)
	img, ierr = DeviceCube(ctx, 5, T, 9, 10), DeviceCube(ctx, 5, T, 9, 10)
	bser = series.copy(); bser[np.isnan(bser)] = 90.0
	dser = ctx.array(bser)
	engine.subtract_background(ctx, raw, dser, raw_err=err, pixel_flags=ctx.array(flags.reshape(5, 90, T)), images=img, images_err=ierr)
	rimg, rerr = ob.subtract_background(s.raw, s.raw_err, bser[:, None, None, :T], flags)
	np.testing.assert_array_equal(img.to_host(), rimg)
	np.testing.assert_array_equal(ierr.to_host(), rerr)
	engine.subtract_background(ctx, raw, dser, raw_err=err) # in place, no flags
	rimg2, _ = ob.subtract_background(s.raw, s.raw_err, bser[:, None, None, :T])
	np.testing.assert_array_equal(raw.to_host(), rimg2)


def test_on_the_fly_subtraction_equals_materialised(ctx):
	"""A1 / A6 with subtract= series == A1 / A6 on the cube produced by B3 (bit for bit)."""
	from photometry_amd import engine
	from photometry_amd.device import DeviceCube
	s = _scene(8, 52, 11, 11, seed=41)
	rng = np.random.default_rng(4)
	raw, err = DeviceCube.from_host(ctx, s.raw), DeviceCube.from_host(ctx, s.raw_err)
	bkg = engine.smooth_time(ctx, engine.background_stamp(ctx, raw), 52, 3)
	q = ctx.array(s.quality.astype('int32'))
	m = (rng.random((8, 11, 11)) < 0.2)
	m[:, 5, 5] = True
	mask, stamps = ctx.array(m.astype('uint8')), ctx.array(s.stamps.astype('int32'))
	S_fly = engine.sumimage(ctx, raw, q, subtract=bkg).to_host()
	lc_fly = engine.aperture_extract(ctx, raw, err, bkg, mask, stamps, subtract=bkg).to_host()
	img = DeviceCube(ctx, 8, 52, 11, 11)
	engine.subtract_background(ctx, raw, bkg, images=img)
	S_mat = engine.sumimage(ctx, img, q).to_host()
	lc_mat = engine.aperture_extract(ctx, img, err, bkg, mask, stamps).to_host()
	np.testing.assert_array_equal(S_fly, S_mat)
	for key in ('flux', 'flux_err', 'flux_background', 'pos_centroid'):
		np.testing.assert_array_equal(lc_fly[key], lc_mat[key])
	# and against the oracle on the materialised cube
	from oracle import aperture as oap
	imgh = img.to_host()
	b = bkg.to_host()[:, :52]
	for i in range(8):
		ref = oap.extract(imgh[i], s.raw_err[i], np.broadcast_to(b[i][None, None, :], imgh[i].shape), m[i], tuple(s.stamps[i]))
		np.testing.assert_array_equal(lc_fly['flux'][i], ref['flux'])
		np.testing.assert_array_equal(lc_fly['flux_background'][i], ref['flux_background'])


def test_b2_b3_golden_on_device(ctx, golden_dir):
	"""The reference's own smoothing / subtraction statements (golden_background.npz, prepare.py:317-335, 419-425) through
	tp_smooth_time and tp_subtract_background: every pixel of the small frame stack is one 1x1 'target'."""
	import os
	from photometry_amd import engine
	from photometry_amd.device import DeviceCube
	g = np.load(os.path.join(golden_dir, 'golden_background.npz'))
	frames = g['b2_frames']
	N, H, W = frames.shape
	series = np.zeros((H * W, 32), dtype='float32')
	series[:, :N] = frames.reshape(N, H * W).T
	d = ctx.array(series)
	sm = {}
	for ts in (3, 9):
		sm[ts] = engine.smooth_time(ctx, d, N, ts)
		got = sm[ts].to_host()[:, :N].T.reshape(N, H, W)
		np.testing.assert_array_equal(got, g[f'b2_ts{ts}_smoothed'])
	def as_cube(a):   # (N, H, W) -> (H*W targets, 1, 1, N)
		return np.ascontiguousarray(a.reshape(N, H * W).T.reshape(H * W, 1, 1, N).astype('float32'))
	raw, err = DeviceCube.from_host(ctx, as_cube(g['b3_raw'])), DeviceCube.from_host(ctx, as_cube(g['b3_raw_err']))
	img, ierr = DeviceCube(ctx, H * W, N, 1, 1), DeviceCube(ctx, H * W, N, 1, 1)
	flags = ctx.array(np.ascontiguousarray(g['b3_flags'].reshape(N, H * W).T.reshape(H * W, 1, N).astype('uint8')))
	engine.subtract_background(ctx, raw, sm[3], raw_err=err, pixel_flags=flags, images=img, images_err=ierr)
	np.testing.assert_array_equal(img.to_host(), as_cube(g['b3_images_backapp0']))
	np.testing.assert_array_equal(ierr.to_host(), as_cube(g['b3_errors_backapp0']))
