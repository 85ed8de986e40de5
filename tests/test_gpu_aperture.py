# -*- coding: utf-8 -*-
"""
GPU parity tests (through the C ABI) of A1 (sum image) and A6 (aperture extraction) against
the oracle and against the golden vectors produced by the reference's own code.

Tolerances: float32 quantities the reference computes in float32 (flux = np.sum, flux_err,
background nansum) are compared BIT-EXACT; float64 centroids and the sum image to 1e-12 relative
(north_star allows 1e-5 on flux).
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


def _scene(nt, T, H, W, seed, **kw):
	from photometry_amd import simulate
	s = simulate.make_scene(nt, T, H, W, seed=seed, **kw)
	simulate.fill_cubes(s, nan_fraction=0.004)
	return s


@pytest.mark.parametrize("nt,T,H,W", [(24, 61, 11, 11), (12, 200, 15, 15), (5, 7, 9, 13), (3, 1300, 15, 15)])
def test_sumimage_parity(ctx, nt, T, H, W):
	from photometry_amd import engine
	from photometry_amd.device import DeviceCube
	from oracle import sumimage as osum
	s = _scene(nt, T, H, W, seed=100 + T)
	s.images[0, 1, 2, :] = np.nan # never observed -> NaN
	s.images[1 % nt, 0, 0, s.quality == 0] = np.nan # only bad-quality cadences finite -> NaN
	cube = DeviceCube.from_host(ctx, s.images)
	q = ctx.array(s.quality.astype('int32'))
	S = engine.sumimage(ctx, cube, q).to_host()
	ref = osum.sumimage_batch(s.images, s.quality)
	np.testing.assert_array_equal(np.isnan(S), np.isnan(ref))
	np.testing.assert_allclose(S, ref, rtol=1e-12, atol=0, equal_nan=True)
	# per-target quality table
	q2 = np.tile(s.quality, (nt, 1)).astype('int32')
	q2[0, :T//2] |= 4
	S2 = engine.sumimage(ctx, cube, ctx.array(q2)).to_host()
	ref0 = osum.sumimage(s.images[0], q2[0])
	np.testing.assert_allclose(S2[0], ref0, rtol=1e-12, equal_nan=True)
	np.testing.assert_allclose(S2[1:], ref[1:], rtol=1e-12, equal_nan=True)


def test_sumimage_golden(ctx, golden_dir):
	from photometry_amd import engine
	from photometry_amd.device import DeviceCube
	g = np.load(os.path.join(golden_dir, 'golden_sumimage.npz'))
	cube = DeviceCube.from_host(ctx, g['images'])
	S = engine.sumimage(ctx, cube, ctx.array(g['quality'].astype('int32'))).to_host()
	np.testing.assert_array_equal(np.isnan(S), np.isnan(g['sumimage']))
	np.testing.assert_allclose(S, g['sumimage'], rtol=1e-12, equal_nan=True)


def test_sumimage_unaligned_pitch(ctx):
	"""t_pitch not a multiple of 4 -> scalar load path, same answer."""
	from photometry_amd import engine
	from photometry_amd.device import DeviceCube
	from oracle import sumimage as osum
	s = _scene(4, 37, 7, 7, seed=7)
	cube = DeviceCube(ctx, 4, 37, 7, 7, t_pitch=37)
	ctx._check(ctx.lib.tp_upload_cube(ctx.handle, cube.ptr, 37, np.ascontiguousarray(s.images).ctypes.data, 37, 4*49, 37))
	S = engine.sumimage(ctx, cube, ctx.array(s.quality.astype('int32'))).to_host()
	np.testing.assert_allclose(S, osum.sumimage_batch(s.images, s.quality), rtol=1e-12, equal_nan=True)


def _masks_for(s, kind, rng):
	"""(Nt, H, W) bool masks of various sizes incl. 0, <8, 8..128, >128 pixels."""
	Nt, H, W = s.n_targets, s.height, s.width
	m = np.zeros((Nt, H, W), dtype=bool)
	for i in range(Nt):
		if kind == 'small':
			n = [0, 1, 5, 7, 8, 9, 15, 16, 17, 31, 64, 100, 127, 128][i % 14]
		else:
			n = [129, 130, 136, 200, 255, 256, 257, 300, 511, 777, H*W][i % 11]
		n = min(n, H*W)
		idx = rng.choice(H*W, n, replace=False)
		m[i].flat[idx] = True
	return m


def _check_extract(ctx, s, masks, vec4=True, series_bkg=False):
	from photometry_amd import engine
	from photometry_amd.device import DeviceCube
	from oracle import aperture as oap
	Nt, T, H, W = s.n_targets, s.n_cad, s.height, s.width
	if vec4:
		img, err = DeviceCube.from_host(ctx, s.images), DeviceCube.from_host(ctx, s.images_err)
		bkg_cube = DeviceCube.from_host(ctx, s.backgrounds)
	else:
		def up(a):
			c = DeviceCube(ctx, Nt, T, H, W, t_pitch=T)
			ctx._check(ctx.lib.tp_upload_cube(ctx.handle, c.ptr, T, np.ascontiguousarray(a, dtype='float32').ctypes.data, T, Nt*H*W, T))
			return c
		img, err, bkg_cube = up(s.images), up(s.images_err), up(s.backgrounds)
	backgrounds = s.backgrounds
	if series_bkg:
		ser = np.ascontiguousarray(s.backgrounds[:, 0, 0, :])
		pitch = img.t_pitch
		serp = np.zeros((Nt, pitch), dtype='float32')
		serp[:, :T] = ser
		bkg = ctx.array(serp)
		backgrounds = np.broadcast_to(ser[:, None, None, :], s.backgrounds.shape)
	else:
		bkg = bkg_cube
	lc = engine.aperture_extract(ctx, img, err, bkg, ctx.array(masks.astype('uint8')), ctx.array(s.stamps.astype('int32'))).to_host()
	for i in range(Nt):
		ref = oap.extract(s.images[i], s.images_err[i], backgrounds[i], masks[i], tuple(s.stamps[i]))
		np.testing.assert_array_equal(lc['flux'][i], ref['flux'], err_msg=f"flux target {i} M={masks[i].sum()}")
		np.testing.assert_array_equal(lc['flux_err'][i], ref['flux_err'], err_msg=f"flux_err target {i}")
		np.testing.assert_array_equal(lc['flux_background'][i], ref['flux_background'], err_msg=f"bkg target {i}")
		np.testing.assert_allclose(lc['pos_centroid'][i], ref['pos_centroid'], rtol=1e-12, equal_nan=True)
		np.testing.assert_array_equal(np.isnan(lc['pos_centroid'][i]), np.isnan(ref['pos_centroid']))
	return lc


def test_extract_small_masks(ctx):
	rng = np.random.default_rng(1)
	s = _scene(28, 61, 12, 13, seed=11)
	_check_extract(ctx, s, _masks_for(s, 'small', rng))


def test_extract_big_masks(ctx):
	rng = np.random.default_rng(2)
	s = _scene(11, 45, 36, 40, seed=12)
	_check_extract(ctx, s, _masks_for(s, 'big', rng))


def test_extract_very_big_mask(ctx):
	"""> kChunk mask pixels: several staging rounds and a deep pairwise tree."""
	rng = np.random.default_rng(3)
	s = _scene(3, 21, 60, 64, seed=13)
	m = np.zeros((3, 60, 64), dtype=bool)
	m[0].flat[rng.choice(60*64, 1025, replace=False)] = True
	m[1].flat[rng.choice(60*64, 2500, replace=False)] = True
	m[2][:] = True
	_check_extract(ctx, s, m)


def test_extract_special_frames(ctx):
	rng = np.random.default_rng(4)
	s = _scene(6, 40, 11, 11, seed=14)
	m = _masks_for(s, 'small', rng)
	m[:, 4:7, 4:7] = True
	for i in range(6):
		mm = m[i]
		s.images[i, :, :, 2][mm] = np.nan
		s.images[i, :, :, 3][mm] = 0
		s.images[i, :, :, 4][mm] = -np.abs(s.images[i, :, :, 4][mm]) - 1
		s.backgrounds[i, :, :, 5][mm] = np.nan
		s.backgrounds[i, 5, 5, 6] = np.nan
		s.images[i, 5, 5, 7] = np.nan
		s.images_err[i, 5, 5, 8] = np.nan
		s.images[i, 5, 5, 9] = np.inf
	lc = _check_extract(ctx, s, m)
	assert np.all(np.isnan(lc['flux'][:, 2])) and np.all(np.isnan(lc['flux'][:, 3]))
	assert np.all(np.isfinite(lc['flux'][:, 4])) and np.all(np.isnan(lc['pos_centroid'][:, 4]))
	assert np.all(np.isnan(lc['flux_background'][:, 5])) and np.all(np.isfinite(lc['flux_background'][:, 6]))


def test_extract_scalar_path_and_series_background(ctx):
	rng = np.random.default_rng(5)
	s = _scene(14, 37, 9, 10, seed=15)
	_check_extract(ctx, s, _masks_for(s, 'small', rng), vec4=False)
	_check_extract(ctx, s, _masks_for(s, 'small', rng), vec4=True, series_bkg=True)


def test_extract_status_skip(ctx):
	from photometry_amd import engine
	from photometry_amd.device import DeviceCube
	rng = np.random.default_rng(6)
	s = _scene(4, 16, 9, 9, seed=16)
	m = _masks_for(s, 'small', rng)
	m[:, 3:6, 3:6] = True
	img, err, bkg = (DeviceCube.from_host(ctx, a) for a in (s.images, s.images_err, s.backgrounds))
	status = ctx.array(np.array([1, 2, 3, 2], dtype='int32'))
	lc = engine.aperture_extract(ctx, img, err, bkg, ctx.array(m.astype('uint8')), ctx.array(s.stamps.astype('int32')), status=status).to_host()
	assert np.all(lc['flux'][1] == 0) and np.all(lc['flux'][3] == 0) # ERROR targets untouched (zero-initialised)
	assert np.any(lc['flux'][0] != 0) and np.any(lc['flux'][2] != 0)


def test_extract_golden(ctx, golden_dir):
	"""The reference's own do_photometry outputs (golden) through the device kernel."""
	from photometry_amd import engine
	from photometry_amd.device import DeviceCube
	g = np.load(os.path.join(golden_dir, 'golden_aperture.npz'))
	img, err, bkg = (DeviceCube.from_host(ctx, g[k]) for k in ('images', 'images_err', 'backgrounds'))
	stamps = ctx.array(g['stamps'].astype('int32'))
	Nt, H, W = g['images'].shape[:3]
	n_checked = 0
	for n in range(int(g['n_cases'])):
		if not bool(g[f'case{n}_has_mask']):
			continue
		i = int(g[f'case{n}_target'])
		masks = np.zeros((Nt, H, W), dtype='uint8')
		masks[i] = g[f'case{n}_final_mask']
		lc = engine.aperture_extract(ctx, img, err, bkg, ctx.array(masks), stamps).to_host()
		np.testing.assert_array_equal(lc['flux'][i], g[f'case{n}_flux'])
		np.testing.assert_array_equal(lc['flux_err'][i], g[f'case{n}_flux_err'])
		np.testing.assert_allclose(lc['pos_centroid'][i], g[f'case{n}_pos_centroid'], rtol=1e-12, equal_nan=True)
		np.testing.assert_array_equal(lc['flux_background'][i], g[f'case{n}_flux_background']) # np.nansum, float32 pairwise
		n_checked += 1
	assert n_checked >= 8


def test_extract_empty_batch_and_errors(ctx):
	from photometry_amd import engine
	from photometry_amd.device import DeviceCube
	from photometry_amd._lib import TessphotError
	img = DeviceCube(ctx, 0, 8, 5, 5)
	lc = engine.aperture_extract(ctx, img, img, img, ctx.zeros((0, 5, 5), 'uint8'), ctx.zeros((0, 4), 'int32'))
	assert lc.to_host()['flux'].shape == (0, 8)
	# bad descriptor -> error code + message, no crash
	import ctypes
	from photometry_amd._lib import tp_cube_desc
	bad = tp_cube_desc(1, 8, 5, 5, 4)
	with pytest.raises(TessphotError) as e:
		ctx._check(ctx.lib.tp_sumimage(ctx.handle, ctypes.byref(bad), img.ptr, img.ptr, 0, 4335, None, 0, img.ptr))
	assert 'descriptor' in str(e.value)
