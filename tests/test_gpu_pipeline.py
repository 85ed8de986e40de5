# -*- coding: utf-8 -*-
"""
GPU tests of the batched aperture pipeline: the fused per-target kernel (tp_aperture_photometry) must give
bit-identical outputs to the three stand-alone kernels (same device functions, same per-target arithmetic) and
match the oracle; target-chunk views on several HIP streams ordered by events must equal the full batch.
"""
import numpy as np
import pytest
from photometry_amd import simulate, pipeline

pytestmark = pytest.mark.gpu

KEYS = ('sumimage', 'mask', 'status', 'flags', 'contamination', 'cat_in_mask', 'flux', 'flux_err', 'flux_background',
	'centroid_col', 'centroid_row')


@pytest.fixture(scope='module')
def ctx():
	from photometry_amd.device import Context
	c = Context(0)
	yield c
	c.close()


def _collect(work):
	out = work.lc.to_host()
	for k in ('sumimage', 'mask', 'status', 'flags', 'contamination', 'cat_in_mask'):
		out[k] = getattr(work, k).to_host()
	return out


@pytest.mark.parametrize('n_chunks', [2, 3, 7])
def test_chunk_views_on_two_streams(ctx, n_chunks):
	"""Target chunks (non-owning views of the same HBM) processed alternately on two contexts (= two HIP streams) ordered by
	events give exactly the full-batch result."""
	from photometry_amd.device import Context
	s = simulate.make_scene(37, 203, 15, 15, seed=11)
	simulate.fill_cubes(s)
	s.aperture = None
	batch = pipeline.ApertureBatch(ctx, s)
	ref_work = pipeline.ApertureWork(ctx, batch)
	pipeline.aperture_step(ctx, batch, ref_work)
	ctx.sync()
	ref = _collect(ref_work)

	work = pipeline.ApertureWork(ctx, batch)
	other = Context(ctx.device, high_priority=True)
	bounds = [(37 * i) // n_chunks for i in range(n_chunks + 1)]
	begin, done = ctx.event(), ctx.event()
	ctx.record(begin)
	other.wait_event(begin)
	for c, (a, b) in enumerate(zip(bounds[:-1], bounds[1:])):
		stream = ctx if c % 2 == 0 else other
		pipeline.aperture_step(stream, batch.chunk(a, b - a), work.chunk(a, b - a), fused=(c % 3 != 0))
	other.record(done)
	ctx.wait_event(done)
	ctx.sync()
	got = _collect(work)
	other.close()
	for k in KEYS:
		np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


def test_fused_against_oracle(ctx):
	from oracle import aperture as oap
	s = simulate.make_scene(10, 60, 13, 17, seed=4)
	simulate.fill_cubes(s)
	batch = pipeline.ApertureBatch(ctx, s)
	work = pipeline.ApertureWork(ctx, batch)
	pipeline.aperture_step(ctx, batch, work)
	ctx.sync()
	got = _collect(work)
	ap = np.ones((13, 17), dtype='int32')
	for i in range(s.n_targets):
		ref = oap.do_photometry(got['sumimage'][i], s.images[i], s.images_err[i], s.backgrounds[i], tuple(s.stamps[i]),
			s.target_pos_row[i], s.target_pos_column[i], s.target_tmag[i], s.target_starid[i], s.catalog_of(i), ap)
		assert int(got['status'][i]) == ref['status']
		if ref['mask'] is not None:
			np.testing.assert_array_equal(got['mask'][i].astype(bool), ref['mask'])
			np.testing.assert_array_equal(got['flux'][i], ref['flux'])
			np.testing.assert_array_equal(got['flux_err'][i], ref['flux_err'])


@pytest.mark.parametrize('shape', [(41, 203, 15, 15), (9, 64, 11, 13), (5, 37, 7, 7), (3, 1301, 21, 19)])
def test_fused_equals_three_kernels(ctx, shape):
	"""tp_aperture_photometry (one wavefront per target) against tp_sumimage + tp_k2p2_masks + tp_aperture_extract."""
	n, T, H, W = shape
	s = simulate.make_scene(n, T, H, W, seed=n + T)
	simulate.fill_cubes(s)
	s.aperture = None
	batch = pipeline.ApertureBatch(ctx, s, cubes='host')
	ref_work = pipeline.ApertureWork(ctx, batch)
	pipeline.aperture_step(ctx, batch, ref_work, fused=False)
	ctx.sync()
	ref = _collect(ref_work)
	work = pipeline.ApertureWork(ctx, batch)
	pipeline.aperture_step(ctx, batch, work, fused=True)
	ctx.sync()
	got = _collect(work)
	assert (ref['status'] != 2).any()
	for k in KEYS:
		np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


@pytest.mark.parametrize('shape', [(41, 203, 15, 15), (9, 64, 11, 13), (5, 37, 7, 7), (3, 1301, 21, 19), (6, 97, 11, 11), (4, 1300, 15, 15)])
@pytest.mark.parametrize('time_smooth', [3, 9])
def test_raw_step_reads_the_cube_once(ctx, shape, time_smooth):
	"""The step on RAW cubes: tp_background_sumimage (B* + B2 + A1 in one pass) + tp_aperture_photometry_from_sumimage against
	the stand-alone entries.  The two background series are bit-identical to tp_background_stamp / tp_smooth_time; the sum image
	equals tp_sumimage's to rounding (another order of the float64 additions) and the oracle's to 1e-12; and from THAT sum image
	the fused launch is bit-identical to tp_k2p2_masks + tp_aperture_extract."""
	from photometry_amd import engine
	from oracle import sumimage as osum
	n, T, H, W = shape
	s = simulate.make_scene(n, T, H, W, seed=n + T)
	simulate.fill_cubes(s, with_raw=True)
	s.aperture = None
	s.cadence_s = {3: 1800, 9: 600}[time_smooth]
	batch = pipeline.ApertureBatch(ctx, s, cubes='host_raw')
	assert batch.time_smooth == time_smooth
	ref_work = pipeline.ApertureWork(ctx, batch)
	pipeline.aperture_step(ctx, batch, ref_work, fused=False)     # B*, B2, A1, K2P2, A6: five launches, the cube read twice
	ctx.sync()
	ref = _collect(ref_work)
	work = pipeline.ApertureWork(ctx, batch)
	pipeline.aperture_step(ctx, batch, work, fused=True)
	ctx.sync()
	got = _collect(work)
	np.testing.assert_array_equal(work.bkg_raw.to_host()[:, :T], ref_work.bkg_raw.to_host()[:, :T])
	np.testing.assert_array_equal(work.bkg.to_host()[:, :T], ref_work.bkg.to_host()[:, :T])
	from oracle import backgrounds as ob
	for i in range(min(n, 6)):   # ... and the oracle's own series, bit for bit (B* is defined to the last bit)
		braw = ob.background_series(s.raw[i])
		np.testing.assert_array_equal(work.bkg_raw.to_host()[i, :T], braw)
		np.testing.assert_array_equal(work.bkg.to_host()[i, :T], ob.smooth_time(braw, time_smooth))
	np.testing.assert_allclose(got['sumimage'], ref['sumimage'], rtol=1e-13, atol=0, equal_nan=True)
	diff = s.raw - work.bkg.to_host()[:, None, None, :T]           # float32, prepare.py:421
	np.testing.assert_allclose(got['sumimage'], osum.sumimage_batch(diff, s.quality), rtol=1e-12, atol=0, equal_nan=True)
	# the rest of the step from the one-pass sum image, stage by stage
	ref_work.sumimage = work.sumimage
	engine.k2p2_masks(ctx, batch, ref_work)
	engine.aperture_extract(ctx, batch.images, batch.images_err, work.bkg, ref_work.mask, batch.stamps, status=ref_work.status, out=ref_work.lc, subtract=work.bkg)
	ctx.sync()
	ref = _collect(ref_work)
	assert (ref['status'] != 2).any()
	for k in KEYS:
		np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


def test_fused_large_masks_and_unaligned_pitch(ctx):
	"""A saturated star gives a mask above 128 pixels (big kernel after the fused one); odd T -> scalar path when the pitch is unpadded."""
	from photometry_amd.device import DeviceCube
	s = simulate.make_scene(4, 51, 25, 25, seed=77, tmag_range=(3.0, 4.5))
	simulate.fill_cubes(s)
	s.aperture = None
	for pitch in (None, 51):
		cubes = {}
		for name in ('images', 'images_err', 'backgrounds'):
			host = getattr(s, name)
			if pitch is None:
				cubes[name] = DeviceCube.from_host(ctx, host)
			else:
				d = DeviceCube(ctx, 4, 51, 25, 25, t_pitch=pitch)
				ctx._check(ctx.lib.tp_upload_cube(ctx.handle, d.ptr, pitch, np.ascontiguousarray(host).ctypes.data, 51, 4*25*25, 51))
				cubes[name] = d
		batch = pipeline.ApertureBatch(ctx, s, cubes=cubes)
		ref_work = pipeline.ApertureWork(ctx, batch)
		pipeline.aperture_step(ctx, batch, ref_work, fused=False)
		work = pipeline.ApertureWork(ctx, batch)
		pipeline.aperture_step(ctx, batch, work, fused=True)
		ctx.sync()
		ref, got = _collect(ref_work), _collect(work)
		for k in KEYS:
			np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


def test_fused_with_per_target_quality(ctx):
	"""quality as (Nt, T): every target has its own good-cadence set (A1 in the fused kernel and the diagnostics)."""
	from oracle import sumimage as osum, diagnostics as odiag
	s = simulate.make_scene(7, 90, 11, 11, seed=14)
	simulate.fill_cubes(s)
	s.aperture = None
	rng = np.random.default_rng(2)
	q = np.zeros((7, 90), dtype='int32')
	q[rng.random((7, 90)) < 0.2] = 32
	q[3, :] = 0
	s.quality = q
	batch = pipeline.ApertureBatch(ctx, s)
	ref_work = pipeline.ApertureWork(ctx, batch)
	pipeline.aperture_step(ctx, batch, ref_work, fused=False)
	work = pipeline.ApertureWork(ctx, batch)
	pipeline.aperture_step(ctx, batch, work, fused=True)
	pipeline.aperture_diagnostics(ctx, batch, work)
	ctx.sync()
	ref, got = _collect(ref_work), _collect(work)
	for k in KEYS:
		np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
	np.testing.assert_allclose(got['sumimage'], osum.sumimage_batch(s.images, q), rtol=1e-12, equal_nan=True)
	d = work.diagnostics.to_host()
	for i in range(7):
		if int(got['status'][i]) not in (1, 3):
			continue
		cen = np.stack((got['centroid_col'][i], got['centroid_row'][i]), axis=-1)
		o = odiag.diagnostics(s.time, q[i], got['flux'][i], got['flux_err'][i], cen, sumimage=got['sumimage'][i], mask=got['mask'][i])
		assert d[i][0] == o['mean_flux'] and d[i][3] == o['ptp']
		np.testing.assert_allclose(d[i][2], o['rms_hour'], rtol=1e-12)
