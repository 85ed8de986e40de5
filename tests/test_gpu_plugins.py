# -*- coding: utf-8 -*-
"""
GPU tests of the drop-in plugin API: they read like the reference's tests/test_aperturephotometry.py
(status in {OK, WARNING}; no NaN in the sum image; flux / err / centroid not all 0 / NaN; aperture bits
2 and 8 set in the saved file; header keys) plus parity against the oracle, the stamp-resize retry
loop, tessphot() dispatch and the batch API.
"""
import os
import numpy as np
import pytest
from photometry_amd import STATUS, tessphot, tessphot_batch, simulate
from photometry_amd.plugins import AperturePhotometry, LinPSFPhotometry
from photometry_amd.source import MemoryStampSource, source_from_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
	from photometry_amd.device import Context
	c = Context(0)
	yield c
	c.close()


def _scene(n=6, T=50, H=15, W=15, seed=5, **kw):
	s = simulate.make_scene(n, T, H, W, seed=seed, **kw)
	simulate.fill_cubes(s)
	return s


def test_aperturephotometry_like_the_reference(ctx, tmp_path):
	s = _scene()
	from oracle import aperture as oap
	for i in range(s.n_targets):
		src = source_from_scene(s, i)
		with AperturePhotometry(int(s.target_starid[i]), src, str(tmp_path), datasource='ffi', ctx=ctx) as pho:
			pho.photometry()
			assert pho.status in (STATUS.OK, STATUS.WARNING)
			assert not np.any(np.isnan(pho.sumimage))
			assert not np.all(pho.lightcurve['flux'] == 0) and not np.all(np.isnan(pho.lightcurve['flux']))
			assert not np.all(pho.lightcurve['flux_err'] == 0) and not np.all(np.isnan(pho.lightcurve['pos_centroid']))
			# parity with the oracle (same stamp, fixed cube)
			ref = oap.do_photometry(pho.sumimage, s.images[i], s.images_err[i], s.backgrounds[i], tuple(s.stamps[i]),
				s.target_pos_row[i], s.target_pos_column[i], s.target_tmag[i], s.target_starid[i], s.catalog_of(i), pho.aperture)
			assert pho.status.value == ref['status']
			np.testing.assert_array_equal(pho.final_phot_mask, ref['mask'])
			np.testing.assert_array_equal(pho.lightcurve['flux'], ref['flux'])
			np.testing.assert_array_equal(pho.lightcurve['flux_err'], ref['flux_err'])
			np.testing.assert_array_equal(pho.lightcurve['flux_background'], ref['flux_background'])
			assert pho._details.get('skip_targets', []) == ref['skip_targets']
			for key in ('mean_flux', 'variance', 'rms_hour', 'ptp', 'pos_centroid', 'variability', 'mask_size', 'edge_flux'):
				assert key in pho._details
			assert pho._details['mask_size'] == int(ref['mask'].sum())
			assert pho.additional_headers['KP_THRES'][0] == 0.8 and pho.additional_headers['KP_MIPIX'][0] == 4
			fname = pho.save_lightcurve()
		# saved file: same table, aperture bits 2 and 8 set (tests/test_aperturephotometry.py:70-96)
		from photometry_amd import fitsio
		assert fname.endswith('-tasoc_lc.fits.gz')
		hdus = fitsio.read(fname)
		assert [h.get('EXTNAME') for h, _ in hdus] == ['PRIMARY', 'LIGHTCURVE', 'SUMIMAGE', 'APERTURE']
		assert all(h['__checksum_ok__'] and h['__datasum_ok__'] for h, _ in hdus)
		prim, tab, aper = hdus[0][0], hdus[1][1], hdus[3][1]
		np.testing.assert_array_equal(tab['FLUX_RAW'], ref['flux'])
		np.testing.assert_array_equal(tab['MOM_CENTR1'], pho.lightcurve['pos_centroid'][:, 0])
		assert np.any(aper & 2 != 0) and np.any(aper & 8 != 0) and np.all(aper & 1 != 0)
		assert prim['KP_THRES'] == 0.8 and prim['TICID'] == int(s.target_starid[i]) and prim['FILEVER'] == '1.5'
		np.testing.assert_array_equal(hdus[2][1], pho.sumimage)


def test_tessphot_dispatch_and_details(ctx, tmp_path):
	s = _scene(n=3, seed=8)
	for method in ('aperture', None):
		pho = tessphot(method, int(s.target_starid[1]), source_from_scene(s, 1), str(tmp_path), ctx=ctx)
		assert pho.status in (STATUS.OK, STATUS.WARNING) and pho.method == 'aperture'
		assert 'filepath_lightcurve' in pho._details and os.path.exists(os.path.join(str(tmp_path), pho._details['filepath_lightcurve']))
		assert pho._details['stamp'] == tuple(int(v) for v in s.stamps[1])


def test_stamp_resize_retry(ctx, tmp_path):
	"""A bright star in a large frame: the mask touches the default stamp's edge -> resize_stamp(+10) and retry (photometry.py:123-165)."""
	from photometry_amd import simulate as sim
	R = C = 61
	T = 40
	s = sim.make_scene(1, T, R, C, seed=3, tmag_range=(6.5, 6.6), max_neighbours=0, sigma_psf=2.0)
	sim.fill_cubes(s, nan_fraction=0.0)
	# a bleed trail: +-15 rows of extra flux in the target's column -> the mask touches the top and bottom
	# edge of the 19x19 default stamp and fits after one resize by 10 pixels
	r0, c0 = int(round(s.star_params[0, 0, 0])), int(round(s.star_params[0, 0, 1]))
	s.images[0, r0-15:r0+16, c0:c0+2, :] += 4000.0 # 2 pixels wide: DBSCAN core pixels need 4 neighbours (k2p2v2.py:79)
	st = s.stamps[0]
	frames = {'images': s.images[0], 'images_err': s.images_err[0], 'backgrounds': s.backgrounds[0]}
	cat = s.catalog_of(0)
	src = MemoryStampSource(frames, st[0], st[2], s.time, s.timecorr, np.arange(T), s.quality,
		{'starid': cat['starid'], 'tmag': cat['tmag'], 'row': cat['row'], 'column': cat['column']},
		targets={'starid': s.target_starid, 'tmag': s.target_tmag, 'row': s.target_pos_row, 'column': s.target_pos_column})
	with AperturePhotometry(int(s.target_starid[0]), src, str(tmp_path), ctx=ctx) as pho:
		first = pho.stamp
		assert (first[1] - first[0], first[3] - first[2]) == (19, 19) # default stamp at Tmag 6.5: ceil(17.8) = 18 -> 2*9+1 (BasePhotometry.py:541-564, :646-651)
		pho.photometry()
		assert pho.status in (STATUS.OK, STATUS.WARNING)
		assert pho._details.get('stamp_resizes', 0) >= 1
		assert pho.stamp != first and pho.final_phot_mask.shape == (pho.stamp[1] - pho.stamp[0], pho.stamp[3] - pho.stamp[2])
		m = pho.final_phot_mask
		assert not (m[0].any() or m[-1].any() or m[:, 0].any() or m[:, -1].any()) or pho.stamp == tuple(src.max_stamp)


def test_linpsf_plugin(ctx, tmp_path):
	from photometry_amd import psf as hpsf
	from oracle import psf as opsf, linpsf as olin
	s = simulate.make_scene(3, 20, 11, 11, seed=61, max_neighbours=2, neighbour_tmag_range=(9.0, 15.0))
	simulate.fill_cubes(s, nan_fraction=0.005)
	prf = opsf.synthetic_prf(seed=2)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	for i in range(3):
		src = source_from_scene(s, i)
		src.prf = model
		pho = tessphot('linpsf', int(s.target_starid[i]), src, str(tmp_path), ctx=ctx)
		# flux_err is all NaN in the reference's LinPSF (linpsf_photometry.py:169), which photometry() rejects
		# (BasePhotometry.py:1348-1349) -> STATUS.ERROR exactly like upstream; the fluxes are still filled in:
		assert pho.status == STATUS.ERROR and any('errors are all NaNs' in e for e in pho._details['errors'])
		cat = pho.catalog
		T = s.n_cad
		positions = np.empty((T, len(cat), 2))
		for k in range(T):
			ck = pho.catalog_attime(pho.lightcurve['time'][k] - pho.lightcurve['timecorr'][k])
			positions[k, :, 0] = ck['row_stamp']
			positions[k, :, 1] = ck['column_stamp']
		p = opsf.PSF(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'], pho.stamp)
		ref = olin.do_photometry(s.images[i], p, {k: cat[k] for k in ('starid', 'tmag', 'row_stamp', 'column_stamp')},
			s.target_starid[i], positions, pho.stamp, pho.target_pos_row, pho.target_pos_column, np.ones((11, 11), dtype='int32'))
		np.testing.assert_allclose(pho.lightcurve['flux'], ref['flux'], rtol=1e-8, atol=1e-9*np.nanmax(np.abs(ref['flux'])))


def test_psf_plugin(ctx, tmp_path):
	"""tessphot('psf'): the non-linear PSF plugin end to end; like LinPSF its flux_err is all NaN upstream (psf_photometry.py:175),
	which BasePhotometry.photometry() rejects (BasePhotometry.py:1348-1349) -> STATUS.ERROR, with the light curve filled in."""
	from photometry_amd import psf as hpsf
	from oracle import psf as opsf, psf_photometry as opp
	s = simulate.make_scene(2, 3, 11, 11, seed=71, max_neighbours=1, neighbour_tmag_range=(9.0, 14.0))
	simulate.fill_cubes(s, nan_fraction=0.003)
	prf = opsf.synthetic_prf(seed=4)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	for i in range(2):
		src = source_from_scene(s, i)
		src.prf = model
		pho = tessphot('psf', int(s.target_starid[i]), src, str(tmp_path), ctx=ctx)
		assert pho.method == 'psf' and pho.status == STATUS.ERROR and any('errors are all NaNs' in e for e in pho._details['errors'])
		p = opsf.PSF(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'], pho.stamp)
		cat = pho.catalog
		ref = opp.do_photometry(s.images[i], s.backgrounds[i], p, {k: cat[k] for k in ('row_stamp', 'column_stamp', 'tmag')}, pho.stamp,
			pho.target_pos_row, pho.target_pos_column, pho.target['tmag'], pho.aperture, n_readout=pho.n_readout)
		ok = ref['success'] & ~np.isnan(pho.lightcurve['flux'])
		assert ok.sum() >= 2
		np.testing.assert_allclose(pho.lightcurve['flux'][ok], ref['flux'][ok], rtol=2e-5)
		np.testing.assert_allclose(pho.lightcurve['pos_centroid'][ok], ref['pos_centroid'][ok], atol=2e-4)


def test_batch_api(ctx):
	s = _scene(n=12, T=30, seed=9)
	res = tessphot_batch(ctx, s)
	assert len(res) == 12
	for i, r in enumerate(res):
		pho = None
		assert r.starid == s.target_starid[i] and r.method == 'aperture'
		assert r.status in (STATUS.OK, STATUS.WARNING, STATUS.ERROR)
		if r.status != STATUS.ERROR:
			assert r.lightcurve['flux'].shape == (30,) and r._details['mask_size'] == r.final_phot_mask.sum()
			# the batch carries the same diagnostics as the per-target plugin (host code mirroring BasePhotometry.py:1343-1407)
			with AperturePhotometry(int(s.target_starid[i]), source_from_scene(s, i), '/tmp', datasource='ffi', ctx=ctx) as pho:
				pho.photometry()
				assert pho.status == r.status
				for key in ('mean_flux', 'ptp', 'edge_flux'):
					assert r._details[key] == pho._details[key], key
				for key in ('variance', 'rms_hour', 'variability'):
					np.testing.assert_allclose(r._details[key], pho._details[key], rtol=1e-9, err_msg=key)
				np.testing.assert_array_equal(r._details['pos_centroid'], pho._details['pos_centroid'])


def test_pixel_flags_reach_the_saved_file(ctx, tmp_path):
	"""Background Shenanigans flags inside the stamp become QUALITY bit 256 of that timestamp (BasePhotometry.py:1445-1449);
	``backgrounds_pixels_used`` becomes bit 4 of the aperture image (:1052-1061)."""
	from photometry_amd import fitsio
	R = C = 41
	T = 30
	s = simulate.make_scene(1, T, R, C, seed=12, tmag_range=(10.0, 10.1), max_neighbours=0)
	simulate.fill_cubes(s, nan_fraction=0.0)
	st = s.stamps[0]
	flags = np.zeros((R, C, T), dtype='uint8')
	bpu = np.ones((R, C), dtype=bool)
	cat = s.catalog_of(0)
	frames = {'images': s.images[0], 'images_err': s.images_err[0], 'backgrounds': s.backgrounds[0], 'pixel_flags': flags}
	src = MemoryStampSource(frames, st[0], st[2], s.time, s.timecorr, np.arange(T), s.quality,
		{'starid': cat['starid'], 'tmag': cat['tmag'], 'row': cat['row'], 'column': cat['column']},
		targets={'starid': s.target_starid, 'tmag': s.target_tmag, 'row': s.target_pos_row, 'column': s.target_pos_column},
		backgrounds_pixels_used=bpu)
	with AperturePhotometry(int(s.target_starid[0]), src, str(tmp_path), ctx=ctx) as pho:
		r1, r2, c1, c2 = pho.stamp
		assert (r2 - r1) < R and (c2 - c1) < C, "the default stamp must be smaller than the frame for this test"
		# inside the stamp at cadences 3 and 17, outside it at cadence 9; other bits alone do not count
		flags[r1 - st[0] + 2, c1 - st[2] + 3, 3] = 4
		flags[r2 - st[0] - 1, c2 - st[2] - 1, 17] = 4 | 1
		flags[r1 - st[0] + 1, c1 - st[2] + 1, 20] = 2 | 1
		outside = (r1 - st[0] - 1) if r1 > st[0] else (r2 - st[0])
		flags[outside, c1 - st[2], 9] = 4
		bpu[r1 - st[0]:r1 - st[0] + 3, c1 - st[2]:c2 - st[2]] = False
		src.frames['pixel_flags'] = flags
		pho.photometry()
		assert pho.status in (STATUS.OK, STATUS.WARNING)
		assert pho.stamp == (r1, r2, c1, c2)
		fname = pho.save_lightcurve()
		ap = pho.aperture
	hdus = fitsio.read(fname)
	q = hdus[1][1]['QUALITY']
	expect = np.zeros(T, dtype='int32'); expect[[3, 17]] = 256
	np.testing.assert_array_equal(q, expect)
	aper = hdus[3][1]
	assert np.all(aper[:3] & 4 == 0) and np.all(aper[3:] & 4 != 0)
	np.testing.assert_array_equal(ap & 4, aper & 4)
