# -*- coding: utf-8 -*-
"""
Edge cases of the batched aperture path through the C ABI: empty batches for every entry, ragged catalogues with
empty segments, a single cadence, light curves where every cadence is flagged, the largest LDS-resident stamp.
The oracle (test infrastructure) is the checker, exactly as in the other parity tests.
"""
import ctypes
import numpy as np
import pytest
from photometry_amd import simulate, pipeline, engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
	from photometry_amd.device import Context
	c = Context(0)
	yield c
	c.close()


def _oracle(s, S, i):
	from oracle import aperture as oap
	ap = np.ones((s.height, s.width), dtype='int32')
	return oap.do_photometry(S[i], s.images[i], s.images_err[i], s.backgrounds[i], tuple(s.stamps[i]),
		s.target_pos_row[i], s.target_pos_column[i], s.target_tmag[i], s.target_starid[i], s.catalog_of(i), ap)


def _compare(s, got):
	n_ok = 0
	for i in range(s.n_targets):
		ref = _oracle(s, got['sumimage'], i)
		assert int(got['status'][i]) == ref['status'], (i, got['status'][i], ref['status'], got['flags'][i])
		if ref.get('mask') is not None and ref['status'] != 2:
			np.testing.assert_array_equal(got['mask'][i].astype(bool), ref['mask'], err_msg=str(i))
			np.testing.assert_array_equal(got['flux'][i], ref['flux'], err_msg=str(i))
			np.testing.assert_array_equal(got['flux_background'][i], ref['flux_background'], err_msg=str(i))
			n_ok += 1
	return n_ok


def test_empty_batches_are_ok(ctx):
	"""n_targets == 0 is a valid call for every entry point (the scheduler may hand over an empty chunk)."""
	from photometry_amd._lib import tp_cube_desc
	lib, h = ctx.lib, ctx.handle
	desc = tp_cube_desc(0, 10, 5, 5, 32)
	p = ctx.zeros((64,), 'float64').ptr # any valid device pointer
	assert lib.tp_sumimage(h, ctypes.byref(desc), p, p, 0, 4335, None, 0, p) == 0
	assert lib.tp_aperture_extract(h, ctypes.byref(desc), p, p, p, 0, 0, None, 0, p, p, None, p, p, p, p, p, 10) == 0
	assert lib.tp_aperture_photometry(h, ctypes.byref(desc), p, p, p, 0, 0, None, 0, p, 0, 4335, p, p, p, p, p, p, p, p, p, p, p, p, p, None,
		p, p, p, p, p, p, p, p, p, p, p, p, 10) == 0
	assert lib.tp_background_stamp(h, ctypes.byref(desc), p, 8e4, 50.0, p, 32) == 0
	assert lib.tp_lightcurve_diagnostics(h, 0, 10, p, p, p, p, 10, p, p, 0, 4335, None, None, None, 0, 0, 1/24, p) == 0
	assert lib.tp_cut_stamps(h, p, 10, 4, 4, 4, 16, 0, 0, p, ctypes.byref(desc), p) == 0
	ctx.sync()


def test_ragged_catalogue_with_empty_segments(ctx):
	"""Targets whose catalogue query returned nothing (CSR segment of length 0) next to ordinary ones."""
	s = simulate.make_scene(9, 40, 11, 11, seed=21)
	simulate.fill_cubes(s)
	s.aperture = None
	# drop the catalogue rows of targets 2 and 5
	keep = np.ones(len(s.catalog['starid']), dtype=bool)
	for i in (2, 5):
		keep[s.cat_offsets[i]:s.cat_offsets[i+1]] = False
	counts = np.diff(s.cat_offsets)
	counts[[2, 5]] = 0
	s.catalog = {k: v[keep] for k, v in s.catalog.items()}
	s.cat_offsets = np.concatenate(([0], np.cumsum(counts))).astype('int64')
	got = pipeline.run_aperture(ctx, s)
	assert _compare(s, got) >= 5
	# no catalogue star can fall in the mask: "No targets in mask" (photometry.py:227-230)
	assert int(got['status'][2]) in (2, 3) and int(got['status'][5]) in (2, 3)


def test_single_cadence_and_all_flagged(ctx):
	s = simulate.make_scene(5, 1, 11, 11, seed=3)
	simulate.fill_cubes(s)
	s.aperture = None
	got = pipeline.run_aperture(ctx, s)
	assert _compare(s, got) >= 3
	# every cadence fails the quality bitmask: the sum image is all NaN -> K2P2NoFlux -> ERROR, nothing extracted
	s2 = simulate.make_scene(4, 33, 11, 11, seed=4)
	simulate.fill_cubes(s2)
	s2.aperture = None
	s2.quality[:] = 32
	got2 = pipeline.run_aperture(ctx, s2)
	assert np.all(np.isnan(got2['sumimage']))
	assert np.all(got2['status'] == 2)
	assert not got2['mask'].any()
	d = got2['diagnostics']
	assert np.all(np.isnan(d)) # ERROR targets get no diagnostics


def test_largest_lds_resident_stamp(ctx):
	"""52 x 52 pixels run on the LDS-resident mask builder (the limit is about 54 x 54); a 60 x 60 stamp takes the same code
	with its work arrays in HBM (tp_aperture_photometry then runs its three stages in turn): both against the oracle."""
	s = simulate.make_scene(2, 20, 52, 52, seed=6, tmag_range=(6.5, 8.0))
	simulate.fill_cubes(s)
	s.aperture = None
	got = pipeline.run_aperture(ctx, s)
	assert _compare(s, got) >= 1
	s3 = simulate.make_scene(2, 8, 60, 60, seed=6, tmag_range=(6.0, 8.0))
	simulate.fill_cubes(s3)
	s3.aperture = None
	got3 = pipeline.run_aperture(ctx, s3)
	assert _compare(s3, got3) >= 1


def test_two_minute_cadence_length(ctx):
	"""A sector of 2-minute data (19 500 cadences, the TPF data source): every stage of the batch path, against the oracle."""
	s = simulate.make_scene(3, 19500, 11, 11, seed=12, cadence_s=120.0)
	simulate.fill_cubes(s)
	s.aperture = None
	got = pipeline.run_aperture(ctx, s)
	assert _compare(s, got) >= 2
	d = got['diagnostics']
	from oracle import diagnostics as odiag
	for i in range(3):
		if int(got['status'][i]) in (1, 3):
			o = odiag.diagnostics(s.time, s.quality, got['flux'][i], got['flux_err'][i], got['pos_centroid'][i], sumimage=got['sumimage'][i], mask=got['mask'][i])
			assert d[i][0] == o['mean_flux'] and d[i][3] == o['ptp']
			np.testing.assert_allclose(d[i][[1, 2]], [o['variance'], o['rms_hour']], rtol=1e-12)
			np.testing.assert_allclose(d[i][6], o['variability'], rtol=1e-9)


def test_allocation_takes_back_the_blocks_other_contexts_have_cached():
	"""Blocks a context has freed stay in ITS cache; an allocation on another context of the device that would fail for them
	gets them back to the driver and succeeds (the batched frames entry runs several contexts side by side)."""
	from photometry_amd.device import Context
	from photometry_amd._lib import TessphotError
	a, b = Context(0), Context(0)
	total = a.info()['hbm_bytes']
	if total < 200e9:
		pytest.skip("sized for the 288 GB device")
	gb = 1 << 30
	# what one block can get right now (other tests of the process may hold memory): the largest multiple of 8 GiB that fits;
	# blocks above 32 GiB bypass the cache, so the probe leaves nothing behind
	fit = 0
	for size in range(int(total) // gb // 8 * 8, 0, -8):
		try:
			probe = b.empty((size * gb,), 'uint8')
		except TessphotError:
			continue
		probe.free()
		fit = size
		break
	if fit < 120:
		pytest.skip("less than 120 GiB free in this process")
	blocks = [a.empty((28 * gb,), 'uint8') for _ in range(2)]
	for blk in blocks:
		blk.free()                                     # 56 GiB idle in a's cache
	big = b.empty(((fit - 8) * gb,), 'uint8')          # does not fit beside them
	big.free()
	again = a.empty((28 * gb,), 'uint8')              # a's cache is empty now: from the driver
	again.free()
	a.close()
	b.close()
