# -*- coding: utf-8 -*-
"""
GPU parity of the light-curve diagnostics (BasePhotometry.py:1343-1407) through the C ABI:
against the golden vectors produced by the reference's own ``photometry()`` and against the oracle on seeded
light curves with NaNs, flagged cadences, gaps and degenerate cases.

Tolerances: medians (mean_flux, ptp, centroid), mask_size and edge_flux are selections / integer / the same summation
order -> exact; variance, rms_hour use tree sums instead of numpy's pairwise sums -> 1e-12 relative; variability goes
through a differently conditioned least-squares solve -> 1e-9 relative.
"""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EXACT = ('mean_flux', 'ptp', 'pos_centroid_col', 'pos_centroid_row', 'mask_size', 'edge_flux')


@pytest.fixture(scope='module')
def ctx():
	from photometry_amd.device import Context
	c = Context(0)
	yield c
	c.close()


def _run(ctx, time, quality, flux, flux_err, cen, status=None, sumimage=None, mask=None):
	from photometry_amd import engine
	Nt, T = flux.shape
	lc = engine.LightCurves(ctx, Nt, T)
	block = np.zeros((5, Nt, T))
	block[0], block[1], block[3], block[4] = flux, flux_err, cen[..., 0], cen[..., 1]
	ctx._check(ctx.lib.tp_memcpy_h2d(ctx.handle, lc.block.ptr, np.ascontiguousarray(block).ctypes.data, block.nbytes))
	out = engine.lightcurve_diagnostics(ctx, lc, ctx.array(np.asarray(time, dtype='float64')), ctx.array(np.asarray(quality, dtype='int32')),
		status=None if status is None else ctx.array(np.asarray(status, dtype='int32')),
		sumimage=None if sumimage is None else ctx.array(np.asarray(sumimage, dtype='float64')),
		mask=None if mask is None else ctx.array(np.asarray(mask, dtype='uint8')))
	ctx.sync()
	return out.to_host()


def _check(got, ref, tag=''):
	from photometry_amd.engine import DIAGNOSTICS_COLUMNS as COLS
	for j, key in enumerate(COLS):
		g, r = got[j], ref[key]
		if key == 'flags':
			assert int(g) == int(r), (tag, key, g, r)
		elif key in EXACT:
			assert (g == r) or (np.isnan(g) and np.isnan(r)), (tag, key, g, r)
		else:
			np.testing.assert_allclose(g, r, rtol=1e-9 if key == 'variability' else 1e-12, equal_nan=True, err_msg=f'{tag} {key}')


def test_golden_reference_photometry(ctx, golden_dir):
	g = np.load(os.path.join(golden_dir, 'golden_diagnostics.npz'))
	n = int(g['n_cases'])
	flux = np.array([g[f'case{i}_flux'] for i in range(n)])
	ferr = np.array([g[f'case{i}_flux_err'] for i in range(n)])
	cen = np.array([g[f'case{i}_pos_centroid'] for i in range(n)])
	mask = np.array([g[f'case{i}_mask'] for i in range(n)])
	got = _run(ctx, g['time'], g['quality'], flux, ferr, cen, status=[int(g[f'case{i}_status']) for i in range(n)],
		sumimage=g['sumimage'], mask=mask)
	for i in range(n):
		ref = {k: float(g[f'case{i}_{k}']) for k in ('mean_flux', 'variance', 'rms_hour', 'ptp', 'variability', 'mask_size', 'edge_flux')}
		ref['pos_centroid_col'], ref['pos_centroid_row'] = g[f'case{i}_det_pos_centroid']
		ref['flags'] = 0
		_check(got[i], ref, tag=f'case{i}')


# 6000 and 19500 cadences (2-minute data of a sector): the series arrays no longer fit the LDS and move to HBM scratch
@pytest.mark.parametrize('T', [7, 300, 1300, 2500, 6000, 19500])
def test_against_oracle(ctx, T):
	from oracle import diagnostics as odiag
	rng = np.random.default_rng(T)
	Nt = 12
	time = 1400.0 + np.arange(T) * (1800.0 / 86400.0) + rng.normal(0, 1e-5, T)
	time = np.sort(time)
	if T > 100:
		time[T // 2:] += 1.3 # a data gap
	quality = np.zeros(T, dtype='int32')
	quality[rng.random(T) < 0.03] = 32
	quality[rng.random(T) < 0.02] = 16 # not in the default bitmask
	mean = 10**rng.uniform(2, 5, Nt)
	flux = mean[:, None] * (1 + 1e-3 * rng.standard_normal((Nt, T)) + 2e-3 * np.sin(np.arange(T) / 50.0)[None, :])
	ferr = np.sqrt(np.abs(flux)) * (1 + 0.01 * rng.standard_normal((Nt, T)))
	cen = np.stack((100.3 + 0.01 * rng.standard_normal((Nt, T)), 200.7 + 0.01 * rng.standard_normal((Nt, T))), axis=-1)
	flux[rng.random((Nt, T)) < 0.01] = np.nan
	ferr[np.isnan(flux)] = np.nan
	cen[rng.random((Nt, T)) < 0.01] = np.nan
	status = np.ones(Nt, dtype='int32')
	status[3] = 3
	if T >= 300:
		flux[4] = np.nan; ferr[4] = np.nan          # all-NaN light curve -> flag 1
		ferr[5] = np.nan                            # all-NaN errors -> flag 2
		status[6] = 2                               # ERROR target: no diagnostics at all
		flux[7, :] = flux[7, 0]                     # constant light curve
	H, W = 9, 13
	S = rng.uniform(-5, 500, (Nt, H, W))
	S[rng.random((Nt, H, W)) < 0.05] = np.nan
	mask = rng.random((Nt, H, W)) < 0.4
	got = _run(ctx, time, quality, flux, ferr, cen, status=status, sumimage=S, mask=mask)
	for i in range(Nt):
		if status[i] == 2:
			assert np.all(np.isnan(got[i]))
			continue
		ref = odiag.diagnostics(time, quality, flux[i], ferr[i], cen[i], sumimage=S[i], mask=mask[i])
		_check(got[i], ref, tag=f'T{T} target{i}')


def test_reference_rms_timescale_known_answers(ctx):
	"""The known answers of the reference's own tests/test_utilities.py:75-116 (rms_timescale), through the device kernel:
	zero flux -> 0, all-NaN flux -> NaN, a timescale longer than the time span -> 0, and the three invalid time vectors
	(+inf, -inf, all timestamps equal) that raise ValueError upstream -> the BAD_TIME flag (4)."""
	from photometry_amd import engine
	from photometry_amd.engine import DIAGNOSTICS_COLUMNS as COLS
	jr, jf = COLS.index('rms_hour'), COLS.index('flags')

	def rms(time, flux, timescale=3600 / 86400):
		T = len(time)
		lc = engine.LightCurves(ctx, 1, T)
		block = np.zeros((5, 1, T))
		block[0, 0] = flux
		block[1, 0] = 1.0
		ctx._check(ctx.lib.tp_memcpy_h2d(ctx.handle, lc.block.ptr, np.ascontiguousarray(block).ctypes.data, block.nbytes))
		out = engine.lightcurve_diagnostics(ctx, lc, ctx.array(np.asarray(time, dtype='float64')), ctx.array(np.zeros(T, dtype='int32')),
			timescale=timescale)
		ctx.sync()
		row = out.to_host()[0]
		return row[jr], int(row[jf]) if np.isfinite(row[jf]) else 0

	# the kernel works on rel = flux / median - 1: a constant light curve is the reference's all-zero series
	time = np.linspace(0, 27, 100)
	r, f = rms(time, np.full(100, 7.5))
	assert r == 0 and (f & 4) == 0
	r, f = rms(time, np.full(100, np.nan))
	assert np.isnan(r)
	rng = np.random.default_rng(0)
	flux = 1000 + rng.standard_normal(1000)
	time = np.linspace(0, 27, 1000)
	r, f = rms(time, flux, timescale=30.0)
	assert r == 0 and (f & 4) == 0
	for bad in (np.inf, -np.inf):
		t2 = time.copy()
		t2[1] = bad
		r, f = rms(t2, flux)
		assert (f & 4) == 4 and np.isnan(r), bad
	r, f = rms(np.full(1000, 1.2), flux)
	assert (f & 4) == 4 and np.isnan(r)
	r, f = rms(time * np.nan, flux)
	assert (f & 4) == 4 and np.isnan(r)
	# a sparse series (more one-hour bins than samples) with real scatter: against the oracle
	from oracle.utilities import rms_timescale
	time = np.sort(rng.uniform(0, 27, 100))
	flux = 1000 + rng.standard_normal(100)
	flux[7] = np.nan
	r, f = rms(time, flux)
	keep = np.isfinite(flux)
	rel = flux / np.nanmedian(flux) - 1
	np.testing.assert_allclose(r, rms_timescale(time, rel), rtol=1e-12)
	assert f == 0


def test_edge_flux_of_large_stamps(ctx):
	"""Stamps of bright stars (BasePhotometry.py:541-564: 163 x 69 at Tmag 2): more in-mask edge pixels than one pairwise leaf
	or two -- numpy's recursive pairwise sum (edge_flux, :1400-1403) at any length, exact."""
	from oracle import diagnostics as odiag
	rng = np.random.default_rng(5)
	T = 30
	time = 1400.0 + np.arange(T) * (1800.0 / 86400.0)
	quality = np.zeros(T, dtype='int32')
	for (H, W) in ((90, 70), (163, 69), (40, 30)):
		Nt = 3
		flux = 1e4 * (1 + 1e-3 * rng.standard_normal((Nt, T)))
		ferr = np.sqrt(flux)
		cen = np.stack((100 + 0.01 * rng.standard_normal((Nt, T)), 200 + 0.01 * rng.standard_normal((Nt, T))), axis=-1)
		S = rng.uniform(1, 500, (Nt, H, W))
		S[rng.random((Nt, H, W)) < 0.02] = np.nan
		mask = np.ones((Nt, H, W), dtype=bool)
		mask[1] = rng.random((H, W)) < 0.7
		mask[2, 1:-1, :] = False # the two edge rows only
		got = _run(ctx, time, quality, flux, ferr, cen, status=np.ones(Nt, dtype='int32'), sumimage=S, mask=mask)
		for i in range(Nt):
			ref = odiag.diagnostics(time, quality, flux[i], ferr[i], cen[i], sumimage=S[i], mask=mask[i])
			_check(got[i], ref, tag=f'{H}x{W} target{i}')
