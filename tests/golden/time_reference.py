#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""
Calibration of the CPU baseline: the REFERENCE'S OWN loops, executed here through ``_refstub``, timed
beside the oracle's restatement of them on the same inputs and the same core.  Runs only in the dev container (needs
/root/reference); prints the per-target times and the oracle / reference ratio that DESIGN.md quotes.

* A5b/A6/A7: ``AperturePhotometry.do_photometry`` (photometry.py:44-257) with ``k2p2FixFromSum`` patched to return a prescribed
  mask, 15 x 15 x 1300 targets -- against ``oracle.aperture.do_photometry(masks=...)``;
* P4: ``LinPSFPhotometry.do_photometry`` (linpsf_photometry.py:79-219), 15 x 15 stamps, 3 fitted stars, a few cadences (every
  cadence is 225 x nstars FITPACK integrals upstream) -- against ``oracle.linpsf.do_photometry``.
"""
import os
import sys
import time
import warnings
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg # noqa: E402  (imports the reference through the stub finder)
from oracle import sumimage as osum, aperture as oap, k2p2 as ok2p2, psf as opsf, linpsf as olin # noqa: E402
from photometry_amd import simulate # noqa: E402


def aperture_loop(n=6, T=1300):
	scene = simulate.make_scene(n, T, 15, 15, seed=1)
	simulate.fill_cubes(scene)
	S = osum.sumimage_batch(scene.images, scene.quality)
	t_ref = t_ora = 0.0
	done = 0
	for i in range(n):
		cat = scene.catalog_of(i)
		c = np.column_stack((cat['column_stamp'], cat['row_stamp'], cat['tmag']))
		try:
			mm, _ = ok2p2.k2p2FixFromSum(S[i], catalog=c, **oap.K2P2_SETTINGS)
		except Exception: # noqa: B902
			continue
		if mm is None:
			continue
		f = mg.make_fake(mg.AperturePhotometry, scene, i, S[i])
		mg.ap_module.k2p2.k2p2FixFromSum = lambda SumImage, _mm=mm, **kw: (np.array(_mm, dtype='float64'), 1.0)
		with warnings.catch_warnings():
			warnings.simplefilter('ignore')
			t0 = time.perf_counter()
			status = mg.AperturePhotometry.do_photometry(f)
			t_ref += time.perf_counter() - t0
			t0 = time.perf_counter()
			r = oap.do_photometry(S[i], scene.images[i], scene.images_err[i], scene.backgrounds[i], tuple(scene.stamps[i]),
				scene.target_pos_row[i], scene.target_pos_column[i], scene.target_tmag[i], scene.target_starid[i], cat, scene.aperture[i],
				masks=np.asarray(mm, dtype=bool))
			t_ora += time.perf_counter() - t0
		assert status.value == r['status']
		np.testing.assert_array_equal(np.asarray(f.lightcurve['flux']), r['flux'])
		done += 1
	mg.ap_module.k2p2.k2p2FixFromSum = mg.k2p2v2.k2p2FixFromSum
	print(f'aperture loop (mask given), {done} targets x {T} cadences x 15x15: reference {t_ref / done:.4f} s/target = {done / t_ref:.1f} targets/s/core;'
		f' oracle {t_ora / done:.4f} s/target = {done / t_ora:.1f} targets/s/core; oracle / reference time = {t_ora / t_ref:.2f}')


def linpsf_loop(n=3, T=8):
	from photometry.linpsf_photometry import LinPSFPhotometry
	from photometry.psf import PSF
	x, img, spline = mg._synthetic_spline()
	H = W = 15
	scene = simulate.make_scene(n, T, H, W, seed=41, max_neighbours=2, neighbour_tmag_range=(9.0, 13.0))
	simulate.fill_cubes(scene, nan_fraction=0.0)
	t_ref = t_ora = 0.0
	stars = 0
	for i in range(n):
		f = mg.make_fake(LinPSFPhotometry, scene, i, None)
		f.cutoff_radius = 5
		p = PSF.__new__(PSF)
		p.shape = (H, W)
		p.stamp = f._stamp
		p.splineInterpolation = spline
		f._psf = p
		cat0 = scene.catalog_of(i)
		nall = len(cat0['starid'])
		positions = np.empty((T, nall, 2))
		for k in range(T):
			positions[k, :, 0] = cat0['row_stamp'] + scene.jitter[k, 1]
			positions[k, :, 1] = cat0['column_stamp'] + scene.jitter[k, 0]
		times = np.asarray(f.lightcurve['time']) - np.asarray(f.lightcurve['timecorr'])

		def catalog_attime(t, _cat0=cat0, _pos=positions, _times=times):
			k = int(np.argmin(np.abs(_times - t)))
			c = {kk: vv.copy() for kk, vv in _cat0.items()}
			c['row_stamp'] = _pos[k, :, 0].copy()
			c['column_stamp'] = _pos[k, :, 1].copy()
			return mg._refstub.FakeCatalog(**c)
		f.catalog_attime = catalog_attime
		op = opsf.PSF.__new__(opsf.PSF)
		op.shape, op.stamp = (H, W), f._stamp
		op.tx, op.ty, op.coeffs = spline.get_knots()[0], spline.get_knots()[1], spline.get_coeffs().reshape(len(x), len(x))
		with warnings.catch_warnings():
			warnings.simplefilter('ignore')
			t0 = time.perf_counter()
			mg_status = LinPSFPhotometry.do_photometry(f)
			t_ref += time.perf_counter() - t0
			t0 = time.perf_counter()
			r = olin.do_photometry(scene.images[i], op, cat0, scene.target_starid[i], positions, f._stamp, scene.target_pos_row[i],
				scene.target_pos_column[i], scene.aperture[i])
			t_ora += time.perf_counter() - t0
		np.testing.assert_allclose(np.asarray(f.lightcurve['flux']), r['flux'], rtol=1e-8)
		stars += r['nstars']
	print(f'LinPSF loop, {n} targets x {T} cadences x 15x15, {stars / n:.1f} fitted stars: reference {t_ref / (n * T) * 1e3:.2f} ms/cadence;'
		f' oracle {t_ora / (n * T) * 1e3:.2f} ms/cadence; oracle / reference time = {t_ora / t_ref:.2f}')


if __name__ == '__main__':
	aperture_loop()
	linpsf_loop()
