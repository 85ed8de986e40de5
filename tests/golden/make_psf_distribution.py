#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""
Fixture behind the PSFPhotometry parity-by-distribution test (tests/test_gpu_psfphot.py::test_psf_parity_by_distribution):
the ORACLE's Nelder-Mead fit (oracle/psf_photometry.py: scipy's routine restated step for step on the FITPACK pixel integral,
psf_photometry.py:52-108, 143-196) of NT targets x T cadences of a seeded scene -- flux, centroid and iteration count per cadence.
The scene itself is not stored: the test rebuilds it from the seeds recorded here (photometry_amd.simulate is deterministic).

    python tests/golden/make_psf_distribution.py          # ~12 min on 7 processes; writes tests/golden/golden_psf_distribution.npz

CPU only; the oracle runs in a forked process pool.
"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from multiprocessing import get_context

NT, T, H, W = int(os.environ.get('NT', 120)), int(os.environ.get('T', 20)), 11, 11
SCENE_SEED, PRF_SEED, NAN_FRACTION = 191, 5, 0.004
SCENE = PRF = CATS = None


def build_scene():
	"""The scene of the fixture (also called by the test)."""
	from photometry_amd import simulate
	from oracle import psf as opsf
	s = simulate.make_scene(NT, T, H, W, seed=SCENE_SEED, max_neighbours=3, neighbour_tmag_range=(9.0, 15.0))
	simulate.fill_cubes(s, nan_fraction=NAN_FRACTION)
	return s, opsf.synthetic_prf(seed=PRF_SEED)


def oracle_job(i):
	from oracle import psf as opsf, psf_photometry as opp
	s = SCENE
	p = opsf.PSF(PRF['values'], PRF['ccdColumn'], PRF['ccdRow'], PRF['prfColumn'], PRF['prfRow'], tuple(s.stamps[i]))
	ref = opp.do_photometry(s.images[i], s.backgrounds[i], p, CATS[i], tuple(s.stamps[i]), s.target_pos_row[i], s.target_pos_column[i],
		s.target_tmag[i], s.aperture[i], use_scipy=False)
	return i, np.asarray(ref['flux']), np.asarray(ref['pos_centroid']), np.asarray(ref['nit']), int(ref['status'])


if __name__ == '__main__':
	SCENE, PRF = build_scene()
	CATS = [SCENE.catalog_of(i) for i in range(NT)]
	t0 = time.time()
	nproc = max(1, min(15, len(os.sched_getaffinity(0)) - 1))
	flux = np.full((NT, T), np.nan)
	cen = np.full((NT, T, 2), np.nan)
	nit = np.zeros((NT, T), dtype='int32')
	status = np.zeros(NT, dtype='int32')
	done = 0
	with get_context('fork').Pool(nproc) as pool:
		for i, f, c, n, st in pool.imap_unordered(oracle_job, range(NT), chunksize=1):
			flux[i], cen[i], nit[i], status[i] = f, c, n, st
			done += 1
			if done % 10 == 0:
				print(f'oracle: {done} of {NT} targets after {time.time() - t0:.0f} s', flush=True)
	out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden_psf_distribution.npz')
	np.savez_compressed(out, flux=flux, pos_centroid=cen, nit=nit, status=status, shape=np.array([NT, T, H, W]),
		seeds=np.array([SCENE_SEED, PRF_SEED]), nan_fraction=np.array([NAN_FRACTION]),
		images_checksum=np.array([float(np.nansum(SCENE.images.astype('float64')))]))
	print(f'wrote {out}: {NT} targets x {T} cadences in {time.time() - t0:.0f} s on {nproc} processes')
