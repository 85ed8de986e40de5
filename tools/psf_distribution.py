#!/usr/bin/env python3
"""
Measurement behind the PSFPhotometry parity statement: the distribution of |dflux| / flux and of the centroid difference between the
device's Nelder-Mead fit (tp_psf_fit) and the oracle's (scipy's routine restated step for step, FITPACK pixel integral) on NT targets
x T cadences, and how often the finite / NaN pattern (`success` = finished before maxiter, psf_photometry.py:190-194) differs.
The oracle runs in a process pool, forked before this process opens the GPU.
    NT=200 T=20 python tools/psf_distribution.py > profiles/r4_psf_flux_distribution.txt
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from multiprocessing import get_context

Nt, T, H, W = int(os.environ.get('NT', 200)), int(os.environ.get('T', 20)), 11, 11
SCENE = PRF = CATS = None


def setup():
	global SCENE, PRF, CATS
	from photometry_amd import simulate
	from oracle import psf as opsf
	SCENE = simulate.make_scene(Nt, T, H, W, seed=191, max_neighbours=3, neighbour_tmag_range=(9.0, 15.0))
	simulate.fill_cubes(SCENE, nan_fraction=0.004)
	PRF = opsf.synthetic_prf(seed=5)
	CATS = [SCENE.catalog_of(i) for i in range(Nt)]


def oracle_job(i):
	from oracle import psf as opsf, psf_photometry as opp
	s = SCENE
	p = opsf.PSF(PRF['values'], PRF['ccdColumn'], PRF['ccdRow'], PRF['prfColumn'], PRF['prfRow'], tuple(s.stamps[i]))
	ref = opp.do_photometry(s.images[i], s.backgrounds[i], p, CATS[i], tuple(s.stamps[i]), s.target_pos_row[i], s.target_pos_column[i],
		s.target_tmag[i], s.aperture[i], use_scipy=False)
	return i, np.asarray(ref['flux']), np.asarray(ref['pos_centroid']), np.asarray(ref['nit'])


if __name__ == '__main__':
	setup()
	t0 = time.time()
	nproc = max(1, min(15, len(os.sched_getaffinity(0)) - 1))
	refs = {}
	with get_context('fork').Pool(nproc) as pool:
		for r in pool.imap_unordered(oracle_job, range(Nt), chunksize=1):
			refs[r[0]] = r[1:]
			if len(refs) % 10 == 0:   # (a run that prints nothing for minutes is taken to be hung)
				print(f'# oracle: {len(refs)} of {Nt} targets after {time.time() - t0:.0f} s', file=sys.stderr, flush=True)
	print(f'# oracle: {Nt} targets x {T} cadences in {time.time() - t0:.0f} s on {nproc} processes', flush=True)
	import test_gpu_psfphot as tg
	from photometry_amd import psf as hpsf
	from photometry_amd.device import Context
	s = SCENE
	model = hpsf.PRFModel(PRF['values'], PRF['ccdColumn'], PRF['ccdRow'], PRF['prfColumn'], PRF['prfRow'])
	ctx = Context(0)
	res = tg._run_device(ctx, s.images, s.backgrounds, model, s.stamps, CATS, s.target_pos_row, s.target_pos_column, s.target_tmag, s.aperture)
	rel, dpos, same_nit, n_both, n_pat = [], [], 0, 0, 0
	pat = []
	for i in range(Nt):
		flux, cen, nit = refs[i]
		d = res['flux'][i][:T]
		fin_d, fin_o = np.isfinite(d), np.isfinite(flux)
		for k in np.flatnonzero(fin_d != fin_o):
			pat.append((i, int(k), int(res['nit'][i][k]), int(nit[k])))
		both = fin_d & fin_o
		n_both += int(both.sum())
		rel += list(np.abs(d[both] / flux[both] - 1))
		dc = np.abs(np.stack((res['centroid_row'][i][:T], res['centroid_col'][i][:T]), axis=-1) - cen)[both]
		dpos += list(dc.max(axis=1))
		same_nit += int(np.sum(res['nit'][i][:T][both] == nit[both]))
	rel, dpos = np.asarray(rel), np.asarray(dpos)
	print(f'# {Nt} targets x {T} cadences: {n_both} cadences finite on both sides, {len(pat)} where one side is NaN and the other is not')
	for q in (50, 90, 99, 99.9, 100):
		print(f'|dflux|/flux  percentile {q:5.1f}: {np.percentile(rel, q):.3e}      |dpos| px: {np.percentile(dpos, q):.3e}')
	print(f'identical iteration counts: {same_nit} of {n_both} ({100.0 * same_nit / max(n_both, 1):.1f} %)')
	for i, k, nd, no in pat:
		print(f'finite / NaN pattern differs: target {i} cadence {k}: iterations device {nd}, oracle {no}')
	ctx.close()
