#!/usr/bin/env python3
"""Diagnostic: where the device's non-linear PSF fit and the oracle's differ -- per cadence the relative flux difference, the position
difference, the iteration counts and the ORACLE's chi^2 at both solutions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from photometry_amd import simulate, psf as hpsf
from photometry_amd.device import Context
from oracle import psf as opsf, psf_photometry as opp
import test_gpu_psfphot as tg

Nt, T, H, W = int(os.environ.get('NT', 4)), int(os.environ.get('T', 5)), 11, 11
s = simulate.make_scene(Nt, T, H, W, seed=91, max_neighbours=3, neighbour_tmag_range=(9.0, 15.0))
simulate.fill_cubes(s, nan_fraction=0.004)
prf = opsf.synthetic_prf(seed=5)
model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
cats = [s.catalog_of(i) for i in range(Nt)]
ctx = Context(0)
res = tg._run_device(ctx, s.images, s.backgrounds, model, s.stamps, cats, s.target_pos_row, s.target_pos_column, s.target_tmag, s.aperture)
offs = np.cumsum([0] + [len(opp.select_stars(cats[i], s.target_pos_row[i] - s.stamps[i][0], s.target_pos_column[i] - s.stamps[i][2], s.target_tmag[i])) for i in range(Nt)])
for i in range(Nt):
	p = opsf.PSF(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'], tuple(s.stamps[i]))
	ref = opp.do_photometry(s.images[i], s.backgrounds[i], p, cats[i], tuple(s.stamps[i]), s.target_pos_row[i], s.target_pos_column[i],
		s.target_tmag[i], s.aperture[i], use_scipy=False)
	ns = offs[i+1] - offs[i]
	for k in range(T):
		pd = res['params'][3*offs[i]:3*offs[i+1], k].reshape(ns, 3) if res['params'].shape[0] >= 3*offs[-1] else None
		po = ref['params'][k]
		line = f"target {i} cad {k}: nit dev {res['nit'][i][k]} ora {ref['nit'][k]}  flux dev {res['flux'][i][k]:.6f} ora {ref['flux'][k]:.6f} rel {abs(res['flux'][i][k]/ref['flux'][k]-1):.2e}"
		if pd is not None and np.all(np.isfinite(po)):
			c_dev = opp.lhood(pd.flatten(), p, s.images[i][:, :, k], s.backgrounds[i][:, :, k])
			c_ora = opp.lhood(po.flatten(), p, s.images[i][:, :, k], s.backgrounds[i][:, :, k])
			line += f"  dpos {np.abs(pd[0,:2]-po[0,:2]).max():.2e} dfitflux {abs(pd[0,2]/po[0,2]-1):.2e}  chi2(oracle) at dev {c_dev:.9f} at ora {c_ora:.9f} diff {c_dev-c_ora:.3e}"
		print(line, flush=True)
