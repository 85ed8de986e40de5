# -*- coding: utf-8 -*-
"""
ORACLE (test infrastructure only) -- non-linear PSF photometry.

Restates ``photometry/psf_photometry.py``: ``_minimum_aperture`` (:29-41), ``_lhood`` with the statistic the plugin uses
(``Gaussian_d``, background included; :52-108) and ``PSFPhotometry.do_photometry`` (:111-196): star selection, the per-cadence
Nelder-Mead fit of (row, column, flux) of up to five stars, warm-started from the previous cadence, the MOMF-style aperture
correction on the residuals.  ``scipy.optimize.minimize(method='Nelder-Mead')`` is called directly, exactly as the reference
does (:152-153); :func:`nelder_mead` restates that scipy routine (``_minimize_neldermead``, scipy 1.7.3 = the reference's pin,
no bounds, non-adaptive) for the device kernel to follow and is pinned against scipy in the tests.

Pinned by ``tests/golden/golden_psfphot.npz``: the reference's own ``PSFPhotometry.do_photometry`` executed through
``tests/golden/_refstub.py``.
"""

import numpy as np
from scipy.optimize import minimize
from .aperture import minimum_aperture, STATUS_OK
from .utilities import mag2flux


def lhood(params, psf, img, bkg, n_readout=900, readnoise=10, gain=100, cutoff_radius=5):
	"""``_lhood(params, img, bkg)`` with ``lhood_stat='Gaussian_d', include_bkg=True`` (psf_photometry.py:52-90)."""
	params = params.reshape(len(params)//3, 3)
	minweight = 1e-9
	minvar = 1e-9
	mdl = psf.integrate_to_image(params, cutoff_radius=cutoff_radius)
	var = np.abs(img + bkg)
	var += n_readout * readnoise**2 / gain**2
	var[var < minvar] = minvar
	weightmap = 1 / var
	weightmap[weightmap < minweight] = minweight
	return np.nansum(weightmap * (img - mdl)**2)


def nelder_mead(func, x0, maxiter, xatol=1e-4, fatol=1e-4):
	"""
	``scipy.optimize._optimize._minimize_neldermead`` (1.7.3; ``adaptive=False``, no bounds, ``maxfev = inf`` because
	``maxiter`` is given).  Returns ``(x, fval, success, iterations, nfev)``.
	"""
	rho, chi, psi, sigma = 1.0, 2.0, 0.5, 0.5
	nonzdelt, zdelt = 0.05, 0.00025
	x0 = np.asarray(x0, dtype='float64').flatten()
	N = len(x0)
	sim = np.empty((N + 1, N), dtype=x0.dtype)
	sim[0] = x0
	for k in range(N):
		y = np.array(x0, copy=True)
		if y[k] != 0:
			y[k] = (1 + nonzdelt)*y[k]
		else:
			y[k] = zdelt
		sim[k + 1] = y
	nfev = 0
	fsim = np.empty((N + 1,), float)
	for k in range(N + 1):
		fsim[k] = func(sim[k])
		nfev += 1
	ind = np.argsort(fsim)
	fsim = np.take(fsim, ind, 0)
	sim = np.take(sim, ind, 0)
	iterations = 1
	while iterations < maxiter:
		if (np.max(np.ravel(np.abs(sim[1:] - sim[0]))) <= xatol and np.max(np.abs(fsim[0] - fsim[1:])) <= fatol):
			break
		xbar = np.add.reduce(sim[:-1], 0) / N
		xr = (1 + rho) * xbar - rho * sim[-1]
		fxr = func(xr); nfev += 1
		doshrink = 0
		if fxr < fsim[0]:
			xe = (1 + rho * chi) * xbar - rho * chi * sim[-1]
			fxe = func(xe); nfev += 1
			if fxe < fxr:
				sim[-1] = xe
				fsim[-1] = fxe
			else:
				sim[-1] = xr
				fsim[-1] = fxr
		else: # fsim[0] <= fxr
			if fxr < fsim[-2]:
				sim[-1] = xr
				fsim[-1] = fxr
			else: # fxr >= fsim[-2]
				if fxr < fsim[-1]:
					xc = (1 + psi * rho) * xbar - psi * rho * sim[-1]
					fxc = func(xc); nfev += 1
					if fxc <= fxr:
						sim[-1] = xc
						fsim[-1] = fxc
					else:
						doshrink = 1
				else:
					xcc = (1 - psi) * xbar + psi * sim[-1]
					fxcc = func(xcc); nfev += 1
					if fxcc < fsim[-1]:
						sim[-1] = xcc
						fsim[-1] = fxcc
					else:
						doshrink = 1
				if doshrink:
					for j in range(1, N + 1):
						sim[j] = sim[0] + sigma * (sim[j] - sim[0])
						fsim[j] = func(sim[j]); nfev += 1
		ind = np.argsort(fsim)
		sim = np.take(sim, ind, 0)
		fsim = np.take(fsim, ind, 0)
		iterations += 1
	return sim[0], np.min(fsim), iterations < maxiter, iterations, nfev


def select_stars(catalog, target_pos_row_stamp, target_pos_column_stamp, target_tmag):
	"""psf_photometry.py:117-130: stars within 5 px and not more than 5 mag fainter, the five closest, sorted by distance."""
	dist = np.sqrt((target_pos_row_stamp - catalog['row_stamp'])**2 + (target_pos_column_stamp - catalog['column_stamp'])**2)
	keep = np.flatnonzero((dist < 5) & (target_tmag - catalog['tmag'] > -5))
	order = keep[np.argsort(dist[keep], kind='stable')]   # astropy Table.sort('dist')
	return order[:5]


def do_photometry(images, backgrounds, psf, catalog, stamp, target_pos_row, target_pos_column, target_tmag, aperture,
	n_readout=900, readnoise=10, gain=100, cutoff_radius=5, use_scipy=True):
	"""
	psf_photometry.py:111-196 for one target.  ``images, backgrounds``: ``(H, W, T)`` float32 cubes; ``psf``: :class:`oracle.psf.PSF`;
	``catalog``: dict with ``row_stamp, column_stamp, tmag``.  Returns dict(status, flux, flux_err, pos_centroid, params
	``(T, nstars, 3)``, success ``(T,)``, nit ``(T,)``).
	"""
	T = images.shape[2]
	sel = select_stars(catalog, target_pos_row - stamp[0], target_pos_column - stamp[2], target_tmag)
	params0 = np.empty((len(sel), 3), dtype='float64')
	for k, i in enumerate(sel):
		params0[k, :] = [catalog['row_stamp'][i], catalog['column_stamp'][i], mag2flux(catalog['tmag'][i])]
	params0 = params0.flatten()
	mini_aperture = minimum_aperture(stamp, target_pos_row, target_pos_column, aperture)
	out = {'flux': np.zeros(T), 'flux_err': np.zeros(T), 'pos_centroid': np.zeros((T, 2)), 'params': np.full((T, len(sel), 3), np.nan),
		'success': np.zeros(T, dtype=bool), 'nit': np.zeros(T, dtype='int64'), 'selected': sel, 'mini_aperture': mini_aperture}
	for k in range(T):
		img = images[:, :, k]
		bkg = backgrounds[:, :, k]
		maxiter = 500 if k > 0 else 1500
		args = (psf, img, bkg, n_readout, readnoise, gain, cutoff_radius)
		if use_scipy:
			res = minimize(lhood, params0, args=args, method='Nelder-Mead', options={'maxiter': maxiter})
			x, success, nit = res.x, res.success, res.nit
		else:
			x, _, success, nit, _ = nelder_mead(lambda p: lhood(p, *args), params0, maxiter)
		out['success'][k], out['nit'][k] = success, nit
		if success:
			result = np.array(x.reshape(len(x)//3, 3))
			target_flux = result[0, 2]
			best_fit = psf.integrate_to_image(result, cutoff_radius=cutoff_radius)
			residuals = img - best_fit
			flux_ap = np.nansum(residuals[mini_aperture])
			target_flux += flux_ap
			out['flux'][k] = target_flux
			out['flux_err'][k] = np.nan
			out['pos_centroid'][k] = result[0, 0:2]
			out['params'][k] = result
			params0 = x
		else:
			out['flux'][k] = np.nan
			out['flux_err'][k] = np.nan
			out['pos_centroid'][k] = [np.nan, np.nan]
	out['status'] = STATUS_OK
	return out
