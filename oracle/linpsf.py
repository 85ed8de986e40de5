# -*- coding: utf-8 -*-
"""
ORACLE (test infrastructure only) -- P3 / P4: linear PSF photometry.

Restates ``photometry/linpsf_photometry.py``: ``lsfit`` (:22-34) and
``LinPSFPhotometry.do_photometry`` (:79-219), keeping the per-cadence Python loop
(this is also the timed CPU baseline of the LinPSF path).
"""

import numpy as np
from .aperture import (minimum_aperture, allnan, STATUS_OK, STATUS_ERROR, STATUS_WARNING)


def lsfit(A, b):
	"""linpsf_photometry.py:22-34"""
	try:
		return (np.linalg.pinv(A.T.dot(A)).dot(A.T)).dot(b)
	except np.linalg.LinAlgError:
		pass
	return np.linalg.lstsq(A, b, rcond=None)[0]


def select_stars(catalog, target_starid):
	"""linpsf_photometry.py:87-104.  Returns ``(indx bool, staridx in reduced catalog)``."""
	starid = np.asarray(catalog['starid'])
	staridx = np.squeeze(np.where(starid == target_starid))
	dist = np.sqrt((catalog['row_stamp'][staridx] - catalog['row_stamp'])**2
		+ (catalog['column_stamp'][staridx] - catalog['column_stamp'])**2)
	indx = (dist < 5) & (catalog['tmag'][staridx]-catalog['tmag'] > -5)
	staridx_red = np.squeeze(np.where(starid[indx] == target_starid))
	return indx, staridx_red


def do_photometry(images, psf, catalog, target_starid, positions, stamp, target_pos_row, target_pos_column, aperture,
	cutoff_radius=5):
	"""
	linpsf_photometry.py:79-219 for one target.

	Parameters:
		images: ``(H, W, T)`` float32 cube.
		psf: :class:`oracle.psf.PSF`.
		catalog: dict of arrays (``starid, tmag, row_stamp, column_stamp``) at the reference time.
		positions: ``(T, nstars_all, 2)`` per-cadence ``(row_stamp, column_stamp)`` of every
			catalog star -- what ``catalog_attime`` returns (BasePhotometry.py:1224-1258);
			the WCS/jitter interpolation itself is host geometry and an engine *input*.

	Returns dict(status, flux, flux_err, contamination, fluxes_mean, A_last, nstars, staridx, errors).
	"""
	T = images.shape[2]
	indx, staridx = select_stars(catalog, target_starid)
	nstars = int(np.sum(indx))
	fluxes_sum = np.zeros(nstars, dtype='float64')
	flux = np.zeros(T, dtype='float64')
	flux_err = np.zeros(T, dtype='float64')
	mini_aperture = minimum_aperture(stamp, target_pos_row, target_pos_column, aperture)
	res = {'errors': [], 'nstars': nstars, 'staridx': int(staridx), 'indx': indx}
	A = None
	for k in range(T):
		img = images[:, :, k]
		rows_k = positions[k, indx, 0]
		cols_k = positions[k, indx, 1]

		good_pixels = np.isfinite(img)
		npx = int(np.sum(good_pixels))

		A = np.empty([npx, nstars], dtype='float64')
		for col in range(nstars):
			params0 = np.atleast_2d([rows_k[col], cols_k[col], 1.])
			A[:, col] = psf.integrate_to_image(params0, cutoff_radius=cutoff_radius)[good_pixels].flatten()

		b = img[good_pixels].flatten()

		try:
			fluxes = lsfit(A, b)
		except np.linalg.LinAlgError:
			fluxes = None

		if fluxes is None:
			flux[k] = np.nan
			flux_err[k] = np.nan
		else:
			target_flux = fluxes[staridx]
			# (the reference also computes an aperture correction on the residuals here,
			#  linpsf_photometry.py:158-165, but never applies it to the stored flux)
			flux[k] = target_flux
			flux_err[k] = np.nan
			fluxes_sum += fluxes

	res['flux'] = flux
	res['flux_err'] = flux_err
	res['mini_aperture'] = mini_aperture
	if allnan(flux):
		res['errors'].append('All target flux values are NaN.')
		res['status'] = STATUS_ERROR
		return res

	fluxes_mean = fluxes_sum / np.sum(~np.isnan(flux))
	not_target_star = np.arange(len(fluxes_mean)) != staridx
	contamination = np.sum(A[:, not_target_star].dot(fluxes_mean[not_target_star]) * A[:, staridx]) / fluxes_mean[staridx]
	res['fluxes_mean'] = fluxes_mean
	res['contamination'] = contamination
	res['A_last'] = A
	if contamination > 0.1:
		res['errors'].append('High contamination')
		res['status'] = STATUS_WARNING
		return res
	res['status'] = STATUS_OK
	return res
