# -*- coding: utf-8 -*-
"""
ORACLE (test infrastructure only) -- A5b / A6 / A7: the AperturePhotometry plugin.

Restates ``AperturePhotometry.do_photometry``
(photometry/AperturePhotometry/photometry.py:44-257) on plain arrays, keeping the
reference's per-cadence Python loop call-for-call (this is also the timed CPU baseline).

Arithmetic notes (all reproduced here and on the device):
* ``img``/``imgerr``/``bck`` are float32 ``(H, W)`` views of ``(H, W, T)`` cubes
  (BasePhotometry.py:732) so ``np.sum`` (photometry.py:188-189) is numpy's *float32
  pairwise* sum over the masked pixels in raster order.
* flux uses ``np.sum`` (NaN-propagating, :188), background uses ``np.nansum`` (:201; only
  ``allnan`` is imported from bottleneck, :10): NaN replaced by 0, then the same float32
  pairwise ``np.sum``.
* centroid = ``np.average(members, weights=float32)`` -> float64 accumulation (:194).
* pixel coordinates are 1-based int32 CCD coordinates (BasePhotometry.py:696-706).
"""

import logging
import numpy as np
from . import k2p2 as k2p2_oracle
from .utilities import mag2flux

STATUS_UNKNOWN, STATUS_OK, STATUS_ERROR, STATUS_WARNING, STATUS_ABORT, STATUS_SKIPPED, STATUS_STARTED = 0, 1, 2, 3, 4, 5, 6

#: photometry.py:54-64
K2P2_SETTINGS = {
	'thresh': 0.8,
	'min_no_pixels_in_mask': 4,
	'min_for_cluster': 4,
	'cluster_radius': np.sqrt(2) + np.finfo(np.float64).eps,
	'segmentation': True,
	'ws_blur': 0.5,
	'ws_thres': 0,
	'ws_footprint': 3,
	'extend_overflow': True
}


def allnan(x):
	"""bottleneck.allnan (True for empty input)."""
	return bool(np.all(np.isnan(x)))


def get_pixel_grid(stamp):
	"""BasePhotometry.get_pixel_grid (BasePhotometry.py:696-706): 1-based (cols, rows)."""
	return np.meshgrid(
		np.arange(stamp[2]+1, stamp[3]+1, 1, dtype='int32'),
		np.arange(stamp[0]+1, stamp[1]+1, 1, dtype='int32')
	)


def minimum_aperture(stamp, target_pos_row, target_pos_column, aperture):
	"""``_minimum_aperture`` (photometry.py:31-41)."""
	collected_pixels = (aperture & 1 != 0)
	cols, rows = get_pixel_grid(stamp)
	mask_main = ((np.abs(cols - target_pos_column - 1) <= 1)
		& (np.abs(rows - target_pos_row - 1) <= 1))
	return mask_main & collected_pixels


def select_mask(masks, target_pos_row_stamp, target_pos_column_stamp):
	"""
	photometry.py:99-120.  ``masks`` is ``None`` (no masks), or a ``(n, H, W)`` bool array.

	Returns ``(mask_main or None, using_minimum_mask, error)``; ``mask_main is None`` and
	``using_minimum_mask`` means the caller must substitute the minimum aperture.
	"""
	if masks is None or np.ndim(masks) == 0:
		return None, True, None
	masks = np.asarray(masks, dtype='bool')
	# Python round() on numpy float: round-half-to-even (photometry.py:107)
	r = int(round(float(target_pos_row_stamp)))
	c = int(round(float(target_pos_column_stamp)))
	indx_main = masks[:, r, c].flatten() # may raise IndexError exactly like the reference
	if not np.any(indx_main):
		return None, True, None
	elif np.sum(indx_main) > 1:
		return None, False, 'Too many masks.'
	return masks[indx_main, :, :].reshape(masks.shape[1:]), False, None


def edge_flags(mask_main):
	"""photometry.py:123-131 -> dict of resize args; bit image: 1=down(row 0) 2=up 4=left 8=right."""
	resize_args = {}
	if np.any(mask_main[0, :]):
		resize_args['down'] = 10
	if np.any(mask_main[-1, :]):
		resize_args['up'] = 10
	if np.any(mask_main[:, 0]):
		resize_args['left'] = 10
	if np.any(mask_main[:, -1]):
		resize_args['right'] = 10
	return resize_args


def extract(images, images_err, backgrounds, mask_main, stamp):
	"""
	A6: the extraction loop (photometry.py:172-201).

	Parameters:
		images, images_err, backgrounds: ``(H, W, T)`` float32 cubes; ``backgrounds=None`` (aperture-only
			run, BASELINE configs[1]: no background cube exists) leaves ``flux_background`` NaN.
		mask_main: ``(H, W)`` bool.
		stamp: ``(row_min, row_max, col_min, col_max)``.

	Returns:
		dict with float64 ``flux, flux_err, flux_background`` ``(T,)`` and ``pos_centroid`` ``(T, 2)``.
	"""
	T = images.shape[2]
	lc = {
		'flux': np.zeros(T, dtype='float64'),
		'flux_err': np.zeros(T, dtype='float64'),
		'flux_background': np.zeros(T, dtype='float64'),
		'pos_centroid': np.zeros((T, 2), dtype='float64'),
	}
	cols, rows = get_pixel_grid(stamp)
	members = np.column_stack((cols[mask_main], rows[mask_main]))

	for k in range(T):
		img = images[:, :, k]
		imgerr = images_err[:, :, k]
		bck = None if backgrounds is None else backgrounds[:, :, k]

		flux_in_cluster = img[mask_main]

		if allnan(flux_in_cluster) or np.all(flux_in_cluster == 0):
			lc['flux'][k] = np.nan
			lc['flux_err'][k] = np.nan
			lc['pos_centroid'][k, :] = np.nan
		else:
			lc['flux'][k] = np.sum(flux_in_cluster)
			lc['flux_err'][k] = np.sqrt(np.sum(imgerr[mask_main]**2))

			finite_vals = (flux_in_cluster > 0)
			if np.any(finite_vals):
				lc['pos_centroid'][k, :] = np.average(members[finite_vals], weights=flux_in_cluster[finite_vals], axis=0)
			else:
				lc['pos_centroid'][k, :] = np.nan

		if bck is None:
			lc['flux_background'][k] = np.nan
			continue
		bm = bck[mask_main]
		if allnan(bm):
			lc['flux_background'][k] = np.nan
		else:
			lc['flux_background'][k] = np.nansum(bm)
	return lc


def contamination(mask_main, stamp, catalog, target_starid, target_tmag):
	"""
	A7: photometry.py:220-250.

	``catalog`` is a dict of equally long arrays with at least ``starid`` (int64),
	``tmag``, ``row``, ``column`` (float32; CCD coordinates, BasePhotometry.py:1168-1169).

	Returns ``(contamination, status, target_in_mask (indices), skip_targets (starids))``.
	"""
	cols, rows = get_pixel_grid(stamp)
	n = len(catalog['starid'])
	target_in_mask = [k for k in range(n)
		if np.any(mask_main & (rows == np.round(catalog['row'][k])+1) & (cols == np.round(catalog['column'][k])+1))]

	my_status = STATUS_OK
	if len(target_in_mask) == 0:
		cont = np.nan
		my_status = STATUS_ERROR
	elif len(target_in_mask) == 1 and catalog['starid'][target_in_mask][0] == target_starid:
		cont = 0
	else:
		mags_in_mask = catalog['tmag'][target_in_mask]
		mags_total = -2.5*np.log10(np.nansum(10**(-0.4*mags_in_mask)))
		cont = 1.0 - 10**(0.4*(mags_total - target_tmag))
		cont = np.clip(cont, 0, None)

	skip_targets = [int(catalog['starid'][k]) for k in target_in_mask if catalog['starid'][k] != target_starid]
	return cont, my_status, target_in_mask, skip_targets


def do_photometry(sumimage, images, images_err, backgrounds, stamp,
	target_pos_row, target_pos_column, target_tmag, target_starid, catalog, aperture,
	masks='k2p2', resize_stamp=None, haloswitch=(6.0, 0.01), datasource='ffi'):
	"""
	Full plugin restatement (photometry.py:44-257) for one target on a fixed-size stamp.

	``masks='k2p2'`` runs the oracle K2P2 (:93); otherwise pass the ``(n,H,W)`` masks (or
	``None``) that ``k2p2FixFromSum`` would have returned (used by the golden tests where the
	reference's K2P2 is monkey-patched).

	``resize_stamp`` is a callable ``(**resize_args) -> bool``; the default (``None``)
	behaves like a fixed-size cube where the stamp cannot grow
	(BasePhotometry.py:605-612, 678-679 -> ``False``).

	Returns a dict with ``status`` and, unless ERROR occurred before extraction, the light
	curve, the final mask, ``contamination``, ``skip_targets``, ``using_minimum_mask`` and
	``errors`` (the strings the reference would have logged at WARNING/ERROR level).
	"""
	logger = logging.getLogger(__name__)
	res = {'status': STATUS_UNKNOWN, 'errors': [], 'details': {}, 'additional_headers': {}}
	target_pos_row_stamp = target_pos_row - stamp[0]
	target_pos_column_stamp = target_pos_column - stamp[2]

	ExpectedFlux = mag2flux(target_tmag)
	haloswitch_tmag_limit, haloswitch_flux_limit = haloswitch

	allow_retries = 5
	if target_tmag < 6:
		allow_retries = 10

	resize_args = {}
	for retries in range(allow_retries):
		SumImage = sumimage

		if isinstance(masks, str) and masks == 'k2p2':
			cat = np.column_stack((catalog['column_stamp'], catalog['row_stamp'], catalog['tmag']))
			try:
				mm, _ = k2p2_oracle.k2p2FixFromSum(SumImage, catalog=cat, **K2P2_SETTINGS)
				mm = None if mm is None else np.asarray(mm, dtype='bool')
			except k2p2_oracle.K2P2NoStars:
				res['errors'].append('ERROR: No flux above threshold.')
				mm = None
		else:
			mm = masks

		mask_main, using_minimum_mask, err = select_mask(mm, target_pos_row_stamp, target_pos_column_stamp)
		if err is not None:
			res['errors'].append('ERROR: ' + err)
			res['status'] = STATUS_ERROR
			return res
		if using_minimum_mask:
			res['errors'].append('WARNING: No masks found. Using minimum aperture.' if mm is None
				else 'WARNING: No mask found for main target. Using minimum aperture.')
			mask_main = minimum_aperture(stamp, target_pos_row, target_pos_column, aperture)

		resize_args = edge_flags(mask_main)
		res['edge'] = dict(resize_args)
		if resize_args:
			if resize_stamp is None or not resize_stamp(**resize_args):
				resize_args = {}
				res['errors'].append('WARNING: Could not resize stamp any further.')
				break
			# a successful resize would need new cubes: only the fixed-stamp path is restated.
			raise NotImplementedError("stamp resize inside the oracle")
		else:
			break

	if resize_args:
		res['errors'].append('ERROR: Too many stamp resizes.')
		res['status'] = STATUS_ERROR
		return res

	lc = extract(images, images_err, backgrounds, mask_main, stamp)
	res.update(lc)
	res['mask'] = mask_main
	res['using_minimum_mask'] = using_minimum_mask

	for key, hk in (('thresh', 'KP_THRES'), ('min_no_pixels_in_mask', 'KP_MIPIX'), ('min_for_cluster', 'KP_MICLS'),
		('cluster_radius', 'KP_CLSRA'), ('ws_blur', 'KP_WSBLR'), ('ws_thres', 'KP_WSTHR'), ('ws_footprint', 'KP_WSFOT')):
		res['additional_headers'][hk] = K2P2_SETTINGS[key]
	res['additional_headers']['KP_WS'] = bool(K2P2_SETTINGS['segmentation'])
	res['additional_headers']['KP_EX'] = bool(K2P2_SETTINGS['extend_overflow'])

	cont, my_status, target_in_mask, skip_targets = contamination(mask_main, stamp, catalog, target_starid, target_tmag)
	if my_status == STATUS_ERROR:
		res['errors'].append('ERROR: No targets in mask.')
	res['contamination'] = cont
	if not np.isnan(cont):
		res['additional_headers']['AP_CONT'] = cont
	res['target_in_mask'] = target_in_mask
	res['skip_targets'] = skip_targets
	if skip_targets:
		res['details']['skip_targets'] = skip_targets

	if using_minimum_mask:
		my_status = STATUS_WARNING
	res['status'] = my_status
	logger.debug("oracle aperture status %d", my_status)
	return res
