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
from . import sumimage as sumimage_oracle
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


#--------------------------------------------------------------------------------------------------
# The plugin INCLUDING its stamp handling: what run_tessphot does for one FFI target
#--------------------------------------------------------------------------------------------------
def default_stamp(tmag):
	"""``BasePhotometry.default_stamp`` (BasePhotometry.py:521-564)."""
	tm = np.array([0.0, 0.52631579, 1.05263158, 1.57894737, 2.10526316,
		2.63157895, 3.15789474, 3.68421053, 4.21052632, 4.73684211,
		5.26315789, 5.78947368, 6.31578947, 6.84210526, 7.36842105,
		7.89473684, 8.42105263, 8.94736842, 9.47368421, 10.0, 13.0])
	height = np.array([831.98319063, 533.58494422, 344.0840884, 223.73963332,
		147.31365728, 98.77856016, 67.95585074, 48.38157414,
		35.95072974, 28.05639497, 23.043017, 19.85922009,
		17.83731732, 16.5532873, 15.73785092, 15.21999971,
		14.89113301, 14.68228285, 14.54965042, 14.46542084, 14.0])
	width = np.array([157.71602062, 125.1238281, 99.99440209, 80.61896267,
		65.6799962, 54.16166547, 45.28073365, 38.4333048,
		33.15375951, 28.05639497, 23.043017, 19.85922009,
		17.83731732, 16.5532873, 15.73785092, 15.21999971,
		14.89113301, 14.68228285, 14.54965042, 14.46542084, 14.0])
	Ncolumns = np.interp(tmag, tm, width)
	Nrows = np.interp(tmag, tm, height)
	Nrows = np.maximum(np.ceil(Nrows), 15)
	Ncolumns = np.maximum(np.ceil(Ncolumns), 15)
	return Nrows, Ncolumns


class FrameTarget(object):
	"""
	The stamp state of one FFI target of the reference's ``BasePhotometry``: ``_set_stamp`` / ``resize_stamp``
	(BasePhotometry.py:567-693), the FFI branch of ``_load_cube`` (:720-742), ``sumimage`` (:1008-1019), ``aperture`` (:1043)
	and the catalogue of the stamp plus its 5-pixel buffer (:1094-1181), over frames held in memory.

	``frames``: dict of float32 ``(R, C, T)`` arrays ``images, images_err, backgrounds`` covering CCD rows ``[row0, row0+R)``
	and columns ``[col0, col0+C)``; ``catalog``: dict of arrays ``starid, tmag, row, column`` of every star of the region.
	"""

	def __init__(self, frames, row0, col0, quality, catalog, starid, tmag, pos_row, pos_column):
		self.frames, self.row0, self.col0 = frames, int(row0), int(col0)
		R, C, _ = frames['images'].shape
		self._max_stamp = (self.row0, self.row0 + R, self.col0, self.col0 + C)
		self.quality, self.catalog_all = np.asarray(quality), catalog
		self.starid, self.tmag = starid, tmag
		self.target_pos_row, self.target_pos_column = pos_row, pos_column
		self._stamp = None
		self.stamp_resizes = 0
		self._set_stamp()

	def _set_stamp(self, compare_stamp=None):
		if not self._stamp:
			Nrows, Ncolumns = default_stamp(self.tmag)
			self._stamp = (
				int(np.round(self.target_pos_row)) - Nrows//2,
				int(np.round(self.target_pos_row)) + Nrows//2 + 1,
				int(np.round(self.target_pos_column)) - Ncolumns//2,
				int(np.round(self.target_pos_column)) + Ncolumns//2 + 1
			)
		st = list(self._stamp)
		st[0] = int(np.maximum(st[0], self._max_stamp[0]))
		st[1] = int(np.minimum(st[1], self._max_stamp[1]))
		st[2] = int(np.maximum(st[2], self._max_stamp[2]))
		st[3] = int(np.minimum(st[3], self._max_stamp[3]))
		self._stamp = tuple(st)
		if self._stamp[0] > self._stamp[1] or self._stamp[2] > self._stamp[3]:
			raise ValueError("Invalid stamp selected")
		if self._stamp == compare_stamp:
			return False
		self.target_pos_row_stamp = self.target_pos_row - self._stamp[0]
		self.target_pos_column_stamp = self.target_pos_column - self._stamp[2]
		return True

	def resize_stamp(self, down=None, up=None, left=None, right=None):
		old_stamp = self._stamp
		st = list(self._stamp)
		if up:
			st[1] += up
		if down:
			st[0] -= down
		if left:
			st[2] -= left
		if right:
			st[3] += right
		self._stamp = tuple(st)
		stamp_changed = self._set_stamp(compare_stamp=old_stamp)
		if stamp_changed:
			self.stamp_resizes += 1
		return stamp_changed

	@property
	def stamp(self):
		return self._stamp

	def cube(self, name):
		r1, r2, c1, c2 = self._stamp
		return np.ascontiguousarray(self.frames[name][r1 - self.row0:r2 - self.row0, c1 - self.col0:c2 - self.col0, :])

	def sumimage(self):
		return sumimage_oracle.sumimage(self.cube('images'), self.quality)

	def catalog(self, buffer_size=5):
		r1, r2, c1, c2 = self._stamp
		row, col = np.asarray(self.catalog_all['row']), np.asarray(self.catalog_all['column'])
		sel = (row >= r1 - 0.5 - buffer_size) & (row < r2 - 0.5 + buffer_size) & (col >= c1 - 0.5 - buffer_size) & (col < c2 - 0.5 + buffer_size)
		c = {k: np.asarray(v)[sel] for k, v in self.catalog_all.items()}
		col64, row64 = np.asarray(c['column'], dtype='float64'), np.asarray(c['row'], dtype='float64')
		return {'starid': np.asarray(c['starid'], dtype='int64'), 'tmag': np.asarray(c['tmag'], dtype='float32'),
			'column': col64.astype('float32'), 'row': row64.astype('float32'),
			'column_stamp': (col64 - self._stamp[2]).astype('float32'), 'row_stamp': (row64 - self._stamp[0]).astype('float32')}


def photometry_on_frames(tgt, haloswitch=(6.0, 0.01)):
	"""
	``AperturePhotometry.do_photometry`` (photometry.py:44-257) with its stamp-resize loop (:75-170) for a
	:class:`FrameTarget`.  Returns the dict of :func:`do_photometry` plus ``stamp`` and ``stamp_resizes``.
	"""
	res = {'status': STATUS_UNKNOWN, 'errors': [], 'details': {}, 'additional_headers': {}}
	ExpectedFlux = mag2flux(tgt.tmag)
	haloswitch_tmag_limit, haloswitch_flux_limit = haloswitch
	allow_retries = 5
	if tgt.tmag < 6:
		allow_retries = 10

	resize_args = {}
	for retries in range(allow_retries):
		SumImage = tgt.sumimage()
		catalog = tgt.catalog()
		cat = np.column_stack((catalog['column_stamp'], catalog['row_stamp'], catalog['tmag']))
		try:
			mm, _ = k2p2_oracle.k2p2FixFromSum(SumImage, catalog=cat, **K2P2_SETTINGS)
			mm = None if mm is None else np.asarray(mm, dtype='bool')
		except k2p2_oracle.K2P2NoStars:
			res['errors'].append('ERROR: No flux above threshold.')
			mm = None
		mask_main, using_minimum_mask, err = select_mask(mm, tgt.target_pos_row_stamp, tgt.target_pos_column_stamp)
		if err is not None:
			res['errors'].append('ERROR: ' + err)
			res['status'] = STATUS_ERROR
			res['stamp'], res['stamp_resizes'] = tgt.stamp, tgt.stamp_resizes
			return res
		if using_minimum_mask:
			res['errors'].append('WARNING: No masks found. Using minimum aperture.' if mm is None
				else 'WARNING: No mask found for main target. Using minimum aperture.')
			aperture = np.asarray(np.isfinite(SumImage), dtype='int32') # BasePhotometry.py:1043
			mask_main = minimum_aperture(tgt.stamp, tgt.target_pos_row, tgt.target_pos_column, aperture)

		resize_args = edge_flags(mask_main)
		if resize_args:
			stamp_before = tgt.stamp
			sumimage_before = SumImage
			if not tgt.resize_stamp(**resize_args):
				resize_args = {}
				res['errors'].append('WARNING: Could not resize stamp any further.')
				break
			if tgt.tmag <= haloswitch_tmag_limit:
				edge = np.zeros_like(mask_main, dtype='bool')
				if resize_args.get('down') and tgt.stamp[0] == stamp_before[0]:
					edge[0, :] = True
				if resize_args.get('up') and tgt.stamp[1] == stamp_before[1]:
					edge[-1, :] = True
				if resize_args.get('left') and tgt.stamp[2] == stamp_before[2]:
					edge[:, 0] = True
				if resize_args.get('right') and tgt.stamp[3] == stamp_before[3]:
					edge[:, -1] = True
				if np.any(edge):
					EdgeFlux = np.nansum(sumimage_before[mask_main & edge])
					if EdgeFlux/ExpectedFlux > haloswitch_flux_limit:
						res['errors'].append('ERROR: Stamp resize hit limit. Haloswitch quick break.')
						res['details']['edge_flux'] = EdgeFlux
						res['status'] = STATUS_ERROR
						res['stamp'], res['stamp_resizes'] = tgt.stamp, tgt.stamp_resizes
						return res
		else:
			break

	res['stamp'], res['stamp_resizes'] = tgt.stamp, tgt.stamp_resizes
	if resize_args:
		res['errors'].append('ERROR: Too many stamp resizes.')
		res['status'] = STATUS_ERROR
		return res

	lc = extract(tgt.cube('images'), tgt.cube('images_err'), tgt.cube('backgrounds'), mask_main, tgt.stamp)
	res.update(lc)
	res['mask'] = mask_main
	res['sumimage'] = SumImage
	res['using_minimum_mask'] = using_minimum_mask
	cont, my_status, target_in_mask, skip_targets = contamination(mask_main, tgt.stamp, catalog, tgt.starid, tgt.tmag)
	if my_status == STATUS_ERROR:
		res['errors'].append('ERROR: No targets in mask.')
	res['contamination'] = cont
	res['skip_targets'] = skip_targets
	if using_minimum_mask:
		my_status = STATUS_WARNING
	res['status'] = my_status
	return res
