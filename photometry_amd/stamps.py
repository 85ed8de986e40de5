# -*- coding: utf-8 -*-
"""
Stamp geometry and the stamp-resize decisions of the aperture plugin, as pure host functions shared by the per-target
plugin (``plugins.AperturePhotometry``) and the batched pipeline (``pipeline.aperture_frames``).

What they must reproduce (behaviour, not code): ``BasePhotometry.default_stamp`` / ``resize_stamp`` / ``_set_stamp``
(photometry/BasePhotometry.py:521-693) and the retry loop of ``AperturePhotometry.do_photometry``
(photometry/AperturePhotometry/photometry.py:75-170): a mask touching a stamp edge asks for 10 more pixels on that side,
the stamp is clipped to the CCD region that exists, an unchanged stamp ends the loop, at most 5 (10 below Tmag 6)
attempts, and for bright targets a side that can no longer grow while carrying more than ``flux_limit`` of the expected
flux ends the attempt early (the "haloswitch quick break").

A stamp is ``(row_min, row_max, col_min, col_max)``: half-open, 0-based CCD coordinates (BasePhotometry.py:643-662).
"""

import numpy as np

#: rows: Tmag, stamp height, stamp width -- the lookup table of BasePhotometry.default_stamp (BasePhotometry.py:541-560),
#: fitted upstream to the size a saturated star's bleed columns need
_TMAG_HEIGHT_WIDTH = np.array([
	(0.0, 831.98319063, 157.71602062), (0.52631579, 533.58494422, 125.1238281), (1.05263158, 344.0840884, 99.99440209),
	(1.57894737, 223.73963332, 80.61896267), (2.10526316, 147.31365728, 65.6799962), (2.63157895, 98.77856016, 54.16166547),
	(3.15789474, 67.95585074, 45.28073365), (3.68421053, 48.38157414, 38.4333048), (4.21052632, 35.95072974, 33.15375951),
	(4.73684211, 28.05639497, 28.05639497), (5.26315789, 23.043017, 23.043017), (5.78947368, 19.85922009, 19.85922009),
	(6.31578947, 17.83731732, 17.83731732), (6.84210526, 16.5532873, 16.5532873), (7.36842105, 15.73785092, 15.73785092),
	(7.89473684, 15.21999971, 15.21999971), (8.42105263, 14.89113301, 14.89113301), (8.94736842, 14.68228285, 14.68228285),
	(9.47368421, 14.54965042, 14.54965042), (10.0, 14.46542084, 14.46542084), (13.0, 14.0, 14.0)])

MIN_STAMP = 15      #: smallest FFI stamp side (BasePhotometry.py:561-562)
RESIZE_STEP = 10    #: pixels added per touched side and attempt (photometry.py:124-131)

#: (name of the resize argument, edge bit in the device flags, index into the stamp tuple, direction of growth)
SIDES = (('down', 2, 0, -1), ('up', 4, 1, +1), ('left', 8, 2, -1), ('right', 16, 3, +1))


def default_stamp_size(tmag):
	"""(rows, columns) of the default FFI stamp of a star of magnitude ``tmag``, as floats like the reference returns them."""
	t = _TMAG_HEIGHT_WIDTH
	size = [max(np.ceil(np.interp(tmag, t[:, 0], t[:, k])), MIN_STAMP) for k in (1, 2)]
	return size[0], size[1]


def centred_stamp(pos_row, pos_column, n_rows, n_columns):
	"""A stamp of ``2*(n//2) + 1`` pixels per axis centred on the pixel nearest to the position (BasePhotometry.py:646-651)."""
	r, c = int(np.round(pos_row)), int(np.round(pos_column))
	hr, hc = n_rows // 2, n_columns // 2
	return (r - hr, r + hr + 1, c - hc, c + hc + 1)


def clip_stamp(stamp, limits):
	"""Intersection with the region that exists on the CCD; ``ValueError`` if nothing is left (BasePhotometry.py:653-672)."""
	lo = (max(stamp[0], limits[0]), max(stamp[2], limits[2]))
	hi = (min(stamp[1], limits[1]), min(stamp[3], limits[3]))
	if lo[0] > hi[0] or lo[1] > hi[1]:
		raise ValueError("Invalid stamp selected")
	return (int(lo[0]), int(hi[0]), int(lo[1]), int(hi[1]))


def default_stamp(pos_row, pos_column, tmag, limits):
	n_rows, n_columns = default_stamp_size(tmag)
	return clip_stamp(centred_stamp(pos_row, pos_column, n_rows, n_columns), limits)


def default_stamps(pos_rows, pos_columns, tmags, limits):
	"""
	:func:`default_stamp` for a whole batch: int64 ``(n, 4)`` stamps and a bool vector that is False where the clipped stamp is
	empty (``default_stamp`` raises ``ValueError`` there).  Same arithmetic, vectorised.
	"""
	t = _TMAG_HEIGHT_WIDTH
	tm = np.asarray(tmags, dtype='float64')
	n_rows = np.maximum(np.ceil(np.interp(tm, t[:, 0], t[:, 1])), MIN_STAMP)
	n_cols = np.maximum(np.ceil(np.interp(tm, t[:, 0], t[:, 2])), MIN_STAMP)
	r = np.round(np.asarray(pos_rows, dtype='float64')).astype('int64')
	c = np.round(np.asarray(pos_columns, dtype='float64')).astype('int64')
	hr, hc = (n_rows // 2).astype('int64'), (n_cols // 2).astype('int64')
	st = np.stack((np.maximum(r - hr, limits[0]), np.minimum(r + hr + 1, limits[1]),
		np.maximum(c - hc, limits[2]), np.minimum(c + hc + 1, limits[3])), axis=1)
	valid = (st[:, 0] <= st[:, 1]) & (st[:, 2] <= st[:, 3])
	return st, valid


def moved(stamp, limits, pos_row=None, pos_column=None, down=None, up=None, left=None, right=None, width=None, height=None):
	"""The stamp after a ``resize_stamp(...)`` request (BasePhotometry.py:567-613), clipped; may equal ``stamp``."""
	st = list(stamp)
	for (name, _bit, idx, sign), amount in zip(SIDES, (down, up, left, right)):
		if amount:
			st[idx] += sign * amount
	if height:
		st[0], st[1] = centred_stamp(pos_row, pos_column, height, 1)[:2]
	if width:
		st[2], st[3] = centred_stamp(pos_row, pos_column, 1, width)[2:]
	return clip_stamp(st, limits)


def edge_requests(flags):
	"""Resize arguments for the edges the mask touches, from the edge bits of the device ``flags`` (photometry.py:123-131)."""
	return {name: RESIZE_STEP for name, bit, _idx, _sign in SIDES if flags & bit}


def retry_limit(tmag):
	"""photometry.py:70-73"""
	return 10 if tmag < 6 else 5


def stuck_sides(before, after, requests):
	"""Sides that were asked to grow but did not move: the stamp has reached the limit of the data there."""
	return [name for name, _bit, idx, _sign in SIDES if requests.get(name) and before[idx] == after[idx]]


def edge_pixels(shape, sides):
	"""Boolean image of the outermost row / column of every listed side."""
	e = np.zeros(shape, dtype=bool)
	where = {'down': (0, slice(None)), 'up': (-1, slice(None)), 'left': (slice(None), 0), 'right': (slice(None), -1)}
	for s in sides:
		e[where[s]] = True
	return e


def quick_break_flux(sumimage, mask, before, after, requests):
	"""
	Flux of the in-mask pixels on the stamp edges that can grow no further (photometry.py:146-158), or ``None`` when every
	requested side did move.  The caller compares it with ``flux_limit`` times the expected flux of the target.
	"""
	stuck = stuck_sides(before, after, requests)
	if not stuck:
		return None
	return float(np.nansum(np.asarray(sumimage)[np.asarray(mask, dtype=bool) & edge_pixels(np.shape(mask), stuck)]))
