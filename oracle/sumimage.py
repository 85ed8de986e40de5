# -*- coding: utf-8 -*-
"""
ORACLE (test infrastructure only) -- A1: the sum image.

Follows ``BasePhotometry.sumimage`` TPF branch (photometry/BasePhotometry.py:1008-1019)
and the FFI prepare stage (photometry/prepare.py:348-349, 450-453, 459): per-pixel mean
over the cadences whose ``quality & 4335 == 0`` (quality.py:123-124), non-finite pixels
excluded from both the sum and the count, zero count -> NaN.  The accumulator is float64,
the pixels are float32, the cadences are added in time order.

(The two reference branches differ only for +-inf pixels: prepare.py:452 replaces only
NaN by zero, BasePhotometry.py:1013 zeroes every non-finite value.  This restatement --
and the device kernel -- follow the per-target branch, BasePhotometry.py:1011-1015.)
"""

import numpy as np
from .quality import tess_filter, TESS_DEFAULT_BITMASK


def sumimage(cube, quality, bitmask=TESS_DEFAULT_BITMASK):
	"""
	Parameters:
		cube (ndarray): ``(H, W, T)`` float32 image cube (BasePhotometry.py:732 layout).
		quality (ndarray): ``(T,)`` integer quality flags.

	Returns:
		ndarray: ``(H, W)`` float64 sum image.
	"""
	cube = np.asarray(cube)
	H, W, T = cube.shape
	S = np.zeros((H, W), dtype='float64')
	Nimg = np.zeros((H, W), dtype='int32')
	good = tess_filter(quality, bitmask)
	for k in range(T):
		if good[k]:
			img = cube[:, :, k]
			isgood = np.isfinite(img)
			Nimg += isgood
			S += np.where(isgood, img, np.float32(0))
	out = np.full((H, W), np.nan, dtype='float64')
	ok = (Nimg > 0)
	out[ok] = S[ok] / Nimg[ok]
	return out


def sumimage_batch(cubes, quality, bitmask=TESS_DEFAULT_BITMASK):
	"""Vectorised over targets: ``cubes`` is ``(Nt, H, W, T)``; ``quality`` ``(T,)`` or ``(Nt, T)``.
	Sequential accumulation in time order is kept (cumulative float64 adds)."""
	cubes = np.asarray(cubes)
	Nt = cubes.shape[0]
	quality = np.asarray(quality)
	out = np.empty(cubes.shape[:3], dtype='float64')
	for i in range(Nt):
		q = quality if quality.ndim == 1 else quality[i]
		out[i] = sumimage(cubes[i], q, bitmask)
	return out
