# -*- coding: utf-8 -*-
"""
The arithmetic of the reference's prepare stage (photometry/prepare.py:265-459, photometry/backgrounds.py:52-211) on a frame
stack resident in HBM: background estimation of every full-frame image, its smoothing in time, the subtraction with the
manual-exclude masking, the pixel flags and the sum image -- everything ``BasePhotometry`` later reads per target.

What runs where: every pixel-sized operation is a device kernel (``csrc/fullframe.hip``); the background mesh of a frame
(32 x 32 numbers for a 2048 x 2048 image) goes through its few small steps -- drop the mostly-masked cells, fill them from
their neighbours, 3 x 3 median filter, cubic-spline prefilter -- on the host with the same scipy calls photutils makes.
HDF5 / FITS I/O, WCS and the TESS-specific radial component of ``fit_background`` (backgrounds.py:110-154, 162-197: needs the
camera geometry of an ``FFIImage``) are not part of this module: ``fit_background_frames`` is the branch the reference takes
for a plain image (``bkgiters = 1``, :156-157), which is also what its own test exercises (tests/test_background.py:36-54).
"""

import ctypes
import numpy as np
from scipy import ndimage
from .engine import TESS_DEFAULT_BITMASK

#: PixelQualityFlags (photometry/quality.py:157-166)
PIXEL_NOT_USED_FOR_BACKGROUND = 1
PIXEL_MANUAL_EXCLUDE = 2
PIXEL_BACKGROUND_SHENANIGANS = 4


def finish_mesh(mesh, nmasked, box=64, exclude_percentile=50.0, filter_size=3):
	"""
	The low-resolution part of photutils ``Background2D`` (1.3.0) for one frame, after the per-cell statistics: cells with more
	than ``exclude_percentile`` % masked pixels are replaced by the inverse-distance weighted mean of the 10 nearest kept
	cells, then the 3 x 3 median filter.  Returns the filtered mesh (float64); raises ``ValueError`` when no cell is usable.
	"""
	mesh = np.array(mesh, dtype='float64', copy=True)
	keep = (np.asarray(nmasked) <= exclude_percentile / 100.0 * box * box) & np.isfinite(mesh)
	if not keep.any():
		raise ValueError(f"All meshes contain > {exclude_percentile} percent masked pixels")
	if not keep.all():
		ky, kx = np.nonzero(keep)
		kv = mesh[keep]
		for y, x in zip(*np.nonzero(~keep)):
			dist = np.hypot(ky - y, kx - x)
			nearest = np.argsort(dist, kind='stable')[:10]
			w = 1.0 / dist[nearest]
			mesh[y, x] = np.sum(w * kv[nearest]) / np.sum(w)
	if filter_size > 1:
		mesh = ndimage.generic_filter(mesh, np.nanmedian, size=filter_size, mode='constant', cval=np.nan)
	return mesh


def fit_background_frames(ctx, frames, flux_cutoff=8e4, box=64, exclude=None, out=None, return_mask=False):
	"""
	``fit_background`` (backgrounds.py:52-211, plain-image branch) for every frame of a stack.

	``frames``: float32 DeviceArray ``(T, R, C)``; ``exclude``: optional uint8 DeviceArray ``(R, C)`` or ``(T, R, C)`` of
	manually excluded pixels (backgrounds.py:96-97).  Returns the background as a float32 DeviceArray ``(T, R, C)`` (and, with
	``return_mask``, the host bool array ``(T, ny, nx)`` of dropped cells and the masked-pixel counts).  A frame without any
	usable cell comes back NaN (the reference returns a NaN image when everything is masked, :99-102).
	"""
	T, R, C = frames.shape
	ny, nx = -(-R // box), -(-C // box)
	mesh = ctx.empty((T, ny, nx), 'float64')
	nmasked = ctx.empty((T, ny, nx), 'int32')
	estride = 0 if exclude is None or len(exclude.shape) == 2 else R * C
	ctx._check(ctx.lib.tp_background_mesh(ctx.handle, frames.ptr, T, R, C, C, R * C, None if exclude is None else exclude.ptr, estride,
		float(flux_cutoff), int(box), mesh.ptr, nmasked.ptr))
	mesh_h, nm_h = mesh.to_host(), nmasked.to_host()
	coef = np.empty((T, ny, nx))
	vmin, vmax = np.empty(T), np.empty(T)
	for k in range(T):
		try:
			m = finish_mesh(mesh_h[k], nm_h[k], box)
		except ValueError:
			m = np.full((ny, nx), np.nan)
		vmin[k], vmax[k] = np.min(m), np.max(m)
		# the cubic-spline coefficients scipy.ndimage.zoom(order=3, mode='reflect') interpolates from
		coef[k] = ndimage.spline_filter(m, order=3, mode='reflect') if min(ny, nx) > 1 else m
	if out is None:
		out = ctx.empty((T, R, C), 'float32')
	if min(ny, nx) > 1:
		d_coef, d_vmin, d_vmax = ctx.array(coef), ctx.array(vmin), ctx.array(vmax)   # kept alive until the kernel has run
		ctx._check(ctx.lib.tp_background_zoom(ctx.handle, d_coef.ptr, d_vmin.ptr, d_vmax.ptr, T, ny, nx, int(box), R, C, C, R * C, out.ptr))
		ctx.sync()
	else:
		host = np.empty((T, R, C), dtype='float32')
		host[:] = coef[:, :1, :1]
		ctx._check(ctx.lib.tp_memcpy_h2d(ctx.handle, out.ptr, host.ctypes.data, host.nbytes))
	if return_mask:
		return out, mesh_h, nm_h
	return out


def prepare_frames(ctx, raw, raw_err, quality, cadence=1800, flux_cutoff=8e4, pixel_flags=None, backapp=False):
	"""
	The image arithmetic of ``prepare_photometry`` for one CCD (prepare.py:265-459) on device-resident stacks ``(T, R, C)``:
	backgrounds (B1), their smoothing over ``time_smooth`` frames (B2, :258, :317-335), ``images = raw - background`` with
	manually excluded pixels set to NaN in image and error (B3, :419-425), the sum image over the good-quality frames
	(A1, :450-453, :459).  Returns a dict of DeviceArrays: ``backgrounds, images, images_err`` float32 ``(T, R, C)``,
	``sumimage`` float64 ``(R, C)``.
	"""
	T, R, C = raw.shape
	time_smooth = {1800: 3, 600: 9}[int(cadence)]
	bkg_us = fit_background_frames(ctx, raw, flux_cutoff=flux_cutoff)
	bkg = ctx.empty((T, R, C), 'float32')
	ctx._check(ctx.lib.tp_frames_smooth_time(ctx.handle, T, R * C, R * C, time_smooth, bkg_us.ptr, bkg.ptr))
	bkg_us.free()
	images = ctx.empty((T, R, C), 'float32')
	images_err = ctx.empty((T, R, C), 'float32')
	sub = bkg
	if backapp: # header BACKAPP: the background is already subtracted (prepare.py:419)
		sub = ctx.zeros((T, R, C), 'float32')
	ctx._check(ctx.lib.tp_frames_subtract(ctx.handle, T * R * C, raw.ptr, raw_err.ptr, sub.ptr, None if pixel_flags is None else pixel_flags.ptr,
		PIXEL_MANUAL_EXCLUDE, images.ptr, images_err.ptr))
	sumimage = ctx.empty((R, C), 'float64')
	q = ctx.array(np.asarray(quality, dtype='int32'))
	ctx._check(ctx.lib.tp_frames_sumimage(ctx.handle, T, R * C, R * C, images.ptr, q.ptr, int(TESS_DEFAULT_BITMASK), sumimage.ptr))
	ctx.sync()
	return {'backgrounds': bkg, 'images': images, 'images_err': images_err, 'sumimage': sumimage}


def background_shenanigans(ctx, images, sumimage, pixel_flags, threshold=40.0, size=15, block=25, indicator=None):
	"""
	The "background shenanigans" pass of the prepare stage (prepare.py:515-622) on device-resident stacks:

	1. per frame the indicator image ``median_filter(img - SumImage, size=15)`` (``pixel_flags.pixel_background_shenanigans``,
	   pixel_flags.py:61-79) -- ``tp_frames_median_filter``;
	2. its robust mean over time: the frames are shuffled with ``np.random.seed(0); np.random.shuffle`` exactly like the
	   reference (:566-568), cut into blocks of 25, the per-pixel nanmedian of every block (NaN -> 0) is averaged over
	   ``ceil(T / 25)`` blocks (:569-577) -- ``tp_frames_block_median_accumulate``;
	3. ``|indicator - mean| > bkgshe_threshold`` sets ``PixelQualityFlags.BackgroundShenanigans`` in the pixel flags of that
	   frame, an old flag is cleared (:594-607) -- ``tp_frames_threshold_flags``.

	``images``: float32 DeviceArray ``(T, R, C)``; ``sumimage``: float64 DeviceArray ``(R, C)``; ``pixel_flags``: uint8 DeviceArray
	``(T, R, C)``, updated in place.  Returns ``(indicator float32 (T, R, C), mean float64 (R, C))`` as DeviceArrays.
	"""
	T, R, C = images.shape
	if indicator is None:
		indicator = ctx.empty((T, R, C), 'float32')
	ctx._check(ctx.lib.tp_frames_median_filter(ctx.handle, images.ptr, T, R, C, C, R * C, None if sumimage is None else sumimage.ptr, int(size),
		indicator.ptr))
	indices = list(range(T))
	np.random.seed(0)              # the reference's own seed and (legacy) generator: the blocks must be the same frames
	np.random.shuffle(indices)
	mean = ctx.zeros((R, C), 'float64')
	keep = []
	for k in range(0, T, block):
		idx = ctx.array(np.asarray(indices[k:k + block], dtype='int32'))
		keep.append(idx)
		ctx._check(ctx.lib.tp_frames_block_median_accumulate(ctx.handle, indicator.ptr, R * C, R * C, idx.ptr, len(indices[k:k + block]), mean.ptr))
	ctx.sync()
	nblocks = int(np.ceil(T / block))
	m = mean.to_host() / nblocks
	mean = ctx.array(m)
	ctx._check(ctx.lib.tp_frames_threshold_flags(ctx.handle, indicator.ptr, mean.ptr, float(threshold), PIXEL_BACKGROUND_SHENANIGANS, R * C, T,
		pixel_flags.ptr))
	ctx.sync()
	return indicator, mean
