# -*- coding: utf-8 -*-
"""
Batched device operations: thin, typed Python wrappers over the C-ABI entry points.
All inputs/outputs are :class:`~photometry_amd.device.DeviceArray` /
:class:`~photometry_amd.device.DeviceCube` objects resident in HBM.
"""

import ctypes
import numpy as np
from .device import DeviceArray, DeviceCube, round_up
from ._lib import tp_cube_desc

#: TESSQualityFlags.DEFAULT_BITMASK (photometry/quality.py:123-124)
TESS_DEFAULT_BITMASK = 1 | 2 | 4 | 8 | 32 | 64 | 128 | 4096


def _ptr(x):
	if x is None:
		return None
	return x.ptr


def sumimage(ctx, images, quality, bitmask=TESS_DEFAULT_BITMASK, out=None, subtract=None):
	"""
	A1 (BasePhotometry.py:1008-1019).  ``images``: DeviceCube; ``quality``: int32 DeviceArray
	``(T,)`` shared or ``(Nt, T)``.  Returns float64 DeviceArray ``(Nt, H, W)``.
	"""
	if out is None:
		out = ctx.empty((images.n_targets, images.height, images.width), 'float64')
	if len(quality.shape) == 1:
		assert quality.shape[0] >= images.n_cad
		qstride = 0
	else:
		assert quality.shape[0] == images.n_targets and quality.shape[1] >= images.n_cad
		qstride = quality.shape[1]
	desc = images.desc
	spitch = 0 if subtract is None else subtract.shape[1]
	ctx._check(ctx.lib.tp_sumimage(ctx.handle, ctypes.byref(desc), images.ptr, quality.ptr, qstride, int(bitmask),
		_ptr(subtract), spitch, out.ptr))
	return out


class LightCurves(object):
	"""Device-resident light-curve block: float64 ``(Nt, T)`` per column."""
	COLUMNS = ('flux', 'flux_err', 'flux_background', 'centroid_col', 'centroid_row')

	def __init__(self, ctx, n_targets, n_cad, block=None):
		self.ctx = ctx
		self.n_targets, self.n_cad = int(n_targets), int(n_cad)
		# one allocation [5][Nt][T] (or a piece of the caller's packed output block) so that a gather moves a single block
		self.block = ctx.zeros((5, self.n_targets, self.n_cad), 'float64') if block is None else block
		stride = self.n_targets * self.n_cad * 8
		self.ptrs = [self.block.ptr + i*stride for i in range(5)]

	def slice0(self, start, count):
		"""Non-owning view of the light curves of targets ``[start, start+count)`` (same block)."""
		v = LightCurves.__new__(LightCurves)
		v.ctx, v.n_targets, v.n_cad, v.block = self.ctx, int(count), self.n_cad, None
		v.ptrs = [p + int(start) * self.n_cad * 8 for p in self.ptrs]
		return v

	def to_host(self):
		b = self.block.to_host()
		lc = {name: b[i] for i, name in enumerate(self.COLUMNS)}
		lc['pos_centroid'] = np.stack((b[3], b[4]), axis=-1) # (Nt, T, 2): column then row (BasePhotometry.py:428)
		return lc


def _background_mode(images, backgrounds):
	"""(bkg_mode, series pitch) of a background argument: a cube, one series per target, or None (aperture-only)."""
	if backgrounds is None:
		return 0, 0
	if isinstance(backgrounds, DeviceCube):
		assert backgrounds.t_pitch == images.t_pitch and backgrounds.data.shape == images.data.shape
		return 0, 0
	assert backgrounds.dtype == np.float32 and backgrounds.shape[0] == images.n_targets
	return 1, backgrounds.shape[1]


def aperture_extract(ctx, images, images_err, backgrounds, mask, stamps, status=None, out=None, subtract=None):
	"""
	A6 (photometry.py:172-201).  ``backgrounds``: DeviceCube, or a float32 DeviceArray ``(Nt, pitch)``
	holding one background series per target (stamp-constant background), or None (aperture-only:
	``flux_background`` comes out NaN).
	``mask``: uint8 DeviceArray ``(Nt, H, W)``; ``stamps``: int32 ``(Nt, 4)``.
	"""
	if out is None:
		out = LightCurves(ctx, images.n_targets, images.n_cad)
	desc = images.desc
	assert images_err.t_pitch == images.t_pitch and images_err.data.shape == images.data.shape
	bkg_mode, bpitch = _background_mode(images, backgrounds)
	assert mask.dtype == np.uint8 and stamps.dtype == np.int32
	ctx._check(ctx.lib.tp_aperture_extract(ctx.handle, ctypes.byref(desc), images.ptr, images_err.ptr, _ptr(backgrounds),
		bkg_mode, bpitch, _ptr(subtract), 0 if subtract is None else subtract.shape[1], mask.ptr, stamps.ptr, _ptr(status), out.ptrs[0], out.ptrs[1], out.ptrs[2], out.ptrs[3], out.ptrs[4],
		out.n_cad))
	return out


def synth_fill(ctx, scene, n_targets=None, target_offset=0, nan_fraction=1e-3, images=True, images_err=True, backgrounds=True, raw=False):
	"""
	Fill device cubes for ``scene`` (``photometry_amd.simulate.make_scene``) with the device RNG.
	Returns dict of DeviceCube (keys ``images, images_err, backgrounds, raw`` as requested).
	"""
	Nt = scene.n_targets if n_targets is None else int(n_targets)
	sl = slice(target_offset, target_offset + Nt)
	T, H, W = scene.n_cad, scene.height, scene.width
	out = {}
	for name, want in (('images', images), ('images_err', images_err), ('backgrounds', backgrounds), ('raw', raw)):
		out[name] = DeviceCube(ctx, Nt, T, H, W) if want else None
	any_cube = next(v for v in out.values() if v is not None)
	desc = any_cube.desc
	sp = ctx.array(scene.star_params[sl], dtype='float64')
	sig = ctx.array(scene.sigma_psf[sl], dtype='float64')
	lev = ctx.array(scene.bkg_level[sl], dtype='float64')
	pha = ctx.array(scene.bkg_phase[sl], dtype='float64')
	jit = ctx.array(scene.jitter, dtype='float64')
	ctx._check(ctx.lib.tp_synth_fill(ctx.handle, ctypes.byref(desc), scene.star_params.shape[1], sp.ptr, sig.ptr, lev.ptr, pha.ptr,
		jit.ptr, float(scene.readnoise), float(nan_fraction), int(scene.seed) * 7919 + int(target_offset) * 104729 + 1,
		_ptr(out['images']), _ptr(out['images_err']), _ptr(out['backgrounds']), _ptr(out['raw'])))
	ctx.sync()
	for a in (sp, sig, lev, pha, jit):
		a.free()
	return {k: v for k, v in out.items() if v is not None}


def background_stamp(ctx, raw, flux_cutoff=8e4, exclude_percentile=50.0, out=None):
	"""B* (build-defined stamp analogue of backgrounds.py:52-211).  Returns float32 DeviceArray ``(Nt, t_pitch)``."""
	if out is None:
		out = ctx.zeros((raw.n_targets, raw.t_pitch), 'float32')
	desc = raw.desc
	ctx._check(ctx.lib.tp_background_stamp(ctx.handle, ctypes.byref(desc), raw.ptr, float(flux_cutoff), float(exclude_percentile),
		out.ptr, out.shape[1]))
	return out


def _quality_stride(quality, images):
	if len(quality.shape) == 1:
		assert quality.shape[0] >= images.n_cad
		return 0
	assert quality.shape[0] == images.n_targets and quality.shape[1] >= images.n_cad
	return quality.shape[1]


def background_sumimage(ctx, raw, quality, time_smooth=3, bitmask=TESS_DEFAULT_BITMASK, flux_cutoff=8e4, exclude_percentile=50.0,
	bkg_raw=None, bkg=None, sumimage=None):
	"""
	B* + B2 + A1 in one pass over the raw cube (``tp_background_sumimage``): the unsmoothed and the smoothed background series
	(float32 ``(Nt, t_pitch)``) and the sum image of ``raw - smoothed background`` (float64 ``(Nt, H, W)``).
	"""
	if bkg_raw is None:
		bkg_raw = ctx.zeros((raw.n_targets, raw.t_pitch), 'float32')
	if bkg is None:
		bkg = ctx.zeros((raw.n_targets, raw.t_pitch), 'float32')
	if sumimage is None:
		sumimage = ctx.empty((raw.n_targets, raw.height, raw.width), 'float64')
	assert bkg_raw.shape[1] == bkg.shape[1]
	desc = raw.desc
	ctx._check(ctx.lib.tp_background_sumimage(ctx.handle, ctypes.byref(desc), raw.ptr, float(flux_cutoff), float(exclude_percentile), int(time_smooth),
		quality.ptr, _quality_stride(quality, raw), int(bitmask), bkg_raw.ptr, bkg.ptr, bkg.shape[1], sumimage.ptr))
	return bkg_raw, bkg, sumimage


def smooth_time(ctx, series, n_cad, time_smooth=3, out=None):
	"""B2 (prepare.py:317-335) on float32 series ``(Nt, pitch)``."""
	if out is None:
		out = ctx.zeros(series.shape, 'float32')
	ctx._check(ctx.lib.tp_smooth_time(ctx.handle, series.shape[0], int(n_cad), series.shape[1], int(time_smooth), series.ptr, out.ptr))
	return out


def subtract_background(ctx, raw, bkg_series, raw_err=None, pixel_flags=None, flag_mask=2, images=None, images_err=None):
	"""B3 (prepare.py:419-425).  ``images`` defaults to in-place on ``raw``."""
	images = raw if images is None else images
	if raw_err is not None and images_err is None:
		images_err = raw_err
	desc = raw.desc
	ctx._check(ctx.lib.tp_subtract_background(ctx.handle, ctypes.byref(desc), raw.ptr, _ptr(raw_err), bkg_series.ptr, bkg_series.shape[1],
		_ptr(pixel_flags), int(flag_mask), images.ptr, _ptr(images_err)))
	return images, images_err


def k2p2_masks(ctx, batch, work, cut_override=None, params=None):
	"""
	A2..A5b + A7 on the device (k2p2v2.py:344-623, photometry.py:93-131, 220-254).
	``batch``: :class:`photometry_amd.pipeline.ApertureBatch`; ``work``: ``ApertureWork`` (needs ``sumimage``).
	"""
	ctx._check(ctx.lib.tp_k2p2_masks(ctx.handle, batch.n_targets, batch.height, batch.width, work.sumimage.ptr,
		batch.cat_offsets.ptr, batch.cat_column_stamp.ptr, batch.cat_row_stamp.ptr, batch.cat_tmag.ptr,
		batch.cat_column.ptr, batch.cat_row.ptr, batch.cat_starid.ptr,
		batch.target_pos_row.ptr, batch.target_pos_column.ptr, batch.target_tmag.ptr, batch.target_starid.ptr,
		batch.stamps.ptr, batch.aperture.ptr, _ptr(cut_override), None if params is None else ctypes.byref(params),
		work.mask.ptr, work.status.ptr, work.flags.ptr, work.contamination.ptr, work.diag.ptr, work.cat_in_mask.ptr))
	return work


def aperture_photometry(ctx, batch, work, bitmask=TESS_DEFAULT_BITMASK, subtract=None, backgrounds=None, params=None, sumimage_given=False):
	"""
	A1 + A2..A5b + A7 + A6 in one launch (photometry.py:75-257 for every target of the batch): fills
	``work.sumimage, mask, status, flags, contamination, diag, cat_in_mask, lc``.
	``backgrounds`` / ``subtract`` as in :func:`aperture_extract` (default: ``batch.backgrounds``).
	``sumimage_given``: ``work.sumimage`` is an input (:func:`background_sumimage` formed it) and the launch starts at the mask.
	"""
	images, images_err = batch.images, batch.images_err
	backgrounds = batch.backgrounds if backgrounds is None else backgrounds
	desc = images.desc
	assert images_err.t_pitch == images.t_pitch and images_err.data.shape == images.data.shape
	bkg_mode, bpitch = _background_mode(images, backgrounds)
	quality = batch.quality
	qstride = _quality_stride(quality, images)
	lc = work.lc
	entry = ctx.lib.tp_aperture_photometry_from_sumimage if sumimage_given else ctx.lib.tp_aperture_photometry
	ctx._check(entry(ctx.handle, ctypes.byref(desc), images.ptr, images_err.ptr, _ptr(backgrounds), bkg_mode, bpitch,
		_ptr(subtract), 0 if subtract is None else subtract.shape[1],
		quality.ptr, qstride, int(bitmask),
		batch.cat_offsets.ptr, batch.cat_column_stamp.ptr, batch.cat_row_stamp.ptr, batch.cat_tmag.ptr,
		batch.cat_column.ptr, batch.cat_row.ptr, batch.cat_starid.ptr,
		batch.target_pos_row.ptr, batch.target_pos_column.ptr, batch.target_tmag.ptr, batch.target_starid.ptr,
		batch.stamps.ptr, batch.aperture.ptr, None if params is None else ctypes.byref(params),
		work.sumimage.ptr, work.mask.ptr, work.status.ptr, work.flags.ptr, work.contamination.ptr, work.diag.ptr, work.cat_in_mask.ptr,
		lc.ptrs[0], lc.ptrs[1], lc.ptrs[2], lc.ptrs[3], lc.ptrs[4], lc.n_cad))
	return work


def cut_stamps(ctx, frames, stamps, height, width, row_offset=0, col_offset=0, out=None):
	"""
	Stamp cutter (BasePhotometry._load_cube, BasePhotometry.py:720-742, for a batch).  ``frames``: float32 DeviceArray
	``(T, R, C)`` (one HDF5 image group of a CCD resident in HBM); ``stamps``: int32 DeviceArray ``(Nt, 4)`` in CCD
	coordinates, all ``height x width``.  Returns a :class:`DeviceCube`.
	"""
	T, R, C = frames.shape
	Nt = stamps.shape[0]
	if out is None:
		out = DeviceCube(ctx, Nt, T, height, width)   # (not cleared: the cutter writes the padding of the time axis as zeros itself)
	desc = out.desc
	ctx._check(ctx.lib.tp_cut_stamps(ctx.handle, frames.ptr, T, R, C, C, R * C, int(row_offset), int(col_offset), stamps.ptr,
		ctypes.byref(desc), out.ptr))
	return out


def cut_stamps_multi(ctx, frames_list, stamps, height, width, row_offset=0, col_offset=0):
	"""
	:func:`cut_stamps` for several frame stacks of one geometry at once (``tp_cut_stamps_multi``: the image groups of a CCD share their
	stamps, so the stamps are binned into frame tiles once and one launch cuts all stacks).  Returns a list of :class:`DeviceCube`.
	"""
	T, R, C = frames_list[0].shape
	assert all(tuple(f.shape) == (T, R, C) for f in frames_list) and 1 <= len(frames_list) <= 4
	Nt = stamps.shape[0]
	outs = [DeviceCube(ctx, Nt, T, height, width) for _ in frames_list]
	desc = outs[0].desc
	fp = (ctypes.c_void_p * len(frames_list))(*[f.ptr for f in frames_list])
	cp = (ctypes.c_void_p * len(outs))(*[o.ptr for o in outs])
	ctx._check(ctx.lib.tp_cut_stamps_multi(ctx.handle, len(frames_list), fp, T, R, C, C, R * C, int(row_offset), int(col_offset), stamps.ptr,
		ctypes.byref(desc), cp))
	return outs


#: columns of the diagnostics block (BasePhotometry.py:1357-1403)
DIAGNOSTICS_COLUMNS = ('mean_flux', 'variance', 'rms_hour', 'ptp', 'pos_centroid_col', 'pos_centroid_row', 'variability',
	'mask_size', 'edge_flux', 'flags')


def lightcurve_diagnostics(ctx, lc, time, quality, status=None, sumimage=None, mask=None, bitmask=TESS_DEFAULT_BITMASK,
	timescale=3600/86400, out=None):
	"""
	Light-curve diagnostics of a batch (BasePhotometry.py:1343-1407, utilities.py:227-264).
	``lc``: :class:`LightCurves`; ``time``: float64 DeviceArray ``(T,)``; ``quality`` as in :func:`sumimage`.
	Returns float64 DeviceArray ``(Nt, 10)`` with the columns :data:`DIAGNOSTICS_COLUMNS`.
	"""
	Nt, T = lc.n_targets, lc.n_cad
	if out is None:
		out = ctx.empty((Nt, 10), 'float64')
	qstride = 0 if len(quality.shape) == 1 else quality.shape[1]
	H = W = 0
	if mask is not None:
		H, W = int(mask.shape[1]), int(mask.shape[2])
	ctx._check(ctx.lib.tp_lightcurve_diagnostics(ctx.handle, Nt, T, lc.ptrs[0], lc.ptrs[1], lc.ptrs[3], lc.ptrs[4], lc.n_cad,
		time.ptr, quality.ptr, qstride, int(bitmask), _ptr(status), _ptr(sumimage), _ptr(mask), H, W, float(timescale), out.ptr))
	return out


class LinPSFResult(object):
	"""Device-resident outputs of the LinPSF pipeline."""
	def __init__(self, ctx, n_targets, n_fit_stars, n_cad, flux=None, contamination=None, status=None):
		"""``flux`` / ``contamination`` / ``status``: optional caller-owned arrays (pieces of a packed output block)."""
		self.n_cad = int(n_cad)
		self.flux = ctx.zeros((n_targets, n_cad), 'float64') if flux is None else flux
		self.flux_err = ctx.zeros((n_targets, n_cad), 'float64')
		self.fluxes_all = ctx.zeros((max(n_fit_stars, 1), n_cad), 'float64')
		self.contamination = ctx.zeros((n_targets,), 'float64') if contamination is None else contamination
		self.status = ctx.zeros((n_targets,), 'int32') if status is None else status
		self.fluxes_mean = ctx.zeros((max(n_fit_stars, 1),), 'float64')

	def to_host(self, keys=('flux', 'flux_err', 'fluxes_all', 'contamination', 'status', 'fluxes_mean')):
		"""The arrays on the host (all of them by default; ``keys`` picks: the per-star fluxes are the largest piece and few callers want them)."""
		return {k: getattr(self, k).to_host() for k in keys}


def linpsf_prf(ctx, base_coef, weights, out=None):
	"""P1 (psf.py:101-119): per-target spline coefficient tables ``(Nt, n*n)`` from the sample tables."""
	n_samples, n_coef = base_coef.shape
	Nt = weights.shape[0]
	assert weights.shape[1] == n_samples
	if out is None:
		out = ctx.empty((Nt, n_coef), 'float64')
	ctx._check(ctx.lib.tp_linpsf_prf(ctx.handle, Nt, n_samples, n_coef, base_coef.ptr, weights.ptr, out.ptr))
	return out


def linpsf_set_path(ctx, path):
	"""``tp_linpsf_set_path``: 1 = matrix-core fit where a target qualifies (default), 0 = vector-ALU fit kernels for every target."""
	ctx._check(ctx.lib.tp_linpsf_set_path(ctx.handle, int(path)))


def crop_sumimage(ctx, full, stamps, height, width, row0, col0, out=None):
	"""The sum images of a group of stamps as crops of the region's (BasePhotometry.py:1001-1006, ``tp_crop_sumimage``): ``full`` float64
	DeviceArray ``(R, C)`` covering the CCD from ``(row0, col0)``, ``stamps`` int32 DeviceArray ``(n, 4)``; float64 ``(n, height * width)``."""
	n = int(stamps.shape[0])
	if out is None:
		out = ctx.empty((n, height * width), 'float64')
	R, C = full.shape
	ctx._check(ctx.lib.tp_crop_sumimage(ctx.handle, full.ptr, R, C, C, int(row0), int(col0), stamps.ptr, n, int(height), int(width), out.ptr))
	return out


def linpsf_last_counts(ctx):
	"""Which kernels fitted the targets of the last :func:`linpsf_fit` call (``tp_linpsf_last_counts``), as a dict."""
	c = (ctypes.c_int64 * 14)()
	ctx._check(ctx.lib.tp_linpsf_last_counts(ctx.handle, c, 14))
	return {'matrix_core_targets': int(c[0]), 'matrix_core_segments': int(c[1]), 'vector_alu_polynomial_targets': int(c[2]),
		'vector_alu_general_targets': int(c[3]), 'many_star_targets': int(c[4]),
		'matrix_core_targets_by_stars': [int(c[5 + i]) for i in range(4)], 'matrix_core_segments_by_stars': [int(c[9 + i]) for i in range(4)],
		'any_grid_targets': int(c[13])}


def star_positions(ctx, base, shift):
	"""Positions of ``n_stars`` stars at ``T`` cadences for a field that moves as a whole: ``float64(base[s] + shift[k])`` with the sum
	in float32 (what ``catalog_attime`` leaves in the plugin's catalogue for a translation), on the device (``tp_star_positions``).
	``base``, ``shift``: float32 DeviceArrays; returns a float64 DeviceArray ``(n_stars, T)``."""
	n, T = int(base.shape[0]), int(shift.shape[0])
	out = ctx.empty((max(n, 1), T), 'float64')
	ctx._check(ctx.lib.tp_star_positions(ctx.handle, n, T, base.ptr, shift.ptr, out.ptr, T))
	return out


def linpsf_fit(ctx, images, coef, knots_x, knots_y, star_offsets, target_index, pos_row, pos_col, max_stars,
	cutoff_radius=5.0, subtract=None, out=None):
	"""P2-P4 (psf.py:122-148, linpsf_photometry.py:22-34, 79-219).  ``cutoff_radius=None``: no cut-off (psf.py:142)."""
	n, ny = knots_x.shape[0] - 4, knots_y.shape[0] - 4      # (axes of different lengths: tp_linpsf_fit_xy, the any-grid kernels)
	if cutoff_radius is None:
		cutoff_radius = float('inf')
	if out is None:
		out = LinPSFResult(ctx, images.n_targets, pos_row.shape[0], images.n_cad)
	assert pos_row.shape[1] >= images.n_cad and pos_col.shape == pos_row.shape
	desc = images.desc
	ctx._check(ctx.lib.tp_linpsf_fit_xy(ctx.handle, ctypes.byref(desc), images.ptr, _ptr(subtract), 0 if subtract is None else subtract.shape[1],
		coef.ptr, knots_x.ptr, knots_y.ptr, n, ny, int(max_stars), star_offsets.ptr, target_index.ptr,
		pos_row.ptr, pos_col.ptr, pos_row.shape[1], float(cutoff_radius),
		out.flux.ptr, out.flux_err.ptr, out.fluxes_all.ptr, out.n_cad, out.contamination.ptr, out.status.ptr, out.fluxes_mean.ptr))
	return out



def psf_fit(ctx, images, backgrounds, coef, knots_x, knots_y, star_offsets, params0, mini_aperture, variance_floor=9.0,
	cutoff_radius=5.0, maxiter_first=1500, maxiter=500):
	"""
	Non-linear PSF photometry of a batch (psf_photometry.py:111-196, ``tp_psf_fit``).  Returns a dict of DeviceArrays:
	``flux, flux_err, centroid_row, centroid_col`` float64 ``(Nt, T)``, ``params`` float64 ``(n_fit * 3, T)``, ``nit`` int32
	``(Nt, T)``, ``status`` int32 ``(Nt,)``.
	"""
	Nt, T = images.n_targets, images.n_cad
	n, ny = knots_x.shape[0] - 4, knots_y.shape[0] - 4
	if cutoff_radius is None:      # psf.py:142: no cut-off
		cutoff_radius = float('inf')
	out = {k: ctx.zeros((Nt, T), 'float64') for k in ('flux', 'flux_err', 'centroid_row', 'centroid_col')}
	out['params'] = ctx.zeros((max(params0.shape[0], 1) * 3, T), 'float64')
	out['nit'] = ctx.zeros((Nt, T), 'int32')
	out['status'] = ctx.zeros((Nt,), 'int32')
	desc = images.desc
	ctx._check(ctx.lib.tp_psf_fit_xy(ctx.handle, ctypes.byref(desc), images.ptr, _ptr(backgrounds), coef.ptr, knots_x.ptr, knots_y.ptr, n, ny,
		star_offsets.ptr, params0.ptr, mini_aperture.ptr, float(variance_floor), float(cutoff_radius), int(maxiter_first), int(maxiter),
		out['flux'].ptr, out['flux_err'].ptr, out['centroid_row'].ptr, out['centroid_col'].ptr, T, out['params'].ptr, out['nit'].ptr,
		out['status'].ptr))
	return out
