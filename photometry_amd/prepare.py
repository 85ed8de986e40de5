# -*- coding: utf-8 -*-
"""
The arithmetic of the reference's prepare stage (photometry/prepare.py:265-459, photometry/backgrounds.py:52-211) on a frame
stack resident in HBM: background estimation of every full-frame image, its smoothing in time, the subtraction with the
manual-exclude masking, the pixel flags and the sum image -- everything ``BasePhotometry`` later reads per target.

What runs where: every pixel-sized operation is a device kernel (``csrc/fullframe.hip``); the background mesh of a frame
(32 x 32 numbers for a 2048 x 2048 image) goes through its few small steps -- drop the mostly-masked cells, fill them from
their neighbours, 3 x 3 median filter, cubic-spline prefilter -- on the host with the same scipy calls photutils makes.
``fit_background_frames`` is both branches of the reference's ``fit_background``: without ``camera`` / ``ccd`` the one it takes
for a plain image (``bkgiters = 1``, backgrounds.py:156-157 -- what its own test exercises, tests/test_background.py:36-54),
with them the TESS one: ``bkgiters`` rounds of the radial corner-glow component (ring modes on the device,
``csrc/radial.hip``; the 3-point median and the interpolating spline of ~40 ring values on the host with the reference's own
scipy call) and the mesh of ``img0 - img_bkg_radial``.  HDF5 / FITS I/O and WCS are not part of this module.
"""

import ctypes
import logging
import warnings
import numpy as np
from scipy.interpolate import InterpolatedUnivariateSpline
from . import _lib
from .engine import TESS_DEFAULT_BITMASK

#: PixelQualityFlags (photometry/quality.py:157-166)
PIXEL_NOT_USED_FOR_BACKGROUND = 1
PIXEL_MANUAL_EXCLUDE = 2
PIXEL_BACKGROUND_SHENANIGANS = 4


def finish_mesh(mesh, nmasked, box=64, exclude_percentile=50.0, filter_size=3):
	"""
	(Host statement of what ``tp_background_mesh_finish`` does on the device -- kept for the tests that hold the kernel to numpy;
	``fit_background_frames`` does not call it.)

	The low-resolution part of photutils ``Background2D`` (1.3.0) after the per-cell statistics, for one frame ``(ny, nx)`` or
	a batch ``(T, ny, nx)``: cells with more than ``exclude_percentile`` % masked pixels are replaced by the inverse-distance
	weighted mean of the 10 nearest kept cells, then the 3 x 3 median filter.  Returns the filtered mesh (float64).  A single
	frame without any usable cell raises ``ValueError`` like photutils; in a batch such a frame comes back NaN.
	"""
	mesh = np.array(mesh, dtype='float64', copy=True)
	single = (mesh.ndim == 2)
	if single:
		mesh = mesh[None]
		nmasked = np.asarray(nmasked)[None]
	T, ny, nx = mesh.shape
	keep = (np.asarray(nmasked) <= exclude_percentile / 100.0 * box * box) & np.isfinite(mesh)
	for k in np.nonzero(~keep.all(axis=(1, 2)))[0]:
		if not keep[k].any():
			if single:
				raise ValueError(f"All meshes contain > {exclude_percentile} percent masked pixels")
			mesh[k] = np.nan
			continue
		ky, kx = np.nonzero(keep[k])
		kv = mesh[k][keep[k]]
		for y, x in zip(*np.nonzero(~keep[k])):
			dist = np.hypot(ky - y, kx - x)
			nearest = np.argsort(dist, kind='stable')[:10]
			w = 1.0 / dist[nearest]
			mesh[k, y, x] = np.sum(w * kv[nearest]) / np.sum(w)
	if filter_size > 1:
		# generic_filter(mesh, nanmedian, size, mode='constant', cval=nan) as photutils calls it: the NaN-ignoring median of
		# every window, windows running off the mesh padded with NaN -- as one nanmedian over the stack of shifted copies
		h = filter_size // 2
		pad = np.pad(mesh, ((0, 0), (h, filter_size - 1 - h), (h, filter_size - 1 - h)), constant_values=np.nan)
		stack = np.stack([pad[:, dy:dy + ny, dx:dx + nx] for dy in range(filter_size) for dx in range(filter_size)])
		with warnings.catch_warnings():
			warnings.simplefilter('ignore', RuntimeWarning)     # a window without any finite value is NaN, as there
			mesh = np.nanmedian(stack, axis=0)
	return mesh[0] if single else mesh


#: pixel coordinates of the camera centre relative to every CCD (backgrounds.py:118-135)
CAMERA_CENTRE = {
	(1, 1): (2158.222313, 2099.523364), (1, 2): (-5.653058, 2098.018608), (1, 3): (2141.511437, 2099.868226), (1, 4): (-22.406442, 2100.116443),
	(2, 1): (2148.588316, 2094.033024), (2, 2): (-16.806140, 2095.810070), (2, 3): (2151.351646, 2105.747100), (2, 4): (-13.118570, 2105.982211),
	(3, 1): (2152.175481, 2092.337442), (3, 2): (-10.494413, 2093.108135), (3, 3): (2145.029218, 2107.883573), (3, 4): (-17.374782, 2105.296746),
	(4, 1): (2149.259760, 2091.433315), (4, 2): (-12.906931, 2093.350054), (4, 3): (2148.906766, 2110.730620), (4, 4): (-14.629676, 2111.341670),
}
#: first science column of a TESS FFI: the distance image is in the coordinates of the full 2136-column frame (backgrounds.py:144)
TESS_SCIENCE_COLUMN = 44


class RadialGeometry(object):
	"""
	What the radial component needs to know about a CCD and never changes between frames (backgrounds.py:110-154): the ring
	edges, their centres, and for every ring the list of its pixels in row-major order (the order ``r[~mask]`` hands them to
	``binned_statistic``: samples on an edge go to the ring on its right, samples within rounding of the LAST edge into the last
	ring -- scipy 1.7.3 ``_bin_numbers``).
	"""
	def __init__(self, shape, camera, ccd, radial_cutoff=2400, radial_pixel_step=15):
		if (camera, ccd) not in CAMERA_CENTRE:
			raise ValueError(f"Invalid CAMERA or CCD in header: CAMERA={camera}, CCD={ccd}")
		self.xcen, self.ycen = CAMERA_CENTRE[(camera, ccd)]
		R, C = shape
		xx, yy = np.meshgrid(np.arange(TESS_SCIENCE_COLUMN, C + TESS_SCIENCE_COLUMN, 1), np.arange(0, R, 1))
		r = np.sqrt((xx - self.xcen)**2 + (yy - self.ycen)**2)
		self.bins = np.arange(radial_cutoff, np.max(r) + radial_pixel_step, radial_pixel_step)
		if len(self.bins) < 2:
			raise ValueError("no pixel beyond the radial cutoff: the radial component has no rings")
		self.bin_center = self.bins[1:] - radial_pixel_step / 2
		self.n_rings = len(self.bins) - 1
		r = r.ravel()
		number = np.digitize(r, self.bins)
		decimal = int(-np.log10(np.diff(self.bins).min())) + 6
		number[np.around(r, decimal) == np.around(self.bins[-1], decimal)] -= 1
		inside = np.nonzero((number >= 1) & (number <= self.n_rings))[0]
		order = np.argsort(number[inside], kind='stable')
		self.ring_pixels = inside[order].astype('int32')
		self.ring_offsets = np.concatenate(([0], np.cumsum(np.bincount(number[inside] - 1, minlength=self.n_rings)))).astype('int32')


def _move_median_central(x, width):
	"""utilities.move_median_central (utilities.py:52-62) on a short 1-D array: NaN-ignoring centred moving median with the
	ends redone over the first / last ``k + 2`` points."""
	n = len(x)
	y = np.full(n, np.nan)
	def med(v):
		v = v[~np.isnan(v)]
		return np.median(v) if len(v) else np.nan
	trail = np.array([med(x[max(i - width + 1, 0):i + 1]) for i in range(n)])
	y = np.roll(trail, -width // 2 + 1)
	for k in range(width // 2 + 1):
		y[k] = med(x[:k + 2])
		y[-(k + 1)] = med(x[-(k + 2):])
	return y


def radial_profiles(s2, bin_center, radial_smooth=3, max_knots=None):
	"""
	backgrounds.py:178-195 for the ring modes ``s2 (T, n_rings)`` of a batch of frames: smoothing, then the interpolating cubic
	spline through the finite values (``InterpolatedUnivariateSpline(k=3, ext=3)``: the reference's own scipy call) in FITPACK
	form.  Returns ``(knots (T, K), coefs (T, K), n_knots int32 (T,))``; ``n_knots == 0`` where the reference falls back to no
	radial component (fewer than 3 points, or a ``ValueError`` from the fit).
	"""
	logger = logging.getLogger(__name__)
	T, nr = s2.shape
	K = max(nr + 4, 8) if max_knots is None else int(max_knots)
	knots, coefs, n_knots = np.zeros((T, K)), np.zeros((T, K)), np.zeros(T, dtype='int32')
	for k in range(T):
		prof = _move_median_central(s2[k], radial_smooth) if radial_smooth else s2[k]
		good = ~np.isnan(prof)
		ngood = int(np.sum(good))
		if ngood < 3: # "The required number of points for qubic spline" (:183)
			logger.warning("Not enough points for radial interpolation (N=%d).", ngood)
			continue
		try:
			t, c, _ = InterpolatedUnivariateSpline(bin_center[good], prof[good], k=3, ext=3)._eval_args
		except ValueError:
			logger.exception("Background interpolation failed (N=%d).", ngood)
			continue
		n_knots[k] = len(t)
		knots[k, :len(t)] = t
		coefs[k, :len(c)] = c
	return knots, coefs, n_knots


def _square_component(ctx, frames, flux_cutoff, box, exclude, estride, subtract, out, want_host=False, work=None, radial_spec=None):
	"""
	Background2D of ``frames - subtract`` (backgrounds.py:199-206), everything on the device: per-cell statistics
	(``tp_background_mesh``), the low-resolution finishing -- excluded cells filled, 3 x 3 median filter, spline prefilter
	(``tp_background_mesh_finish``) -- and the cubic-spline zoom (``tp_background_zoom``); no host round trip.  ``want_host``:
	also return the host copies of the cell statistics and masked-pixel counts.  ``work``: dict that keeps the small device
	arrays between calls of one ``fit_background_frames``.

	``radial_spec`` (a ``tp_radial_image``) in place of ``subtract``: the radial component is evaluated inside the mesh kernel from
	its ring profile instead of read from a stored image; ``out`` None: the zoomed mesh is not written (the caller evaluates it
	where it needs it from ``work['coef' / 'vmin' / 'vmax']``: ``zoom_image``).
	"""
	T, R, C = frames.shape
	ny, nx = -(-R // box), -(-C // box)
	work = {} if work is None else work
	if 'mesh' not in work:
		work.update(mesh=ctx.empty((T, ny, nx), 'float64'), nmasked=ctx.empty((T, ny, nx), 'int32'), coef=ctx.empty((T, ny, nx), 'float64'),
			vmin=ctx.empty((T,), 'float64'), vmax=ctx.empty((T,), 'float64'))
	mesh, nmasked, coef, vmin, vmax = (work[k] for k in ('mesh', 'nmasked', 'coef', 'vmin', 'vmax'))
	if ny * nx > 8192:   # (the finishing kernel's work arrays live in LDS: 2048 x 2048 in 64-pixel boxes is 1024 cells, 4096 x 4096 or 32-pixel boxes 4096)
		raise ValueError(f'fit_background_frames: a {R} x {C} frame has {ny} x {nx} cells of {box} pixels; the device finishes meshes of at most 8192 cells')
	if radial_spec is not None:
		ctx._check(ctx.lib.tp_background_mesh_radial(ctx.handle, frames.ptr, T, R, C, C, R * C, None if exclude is None else exclude.ptr, estride,
			ctypes.byref(radial_spec), float(flux_cutoff), int(box), mesh.ptr, nmasked.ptr))
	else:
		ctx._check(ctx.lib.tp_background_mesh(ctx.handle, frames.ptr, T, R, C, C, R * C, None if exclude is None else exclude.ptr, estride,
			None if subtract is None else subtract.ptr, R * C, float(flux_cutoff), int(box), mesh.ptr, nmasked.ptr))
	ctx._check(ctx.lib.tp_background_mesh_finish(ctx.handle, mesh.ptr, nmasked.ptr, T, ny, nx, int(box), 50.0, 3, coef.ptr, vmin.ptr, vmax.ptr, None))
	if out is not None:
		ctx._check(ctx.lib.tp_background_zoom(ctx.handle, coef.ptr, vmin.ptr, vmax.ptr, T, ny, nx, int(box), R, C, C, R * C, out.ptr))
	if want_host:
		return mesh.to_host(), nmasked.to_host()
	return None, None


def fit_background_frames(ctx, frames, flux_cutoff=8e4, box=64, exclude=None, out=None, return_mask=False,
	camera=None, ccd=None, bkgiters=3, radial_cutoff=2400, radial_pixel_step=15, radial_smooth=3, geometry=None, details=None, implicit=True):
	"""
	``fit_background`` (backgrounds.py:52-211) for every frame of a stack.

	``frames``: float32 DeviceArray ``(T, R, C)``; ``exclude``: optional uint8 DeviceArray ``(R, C)`` or ``(T, R, C)`` of
	manually excluded pixels (backgrounds.py:96-97).  Returns the background as a float32 DeviceArray ``(T, R, C)`` (and, with
	``return_mask``, the host arrays of the last mesh and its masked-pixel counts).  A frame without any usable cell comes
	back NaN (the reference returns a NaN image when everything is masked, :99-102).

	With ``camera`` and ``ccd`` (the FFI header cards of a TESS image, :116-117) the frames are the 2048 science columns of TESS
	full-frame images and the radial component is fitted as well, ``bkgiters`` times alternating with the mesh (:162-206); a
	``RadialGeometry`` can be passed in ``geometry`` to reuse it between calls.  ``details``: optional dict that receives the
	ring modes, zero points and spline arrays of every round (host arrays).

	``implicit`` (default): inside the alternation neither the radial nor the square component is ever stored as an image -- the
	mesh kernel evaluates the radial component from its ring profile, the zero-point and ring passes evaluate the zoomed mesh from
	its spline coefficients, the last pass writes the total: per iteration the frame is read by the zero-point pass, the ring
	pass (a tenth of it) and the mesh pass, and nothing the size of a frame is written (round 4: 304 MB of traffic per 2048 x
	2048 frame, now the three reads per iteration and one write).  ``implicit=False`` stores both images like round 4 (bit-identical;
	kept for the test that says so).
	"""
	T, R, C = frames.shape
	estride = 0 if exclude is None or len(exclude.shape) == 2 else R * C
	if out is None:
		out = ctx.empty((T, R, C), 'float32')
	work = {}
	if camera is None and ccd is None and geometry is None:
		mesh_h, nm_h = _square_component(ctx, frames, flux_cutoff, box, exclude, estride, None, out, want_host=return_mask, work=work)
		ctx.sync()   # the small work arrays go out of scope with this call
		return (out, mesh_h, nm_h) if return_mask else out

	geo = geometry if geometry is not None else RadialGeometry((R, C), camera, ccd, radial_cutoff, radial_pixel_step)
	d_pixels, d_offsets = ctx.array(geo.ring_pixels), ctx.array(geo.ring_offsets)
	n_ring_pixels = int(geo.ring_offsets[-1])
	n_partial = 256
	d_partial = ctx.empty((T, n_partial), 'float64')
	d_zp = ctx.empty((T,), 'float64')
	d_scratch = ctx.empty((T, max(n_ring_pixels, 1)), 'float64')
	d_modes = ctx.empty((T, geo.n_rings), 'float64')
	d_counts = ctx.empty((T, geo.n_rings), 'int32')
	radial = square = None
	# kernels.Gaussian().normal_reference_constant of statsmodels (order 2, L2 norm 1 / (2 sqrt(pi)), unit variance)
	bw_constant = np.pi**0.5 * 2.0**3 * (1.0 / (2.0 * np.sqrt(np.pi)))
	bw_constant /= (2 * 2 * 24.0 * 1.0**2)
	bw_constant = 2 * bw_constant**(1.0 / 5)
	ex_ptr = None if exclude is None else exclude.ptr
	# the ring profile (3-point median, interpolating spline through the rings that have a mode) on the device when it fits the
	# kernel (<= 64 rings: a 2048 x 2048 CCD has 39); otherwise on the host with the reference's own scipy call
	on_device = geo.n_rings <= 64 and (radial_smooth or 0) <= 8
	K = max(geo.n_rings + 4, 8)
	implicit = bool(implicit) and K <= 72      # (what the mesh kernel stages in LDS: tp_background_mesh_radial)
	d_bin_center = ctx.array(np.asarray(geo.bin_center, dtype='float64'))
	d_knots, d_coefs, d_nk = ctx.zeros((T, K), 'float64'), ctx.zeros((T, K), 'float64'), ctx.zeros((T,), 'int32')
	ny, nx = -(-R // box), -(-C // box)
	radial_spec = _lib.tp_radial_image(float(TESS_SCIENCE_COLUMN), float(geo.xcen), float(geo.ycen), d_knots.ptr, d_coefs.ptr, d_nk.ptr, d_zp.ptr, K, 0)
	zoom_spec = None
	mesh_h = nm_h = None
	for it in range(int(bkgiters)):
		last = it == int(bkgiters) - 1
		if implicit and zoom_spec is not None:
			ctx._check(ctx.lib.tp_radial_zeropoint_zoom(ctx.handle, frames.ptr, T, R * C, R * C, ctypes.byref(zoom_spec), ex_ptr, estride, float(flux_cutoff),
				d_partial.ptr, n_partial, d_zp.ptr))
			ctx._check(ctx.lib.tp_radial_ring_modes_zoom(ctx.handle, frames.ptr, T, R * C, R * C, ctypes.byref(zoom_spec), ex_ptr, estride, float(flux_cutoff),
				d_zp.ptr, d_pixels.ptr, d_offsets.ptr, geo.n_rings, n_ring_pixels, float(bw_constant), d_scratch.ptr, d_modes.ptr, d_counts.ptr))
		else:
			sq_ptr = None if square is None else square.ptr
			ctx._check(ctx.lib.tp_radial_zeropoint(ctx.handle, frames.ptr, T, R * C, R * C, sq_ptr, R * C, ex_ptr, estride, float(flux_cutoff),
				d_partial.ptr, n_partial, d_zp.ptr))
			ctx._check(ctx.lib.tp_radial_ring_modes(ctx.handle, frames.ptr, T, R * C, R * C, sq_ptr, R * C, ex_ptr, estride, float(flux_cutoff),
				d_zp.ptr, d_pixels.ptr, d_offsets.ptr, geo.n_rings, n_ring_pixels, float(bw_constant), d_scratch.ptr, d_modes.ptr, d_counts.ptr))
		if on_device:
			ctx._check(ctx.lib.tp_radial_profiles(ctx.handle, T, geo.n_rings, d_modes.ptr, d_bin_center.ptr, int(radial_smooth or 0), K,
				d_knots.ptr, d_coefs.ptr, d_nk.ptr))
		else:
			knots, coefs, n_knots = radial_profiles(d_modes.to_host(), geo.bin_center, radial_smooth, max_knots=K)
			for dst, src in ((d_knots, knots), (d_coefs, coefs), (d_nk, n_knots)):
				ctx._check(ctx.lib.tp_memcpy_h2d(ctx.handle, dst.ptr, src.ctypes.data, src.nbytes))
		if implicit:
			mesh_h, nm_h = _square_component(ctx, frames, flux_cutoff, box, exclude, estride, None, None, want_host=return_mask and last, work=work,
				radial_spec=radial_spec)
			zoom_spec = _lib.tp_zoom_image(work['coef'].ptr, work['vmin'].ptr, work['vmax'].ptr, ny, nx, int(box), C)
		else:
			if radial is None:
				radial = ctx.empty((T, R, C), 'float32')
			ctx._check(ctx.lib.tp_radial_evaluate(ctx.handle, T, R, C, R * C, float(TESS_SCIENCE_COLUMN), float(geo.xcen), float(geo.ycen),
				d_knots.ptr, d_coefs.ptr, d_nk.ptr, K, d_zp.ptr, None, 0, radial.ptr))
			if square is None:
				square = ctx.empty((T, R, C), 'float32')
			mesh_h, nm_h = _square_component(ctx, frames, flux_cutoff, box, exclude, estride, radial, square, want_host=return_mask and last, work=work)
		if details is not None:
			details.setdefault('s2', []).append(d_modes.to_host())
			details.setdefault('zeropoint', []).append(d_zp.to_host())
			details.setdefault('n_knots', []).append(d_nk.to_host())
			details.setdefault('counts', []).append(d_counts.to_host())
	# total background (:209); a frame in which everything is masked is NaN (:99-102)
	if implicit:
		ctx._check(ctx.lib.tp_radial_evaluate_zoom(ctx.handle, T, R, C, R * C, ctypes.byref(radial_spec), ctypes.byref(zoom_spec), out.ptr))
	else:
		ctx._check(ctx.lib.tp_radial_evaluate(ctx.handle, T, R, C, R * C, float(TESS_SCIENCE_COLUMN), float(geo.xcen), float(geo.ycen),
			d_knots.ptr, d_coefs.ptr, d_nk.ptr, K, d_zp.ptr, square.ptr, R * C, out.ptr))
	ctx.sync()
	return (out, mesh_h, nm_h) if return_mask else out


def manual_exclude_columns(n_frames, n_cols, is_tess=False, camera=None, ccd=None, cadenceno=None, tstart=None, tstop=None):
	"""
	The header rules of ``pixel_flags.pixel_manual_exclude`` (pixel_flags.py:13-52) for a stack of frames: what they exclude is
	always a suffix of the columns, so the result is the first excluded column per frame (``n_cols`` = nothing, int32).
	``cadenceno``: FFIINDEX per frame (None or a negative entry = card missing), ``tstart`` / ``tstop``: TSTART / TSTOP.
	The third rule -- a TESS image that is zero everywhere (:54-56) -- needs the pixels: ``tp_frames_pixel_flags`` applies it.
	"""
	first = np.full(n_frames, n_cols, dtype='int32')
	if not is_tess:
		return first
	tstart = np.asarray(tstart, dtype='float64')
	time = 0.5 * (tstart + np.asarray(tstop, dtype='float64'))
	cad = np.full(n_frames, np.inf) if cadenceno is None else np.asarray(cadenceno, dtype='float64').copy()
	cad[cad < 0] = np.inf
	if camera == 1 and ccd == 4:
		# Mars floods the registers of output channel D in the beginning of Sector 1
		mars = (cad <= 4724) | (tstart <= 1325.881282301840)
		first[mars] = min(1536, n_cols)
	else:
		mars = np.zeros(n_frames, dtype=bool)
	if camera == 1:
		# excessive Earth-shine (an elif of the Mars rule)
		earth = ((11354 <= cad) & (cad <= 11366)) | ((1464.0158778 <= time) & (time <= 1464.265871))
		first[earth & ~mars] = 0
	return first


def pixel_flags_frames(ctx, raw, first_excluded=None, is_tess=False, flux_cutoff=8e4):
	"""
	The pixel flags the prepare stage stores with every frame (prepare.py:296-297: NotUsedForBackground from the mask of
	``fit_background``; :406-408: ManualExclude).  ``raw``: float32 DeviceArray ``(T, R, C)``; ``first_excluded``: the result of
	:func:`manual_exclude_columns` (or None).  Returns ``(flags uint8 DeviceArray (T, R, C), all_zero bool (T,))``.
	"""
	T, R, C = raw.shape
	flags = ctx.empty((T, R, C), 'uint8')
	d_zero = ctx.empty((T,), 'int32')
	d_first = None if first_excluded is None else ctx.array(np.asarray(first_excluded, dtype='int32'))
	ctx._check(ctx.lib.tp_frames_pixel_flags(ctx.handle, raw.ptr, T, R, C, C, R * C, None if d_first is None else d_first.ptr, 1 if is_tess else 0,
		float(flux_cutoff), PIXEL_NOT_USED_FOR_BACKGROUND, PIXEL_MANUAL_EXCLUDE, d_zero.ptr, flags.ptr))
	return flags, d_zero.to_host().astype(bool)


def prepare_frames(ctx, raw, raw_err, quality, cadence=1800, flux_cutoff=8e4, pixel_flags=None, backapp=False, camera=None, ccd=None,
	headers=None, backgrounds_pixels_threshold=0.5):
	"""
	The image arithmetic of ``prepare_photometry`` for one CCD (prepare.py:265-470) on device-resident stacks ``(T, R, C)``:
	pixel flags (manual excludes and the background mask, :296-297, :406-408), backgrounds (B1), their smoothing over
	``time_smooth`` frames (B2, :258, :317-335), ``images = raw - background`` with manually excluded pixels set to NaN in
	image and error (B3, :419-425), the sum image over the good-quality frames (A1, :450-453, :459) and the
	``backgrounds_pixels_used`` image (:435, :464-466).

	``headers``: dict of the FFI header cards the stage reads -- ``is_tess`` (real TESS frames: ``FFIImage.is_tess``), ``CAMERA``,
	``CCD``, and per frame ``FFIINDEX`` (optional), ``TSTART``, ``TSTOP``.  With it the pixel flags are built here; TESS frames get
	the radial background component.  Without it ``pixel_flags`` (uint8 ``(T, R, C)``, ManualExclude / NotUsedForBackground bits
	only, as at this point of the reference) is used as given, and ``camera`` / ``ccd`` select the TESS background branch.

	Returns a dict of DeviceArrays: ``backgrounds, images, images_err`` float32 ``(T, R, C)``, ``sumimage`` float64 ``(R, C)``,
	and, when flags exist, ``pixel_flags`` uint8 ``(T, R, C)`` and ``backgrounds_pixels_used`` uint8 ``(R, C)``.
	"""
	T, R, C = raw.shape
	time_smooth = {1800: 3, 600: 9}[int(cadence)]
	if headers is not None:
		is_tess = bool(headers.get('is_tess', False))
		first = manual_exclude_columns(T, C, is_tess, headers.get('CAMERA'), headers.get('CCD'), headers.get('FFIINDEX'),
			headers.get('TSTART'), headers.get('TSTOP'))
		pixel_flags, _ = pixel_flags_frames(ctx, raw, first, is_tess, flux_cutoff)
		if is_tess:
			camera, ccd = headers.get('CAMERA'), headers.get('CCD')
	# the flags are the mask of backgrounds.py:89-97 (any set bit = masked) as long as only these two bits exist
	bkg_us = fit_background_frames(ctx, raw, flux_cutoff=flux_cutoff, exclude=pixel_flags, camera=camera, ccd=ccd)   # TESS frames: with the radial component
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
	out = {'backgrounds': bkg, 'images': images, 'images_err': images_err, 'sumimage': sumimage}
	if pixel_flags is not None:
		used = ctx.empty((R, C), 'uint8')
		ctx._check(ctx.lib.tp_frames_used_in_background(ctx.handle, pixel_flags.ptr, T, R * C, PIXEL_NOT_USED_FOR_BACKGROUND,
			float(backgrounds_pixels_threshold), used.ptr))
		out['pixel_flags'] = pixel_flags
		out['backgrounds_pixels_used'] = used
	ctx.sync()
	return out


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
		own = indices[k:k + block]
		if len(own) < block and k >= block:
			# the reference fills one (R, C, 25) buffer block after block (prepare.py:562-566): the slots a short last block does
			# not overwrite still hold the frames of the previous block, and its nanmedian runs over all 25 slots
			own = own + indices[k - block + len(own):k]
		idx = ctx.array(np.asarray(own, dtype='int32'))
		keep.append(idx)
		ctx._check(ctx.lib.tp_frames_block_median_accumulate(ctx.handle, indicator.ptr, R * C, R * C, idx.ptr, len(own), mean.ptr))
	ctx.sync()
	nblocks = int(np.ceil(T / block))
	m = mean.to_host() / nblocks
	mean = ctx.array(m)
	ctx._check(ctx.lib.tp_frames_threshold_flags(ctx.handle, indicator.ptr, mean.ptr, float(threshold), PIXEL_BACKGROUND_SHENANIGANS, R * C, T,
		pixel_flags.ptr))
	ctx.sync()
	return indicator, mean
