# -*- coding: utf-8 -*-
"""
Batched per-target pipelines over a stamp cube resident in HBM.

``run_aperture``: A1 sum image -> A2..A5b K2P2 masks + A7 contamination -> A6 extraction,
i.e. ``AperturePhotometry.do_photometry`` (photometry/AperturePhotometry/photometry.py:44-257)
for every target of the batch at once.
"""

import ctypes
import os
import numpy as np
from . import engine, _lib
from .engine import TESS_DEFAULT_BITMASK
from ._lib import TessphotError
from .device import DeviceCube, device_view, round_up


class ApertureBatch(object):
	"""Device-resident inputs of a batch (cubes + per-target metadata)."""

	def __init__(self, ctx, scene, cubes='host'):
		self.ctx = ctx
		self.scene = scene
		# raw mode: the cubes hold the RAW flux; the background is estimated on the device (B*, B2) and
		# subtracted on the fly (B3) -- nothing but the raw + error cubes ever lives in HBM.
		self.raw_mode = False
		if isinstance(cubes, str) and cubes == 'host':
			self.images = DeviceCube.from_host(ctx, scene.images)
			self.images_err = DeviceCube.from_host(ctx, scene.images_err)
			self.backgrounds = DeviceCube.from_host(ctx, scene.backgrounds)
		elif isinstance(cubes, str) and cubes == 'host_aperture_only':
			# BASELINE configs[1]: images + errors, no background cube (flux_background comes out NaN)
			self.images = DeviceCube.from_host(ctx, scene.images)
			self.images_err = DeviceCube.from_host(ctx, scene.images_err)
			self.backgrounds = None
		elif isinstance(cubes, str) and cubes == 'host_raw':
			self.images = DeviceCube.from_host(ctx, scene.raw)
			self.images_err = DeviceCube.from_host(ctx, scene.raw_err)
			self.backgrounds = None
			self.raw_mode = True
		elif 'raw' in cubes:
			self.images, self.images_err, self.backgrounds = cubes['raw'], cubes.get('raw_err', cubes.get('images_err')), None
			self.raw_mode = True
		else:
			self.images, self.images_err, self.backgrounds = cubes['images'], cubes.get('images_err'), cubes.get('backgrounds')
		self.time_smooth = {1800: 3, 600: 9}.get(int(round(getattr(scene, 'cadence_s', 1800))), 3) # prepare.py:258
		self.n_targets = self.images.n_targets
		self.n_cad = self.images.n_cad
		self.height, self.width = self.images.height, self.images.width
		# the per-batch metadata -- quality flags, time stamps, stamps, the ragged catalogue and the target columns -- travel as ONE
		# host block and one upload (a batched entry builds a batch per stamp-size group and round: fifteen small uploads each were
		# a fifth of its time); the arrays below are views into that block
		c = scene.catalog
		fields = [('quality', np.asarray(scene.quality, dtype='int32')), ('time', np.asarray(scene.time, dtype='float64')),
			('stamps', np.asarray(scene.stamps, dtype='int32')), ('cat_offsets', np.asarray(scene.cat_offsets, dtype='int64')),
			('cat_starid', np.asarray(c['starid'], dtype='int64')), ('cat_tmag', np.asarray(c['tmag'], dtype='float32')),
			('cat_row', np.asarray(c['row'], dtype='float32')), ('cat_column', np.asarray(c['column'], dtype='float32')),
			('cat_row_stamp', np.asarray(c['row_stamp'], dtype='float32')), ('cat_column_stamp', np.asarray(c['column_stamp'], dtype='float32')),
			('target_pos_row', np.asarray(scene.target_pos_row, dtype='float64')), ('target_pos_column', np.asarray(scene.target_pos_column, dtype='float64')),
			('target_tmag', np.asarray(scene.target_tmag, dtype='float64')), ('target_starid', np.asarray(scene.target_starid, dtype='int64'))]
		offs, total = [], 0
		for _name, a in fields:
			offs.append(total)
			total = -(-(total + max(a.nbytes, 16)) // 256) * 256
		total = -(-total // 4) * 4
		if total <= (256 << 10):
			# small blocks: the library's ring of page-locked memory (tp_memcpy_h2d returns once the copy is queued)
			blob = np.zeros(total, dtype='uint8')
			for (_name, a), o in zip(fields, offs):
				blob[o:o + a.nbytes] = np.ascontiguousarray(a).view('uint8').ravel()
			self._meta = ctx.array(blob)
		else:
			# (page-locked, from the context's pool: the copy is queued and the host goes on -- a pageable block above 256 KB would wait for its DMA)
			self._meta_host = ctx.pinned_block(total)
			blob = self._meta_host.array
			for (_name, a), o in zip(fields, offs):
				blob[o:o + a.nbytes] = np.ascontiguousarray(a).view('uint8').ravel()
			self._meta = ctx.empty((total,), 'uint8')
			ctx._check(ctx.lib.tp_upload_cube_async(ctx.handle, self._meta.ptr, total // 4, self._meta_host.ptr, total // 4, 1, total // 4))
		for (name, a), o in zip(fields, offs):
			setattr(self, name, device_view(ctx, self._meta.ptr + o, a.shape, a.dtype, base=self._meta))
		ap = getattr(scene, 'aperture', None)
		if ap is None:
			# every pixel collected (bit 1 of BasePhotometry.aperture, BasePhotometry.py:1043; the kernels require a finite sum-image
			# pixel themselves): set on the device, 0x01010101 per pixel -- bit 1 is what the mask builder reads
			self.aperture = ctx.empty((self.n_targets, self.height, self.width), 'int32')
			self.aperture.fill_bytes(1)
		else:
			self.aperture = ctx.array(np.asarray(ap, dtype='int32'))

	def release_host(self):
		"""The page-locked block the metadata were uploaded from goes back to the context's pool (call after a synchronisation)."""
		h = self.__dict__.pop('_meta_host', None)
		if h is not None:
			self.ctx.pinned_release(h)

	def __del__(self):
		# the upload may still be in flight when the last reference goes: wait for the stream, then the block goes back to the pool
		h = self.__dict__.pop('_meta_host', None)
		if h is not None:
			try:
				self.ctx.sync()
				self.ctx.pinned_release(h)
			except Exception: # noqa: B902
				pass

	#: per-target arrays (sliced by :meth:`chunk`); the flat catalogue arrays are shared because the
	#: CSR offsets are absolute
	_PER_TARGET = ('images', 'images_err', 'backgrounds', 'stamps', 'target_pos_row', 'target_pos_column',
		'target_tmag', 'target_starid', 'aperture')

	def chunk(self, start, count):
		"""Non-owning view of the targets ``[start, start+count)`` (same HBM)."""
		v = ApertureBatch.__new__(ApertureBatch)
		v.__dict__.update(self.__dict__)
		v.__dict__.pop('_meta_host', None)   # the view does not own the upload's host block
		for name in self._PER_TARGET:
			a = getattr(self, name)
			setattr(v, name, None if a is None else a.slice0(start, count))
		v.cat_offsets = self.cat_offsets.slice0(start, count + 1)
		if len(self.quality.shape) == 2:
			v.quality = self.quality.slice0(start, count)
		v.n_targets = int(count)
		return v


class ApertureWork(object):
	"""
	Device-resident outputs / scratch of the aperture pipeline (allocated once, reused per step).

	``packed=True`` carves the per-target results a scheduler consumes -- the light-curve block ``[5][Nt][T]`` float64,
	``contamination`` float64, ``status`` / ``flags`` int32 and the ``mask`` uint8 (the output
	block of a rank), with ``psf=True`` also the LinPSF light curve, contamination and status -- out of ONE allocation
	(``self.block``, layout: ``comm.packed_block_layout``), so that the per-step gather of a multi-GPU run is a single message.
	"""

	def __init__(self, ctx, batch, packed=False, psf=False, capacity=None, cat_capacity=0, extras=False):
		"""``capacity`` >= the batch's targets lays the packed block out for that many (the padded shard size every rank of a
		sharded run sends); ``cat_capacity`` > 0 puts the catalogue flags ``cat_in_mask`` into the block as well."""
		from . import comm as tpcomm
		Nt, T, H, W = batch.n_targets, batch.n_cad, batch.height, batch.width
		self.sumimage = self.diagnostics = None
		self.block = None
		self.psf_flux = self.psf_contamination = self.psf_status = None
		n_cat = max(int(batch.scene.cat_offsets[-1]), 1)
		self.cat_in_mask = None
		if packed:
			cap = Nt if capacity is None else int(capacity)
			if cap < Nt or (cat_capacity and cat_capacity < n_cat):
				raise ValueError('block capacity below the size of the batch')
			layout, nbytes = tpcomm.packed_block_layout(cap, T, H, W, psf=psf, n_cat=int(cat_capacity), extras=extras)
			Nt = cap
			self.block = ctx.zeros((nbytes,), 'uint8')
			b = self.block
			view = lambda name: device_view(ctx, b.ptr + layout[name][0], layout[name][1], layout[name][2], base=b) # noqa: E731
			self.lc = engine.LightCurves(ctx, Nt, T, block=view('lc'))
			self.contamination, self.status, self.flags, self.mask = view('contamination'), view('status'), view('flags'), view('mask')
			if psf:
				self.psf_flux, self.psf_contamination, self.psf_status = view('psf_flux'), view('psf_contamination'), view('psf_status')
			if cat_capacity:
				self.cat_in_mask = view('cat_in_mask')
			if extras:   # (with a capacity above the batch the views cover the capacity: the kernels address the first Nt entries)
				self.sumimage, self.diagnostics = view('sumimage'), view('diagnostics')
			self.block_layout = layout
			Nt = batch.n_targets
		else:
			self.mask = ctx.zeros((Nt, H, W), 'uint8')
			self.status = ctx.zeros((Nt,), 'int32')
			self.flags = ctx.zeros((Nt,), 'int32')
			self.contamination = ctx.zeros((Nt,), 'float64')
			self.lc = engine.LightCurves(ctx, Nt, T)
		self.diag = ctx.zeros((Nt, 8), 'float64')
		if self.cat_in_mask is None:
			self.cat_in_mask = ctx.zeros((n_cat,), 'uint8')
		if self.sumimage is None:
			self.sumimage = ctx.empty((Nt, H, W), 'float64')
		if self.diagnostics is None:
			self.diagnostics = ctx.zeros((Nt, 10), 'float64')
		self.bkg_raw = self.bkg = None
		if batch.raw_mode:
			pitch = batch.images.t_pitch
			self.bkg_raw = ctx.zeros((Nt, pitch), 'float32')
			self.bkg = ctx.zeros((Nt, pitch), 'float32')

	def chunk(self, start, count):
		"""Non-owning view of the targets ``[start, start+count)`` (same HBM)."""
		v = ApertureWork.__new__(ApertureWork)
		v.__dict__.update(self.__dict__)
		for name in ('sumimage', 'mask', 'status', 'flags', 'contamination', 'diag', 'lc', 'diagnostics', 'bkg_raw', 'bkg'):
			a = getattr(self, name)
			setattr(v, name, None if a is None else a.slice0(start, count))
		return v


def aperture_step(ctx, batch, work, masks_from=None, fused=True, sumimage_given=False):
	"""
	One pass of the aperture hot path over the batch; everything stays in HBM.

	``fused``: one launch with a wavefront per target (``tp_aperture_photometry``); ``False`` runs the three
	stand-alone kernels A1, K2P2, A6 back to back (bit-identical outputs).
	``masks_from``: optional ``(mask uint8 DeviceArray, status int32 DeviceArray)`` to bypass the
	on-device K2P2 (used by tests that inject the oracle's masks; implies the stand-alone kernels).
	"""
	subtract, backgrounds = None, batch.backgrounds
	if sumimage_given:
		# ``work.sumimage`` already holds the sum images (the FFI branch of BasePhotometry.sumimage: crops of the region's): A2..A7
		if batch.raw_mode or masks_from is not None:
			raise ValueError('sumimage_given: calibrated cubes and the on-device masks only')
		if fused:
			engine.aperture_photometry(ctx, batch, work, backgrounds=backgrounds, sumimage_given=True)
			return work
		engine.k2p2_masks(ctx, batch, work)
		engine.aperture_extract(ctx, batch.images, batch.images_err, backgrounds, work.mask, batch.stamps, status=work.status, out=work.lc)
		return work
	if batch.raw_mode and fused and masks_from is None:
		# the raw cube is read ONCE: B*, B2 and the sum image of raw - background (A1 with B3 on the fly) in one pass,
		# then the mask and the extraction, which read in-mask pixel rows only
		engine.background_sumimage(ctx, batch.images, batch.quality, batch.time_smooth, bkg_raw=work.bkg_raw, bkg=work.bkg, sumimage=work.sumimage)
		engine.aperture_photometry(ctx, batch, work, subtract=work.bkg, backgrounds=work.bkg, sumimage_given=True)   # A2..A7
		return work
	if batch.raw_mode:
		engine.background_stamp(ctx, batch.images, out=work.bkg_raw)                           # B*
		engine.smooth_time(ctx, work.bkg_raw, batch.n_cad, batch.time_smooth, out=work.bkg)    # B2
		subtract = backgrounds = work.bkg                                                      # B3 on the fly
	if fused and masks_from is None:
		engine.aperture_photometry(ctx, batch, work, subtract=subtract, backgrounds=backgrounds)   # A1..A7
		return work
	engine.sumimage(ctx, batch.images, batch.quality, out=work.sumimage, subtract=subtract)    # A1
	if masks_from is None:
		engine.k2p2_masks(ctx, batch, work)                                                    # A2..A5b, A7
		mask, status = work.mask, work.status
	else:
		mask, status = masks_from
	engine.aperture_extract(ctx, batch.images, batch.images_err, backgrounds, mask, batch.stamps, status=status, out=work.lc,
		subtract=subtract)                                                                     # A6
	return work


def aperture_diagnostics(ctx, batch, work, status=None, mask=None):
	"""
	The light-curve diagnostics of the batch (BasePhotometry.py:1343-1407) from the device-resident outputs of
	:func:`aperture_step`; fills ``work.diagnostics`` ``(Nt, 10)`` (columns ``engine.DIAGNOSTICS_COLUMNS``).
	"""
	engine.lightcurve_diagnostics(ctx, work.lc, batch.time, batch.quality, status=work.status if status is None else status,
		sumimage=work.sumimage, mask=work.mask if mask is None else mask, out=work.diagnostics)
	return work.diagnostics


def run_aperture(ctx, scene, cubes='host', masks=None, diagnostics=True):
	"""
	Convenience: upload (or adopt device cubes), run one :func:`aperture_step`, download.

	Returns a dict of host arrays: ``sumimage, mask, status, flags, contamination, diag, cat_in_mask,
	flux, flux_err, flux_background, pos_centroid, diagnostics``.
	"""
	batch = ApertureBatch(ctx, scene, cubes=cubes)
	work = ApertureWork(ctx, batch)
	masks_from = None
	if masks is not None:
		m = ctx.array(np.asarray(masks, dtype='uint8'))
		st = ctx.array(np.ones(batch.n_targets, dtype='int32'))
		masks_from = (m, st)
	aperture_step(ctx, batch, work, masks_from=masks_from)
	if diagnostics and masks_from is None:
		aperture_diagnostics(ctx, batch, work)
	elif diagnostics:
		aperture_diagnostics(ctx, batch, work, status=masks_from[1], mask=masks_from[0])
	ctx.sync()
	out = work.lc.to_host()
	if diagnostics:
		out['diagnostics'] = work.diagnostics.to_host()
	out['sumimage'] = work.sumimage.to_host()
	if batch.raw_mode:
		out['background'] = work.bkg.to_host()[:, :batch.n_cad]
		out['background_raw'] = work.bkg_raw.to_host()[:, :batch.n_cad]
	if masks_from is None:
		out['mask'] = work.mask.to_host()
		out['status'] = work.status.to_host()
		out['flags'] = work.flags.to_host()
		out['contamination'] = work.contamination.to_host()
		out['diag'] = work.diag.to_host()
		out['cat_in_mask'] = work.cat_in_mask.to_host()
	else:
		out['mask'] = masks_from[0].to_host()
		out['status'] = masks_from[1].to_host()
	return out


class LinPSFBatch(object):
	"""
	Device-resident inputs of the LinPSF pipeline (linpsf_photometry.py:79-219) for a batch:
	image cube, per-target spline tables (P1), fitted-star lists and their per-cadence positions
	(``catalog_attime``: reference position + jitter, host geometry).
	"""

	def __init__(self, ctx, scene, prf_model, images=None, subtract=None, work=None):
		"""``work``: an ``ApertureWork(packed=True, psf=True)`` whose block receives the light curve, contamination and status."""
		from . import psf as hpsf
		self.ctx, self.scene, self.model = ctx, scene, prf_model
		self.images = DeviceCube.from_host(ctx, scene.images) if images is None else images
		self.subtract = subtract
		sel, star_offsets, target_index = hpsf.select_stars(scene.catalog, scene.cat_offsets, scene.target_starid)
		self.sel, self.star_offsets_h, self.target_index_h = sel, star_offsets, target_index
		self.max_stars = int(np.diff(star_offsets).max()) if len(star_offsets) > 1 else 1
		rs = scene.catalog['row_stamp'][sel].astype('float64')
		cs = scene.catalog['column_stamp'][sel].astype('float64')
		T = scene.n_cad
		pitch = self.images.t_pitch
		pos_row = np.zeros((len(rs), pitch))
		pos_col = np.zeros((len(rs), pitch))
		pos_row[:, :T] = rs[:, None] + scene.jitter[None, :, 1]
		pos_col[:, :T] = cs[:, None] + scene.jitter[None, :, 0]
		self.pos_row, self.pos_col = ctx.array(pos_row), ctx.array(pos_col)
		self.star_offsets, self.target_index = ctx.array(star_offsets), ctx.array(target_index)
		self.base_coef = ctx.array(prf_model.base_coef)
		self.weights = ctx.array(prf_model.weights(scene.stamps))
		self.tx, self.ty = ctx.array(prf_model.tx), ctx.array(prf_model.ty)
		self.coef = ctx.empty((scene.n_targets, prf_model.base_coef.shape[1]), 'float64')
		self.out = self.result_for(work)
		self.n_fit_stars = len(rs)

	def result_for(self, work=None):
		"""Output arrays of the fit; the per-target results inside ``work``'s packed block when given."""
		n_fit = int(self.star_offsets_h[-1])
		if work is None or work.psf_flux is None:
			return engine.LinPSFResult(self.ctx, self.scene.n_targets, n_fit, self.scene.n_cad)
		return engine.LinPSFResult(self.ctx, self.scene.n_targets, n_fit, self.scene.n_cad, flux=work.psf_flux,
			contamination=work.psf_contamination, status=work.psf_status)


def linpsf_step(ctx, batch, cutoff_radius=5.0, out=None, subtract=None):
	"""One pass of the LinPSF hot path: P1 table blend, P2-P4 fit + contamination."""
	out = batch.out if out is None else out
	engine.linpsf_prf(ctx, batch.base_coef, batch.weights, out=batch.coef)
	engine.linpsf_fit(ctx, batch.images, batch.coef, batch.tx, batch.ty, batch.star_offsets, batch.target_index,
		batch.pos_row, batch.pos_col, batch.max_stars, cutoff_radius=cutoff_radius, subtract=batch.subtract if subtract is None else subtract, out=out)
	return out


#--------------------------------------------------------------------------------------------------
# Aperture photometry of many targets of one CCD region, INCLUDING the stamp-resize retries
#--------------------------------------------------------------------------------------------------
class FrameStack(object):
	"""
	The image groups of one CCD region resident in HBM -- what the reference keeps in its HDF5 file as ``images/%04d``,
	``images_err/%04d``, ``backgrounds/%04d`` (prepare.py:136-141) and reads back cut-out by cut-out for every target and every
	stamp resize (BasePhotometry.py:720-751).  ``frames``: dict of float32 host arrays ``(T, R, C)`` covering CCD rows
	``[row0, row0 + R)`` and columns ``[col0, col0 + C)`` (``row0 / col0`` = PIXEL_OFFSET_ROW / COLUMN of the file).
	"""

	def __init__(self, ctx, frames, row0, col0, sumimage=None):
		"""``frames``: host arrays (uploaded) or float32 DeviceArrays ``(T, R, C)`` already in HBM (e.g. from ``frameio.load_stack``).
		``sumimage`` (or ``frames['sumimage']``): the region's sum image, float64 ``(R, C)`` -- the HDF5 dataset ``sumimage`` of the
		reference's file (prepare.py:450-453, 459; ``prepare.prepare_frames`` returns it); without one it is formed from the image
		stack when a batch first needs it (:meth:`sumimage_for`)."""
		from .device import DeviceArray
		self.ctx = ctx
		self.row0, self.col0 = int(row0), int(col0)
		self.names = ('images', 'images_err', 'backgrounds')
		self.dev = {k: (frames[k] if isinstance(frames[k], DeviceArray) else ctx.array(np.ascontiguousarray(frames[k], dtype='float32'))) for k in self.names}
		self.n_cad, self.n_rows, self.n_cols = self.dev['images'].shape
		self.limits = (self.row0, self.row0 + self.n_rows, self.col0, self.col0 + self.n_cols)
		if sumimage is None and hasattr(frames, 'get'):
			sumimage = frames.get('sumimage')
		self._sumimage = self._sumimage_key = None
		self._time_major = None
		if sumimage is not None:
			self._sumimage = sumimage if isinstance(sumimage, DeviceArray) else ctx.array(np.ascontiguousarray(sumimage, dtype='float64'))
			if tuple(self._sumimage.shape) != (self.n_rows, self.n_cols):
				raise ValueError('the sum image must have the shape of a frame')
			self._sumimage_key = 'given'
			ctx.sync()

	def sumimage_for(self, quality):
		"""
		The sum image of the region, float64 DeviceArray ``(R, C)``: the mean of the finite pixels of the frames with good quality
		(prepare.py:450-453, 459).  For an FFI target the reference crops it (``BasePhotometry.sumimage``, BasePhotometry.py:1001-1006)
		-- no sum over a stamp's own cube -- and so do the passes of :func:`aperture_frames`.  Given with the stack, or formed once per
		quality series (``tp_frames_sumimage``: the float64 sums in cadence order, as prepare.py accumulates them).
		"""
		if self._sumimage_key == 'given':
			return self._sumimage
		q = np.ascontiguousarray(quality, dtype='int32')
		key = q.tobytes()
		if self._sumimage_key != key:
			# (a NEW array per quality series: jobs in flight may still read the one of the series before; they keep it alive)
			ctx = self.ctx
			sumimage = ctx.empty((self.n_rows, self.n_cols), 'float64')
			dq = ctx.array(q)
			ctx._check(ctx.lib.tp_frames_sumimage(ctx.handle, self.n_cad, self.n_rows * self.n_cols, self.n_rows * self.n_cols, self.dev['images'].ptr, dq.ptr,
				int(TESS_DEFAULT_BITMASK), sumimage.ptr))
			ctx.sync()      # other streams (the jobs' contexts) read it
			dq.free()
			self._sumimage, self._sumimage_key = sumimage, key
		return self._sumimage

	def time_major(self):
		"""
		The three stacks once more, TIME-MAJOR: float32 DeviceArrays ``(R * C, t_pitch)`` (``tp_frames_transpose``; formed on first use,
		kept with the stack), or ``None`` when ``TESSPHOT_FRAMES_TIME_MAJOR=0`` or the device has no room for them.  With them the
		native engine cuts nothing: a mask pixel's time series is one row of the stack (``tp_aperture_extract_stack``), where every
		batch of targets used to cut the in-mask rows out of all frames again -- the reference's ``_load_cube`` per target and per stamp
		resize (BasePhotometry.py:720-751).  Costs the stacks' size a second time in HBM, and one pass over them per region.
		"""
		if self._time_major is False or os.environ.get('TESSPHOT_FRAMES_TIME_MAJOR', '1') == '0':
			return None
		if self._time_major is None:
			ctx = self.ctx
			t_pitch = (self.n_cad + 31) // 32 * 32
			npix = self.n_rows * self.n_cols
			out = {}
			try:
				for k in self.names:
					out[k] = ctx.empty((npix, t_pitch), 'float32')
					ctx._check(ctx.lib.tp_frames_transpose(ctx.handle, self.dev[k].ptr, self.n_cad, npix, npix, out[k].ptr, t_pitch))
				ctx.sync()      # other streams (the jobs' contexts) read them
			except TessphotError:
				for a in out.values():
					a.free()
				self._time_major = False
				return None
			self._time_major = (out, t_pitch)
		return self._time_major

	def cut_lazy(self, ctx, stamps, height, width):
		"""The cut cubes of a group on ``ctx``'s stream, the stamp list uploaded there too."""
		d = ctx.array(np.asarray(stamps, dtype='int32'))
		cubes = self.cut(d, height, width, ctx=ctx)
		cubes['_stamps'] = d   # alive until the cut has run
		return cubes

	def cut(self, stamps_dev, height, width, ctx=None):
		"""The three stamp cubes of a group of same-sized stamps (``tp_cut_stamps``), on ``ctx``'s stream (default: the stack's)."""
		ctx = self.ctx if ctx is None else ctx
		cubes = engine.cut_stamps_multi(ctx, [self.dev[k] for k in self.names], stamps_dev, height, width, self.row0, self.col0)
		return dict(zip(self.names, cubes))


class _Messages(object):
	"""Collects what the reference's plugin would have logged at WARNING level and above, as ``"LEVEL: message"`` strings."""
	def __init__(self):
		self.items = []

	def error(self, msg, *a):
		self.items.append('ERROR: ' + (msg % a if a else msg))

	def warning(self, msg, *a):
		self.items.append('WARNING: ' + (msg % a if a else msg))

	def info(self, msg, *a):
		pass


def _catalog_of_stamp(catalog, stamp, buffer_size=5):
	"""Stars inside the stamp plus its 5-pixel buffer with the float32 stamp coordinates of BasePhotometry.catalog
	(BasePhotometry.py:1094-1181); the same selection as ``source.MemoryStampSource.catalog_in_stamp``."""
	r1, r2, c1, c2 = stamp
	row, col = catalog['row'], catalog['column']
	sel = (row >= r1 - 0.5 - buffer_size) & (row < r2 - 0.5 + buffer_size) & (col >= c1 - 0.5 - buffer_size) & (col < c2 - 0.5 + buffer_size)
	col64, row64 = np.asarray(col[sel], dtype='float64'), np.asarray(row[sel], dtype='float64')
	return {'starid': np.asarray(catalog['starid'][sel], dtype='int64'), 'tmag': np.asarray(catalog['tmag'][sel], dtype='float32'),
		'column': col64.astype('float32'), 'row': row64.astype('float32'),
		'column_stamp': (col64 - c1).astype('float32'), 'row_stamp': (row64 - r1).astype('float32')}


class _CatalogIndex(object):
	"""The catalogue of a CCD region binned into cells of ``cell`` x ``cell`` pixels (stars sorted by cell), for
	:func:`_catalogs_of_stamps`: the candidates of a stamp are the stars of the few cells it overlaps -- a row-sorted catalogue
	alone hands every stamp all the stars of its band of rows, hundreds of candidates for the handful that are kept."""
	def __init__(self, catalog, cell=16):
		self.catalog = catalog
		self.cell = int(cell)
		row = np.asarray(catalog['row'], dtype='float64')
		col = np.asarray(catalog['column'], dtype='float64')
		if len(row):
			self.r0 = int(np.floor(np.nanmin(row))) if np.isfinite(row).any() else 0
			self.c0 = int(np.floor(np.nanmin(col))) if np.isfinite(col).any() else 0
			finite = np.isfinite(row) & np.isfinite(col)
			cr = np.where(finite, np.floor((row - self.r0) / self.cell), 0).astype('int64')
			cc = np.where(finite, np.floor((col - self.c0) / self.cell), 0).astype('int64')
			self.n_cr, self.n_cc = int(cr.max()) + 1, int(cc.max()) + 1
			cid = np.where(finite, cr * self.n_cc + cc, self.n_cr * self.n_cc)   # stars without a position: a cell no stamp asks for
		else:
			self.r0 = self.c0 = 0
			self.n_cr = self.n_cc = 1
			cid = np.zeros(0, dtype='int64')
		self.order = np.argsort(cid, kind='stable')
		self.cell_start = np.searchsorted(cid[self.order], np.arange(self.n_cr * self.n_cc + 1), side='left')


def _catalogs_of_stamps(index, stamps, buffer_size=5):
	"""
	:func:`_catalog_of_stamp` for all stamps of a group at once, in CSR form: ``(offsets int64 [n + 1], dict of concatenated
	arrays)``; the stars of a stamp keep their catalogue order.  Candidate stars come from the cells of the binned catalogue the
	stamp (plus its buffer) overlaps, one contiguous run of the cell-sorted catalogue per row of cells; the row / column tests and the
	float32 stamp coordinates are the expressions of the per-stamp function evaluated on arrays.
	"""
	cat = index.catalog
	st = np.asarray(stamps, dtype='int64').reshape(-1, 4)
	n = len(st)
	B = index.cell
	rlo, rhi = st[:, 0] - 0.5 - buffer_size, st[:, 1] - 0.5 + buffer_size      # row >= rlo, row < rhi
	clo, chi = st[:, 2] - 0.5 - buffer_size, st[:, 3] - 0.5 + buffer_size
	cr0 = np.clip(np.floor((rlo - index.r0) / B).astype('int64'), 0, index.n_cr - 1)
	cr1 = np.clip(np.floor((rhi - index.r0) / B).astype('int64'), -1, index.n_cr - 1)
	cc0 = np.clip(np.floor((clo - index.c0) / B).astype('int64'), 0, index.n_cc - 1)
	cc1 = np.clip(np.floor((chi - index.c0) / B).astype('int64'), -1, index.n_cc - 1)
	# one run [cell_start[first cell], cell_start[last cell + 1]) per (stamp, row of cells)
	nrows = np.maximum(cr1 - cr0 + 1, 0) * (cc1 >= cc0)
	run_stamp = np.repeat(np.arange(n), nrows)
	run_row = np.arange(int(nrows.sum())) - np.repeat(np.cumsum(nrows) - nrows, nrows) + np.repeat(cr0, nrows)
	a = index.cell_start[run_row * index.n_cc + cc0[run_stamp]]
	b = index.cell_start[run_row * index.n_cc + cc1[run_stamp] + 1]
	cnt = b - a
	which = np.repeat(run_stamp, cnt)
	pos = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt) + np.repeat(a, cnt)
	star = index.order[pos]
	col, row = np.asarray(cat['column'])[star], np.asarray(cat['row'])[star]
	keep = (row >= rlo[which]) & (row < rhi[which]) & (col >= clo[which]) & (col < chi[which])
	which, star = which[keep], star[keep]
	o = np.argsort(which * (len(index.order) + 1) + star, kind='stable')   # catalogue order inside every stamp
	which, star = which[o], star[o]
	offsets = np.concatenate(([0], np.cumsum(np.bincount(which, minlength=n)))).astype('int64')
	col64, row64 = np.asarray(cat['column'], dtype='float64')[star], np.asarray(cat['row'], dtype='float64')[star]
	arrays = {'starid': np.asarray(cat['starid'], dtype='int64')[star], 'tmag': np.asarray(cat['tmag'], dtype='float32')[star],
		'column': col64.astype('float32'), 'row': row64.astype('float32'),
		'column_stamp': (col64 - st[which, 2]).astype('float32'), 'row_stamp': (row64 - st[which, 0]).astype('float32')}
	return offsets, arrays


class _GroupScene(object):
	"""The metadata of a group of same-sized stamps in the form ``ApertureBatch`` takes (no host cubes)."""
	def __init__(self, stack, time, quality, cadence_s, stamps_list, cat_offsets, cat_arrays, targets, idx):
		self.n_targets = len(idx)
		self.n_cad = stack.n_cad
		self.height, self.width = stamps_list[0][1] - stamps_list[0][0], stamps_list[0][3] - stamps_list[0][2]
		self.time, self.quality, self.cadence_s = time, quality, cadence_s
		self.stamps = np.asarray(stamps_list, dtype='int32')
		self.cat_offsets = cat_offsets
		self.catalog = cat_arrays
		self.target_pos_row = np.asarray(targets['row'], dtype='float64')[idx]
		self.target_pos_column = np.asarray(targets['column'], dtype='float64')[idx]
		self.target_tmag = np.asarray(targets['tmag'], dtype='float64')[idx]
		self.target_starid = np.asarray(targets['starid'], dtype='int64')[idx]
		self.aperture = None


class _RawBlock(object):
	"""``nbytes`` bytes at ``address`` through the array interface (no ctypes array type per size)."""
	__slots__ = ('__array_interface__',)

	def __init__(self, address, nbytes):
		self.__array_interface__ = {'shape': (int(nbytes),), 'typestr': '|u1', 'data': (int(address), True), 'version': 3}


def _view_bytes(address, nbytes):
	"""Read-only uint8 view of host memory owned by the library (a job's page-locked block: alive until the job is released)."""
	return np.asarray(_RawBlock(address, nbytes))


class FramesResult(object):
	"""
	Columnar result of :func:`aperture_frames`: one entry per target in arrays (``status``, ``stamp``, ``stamp_resizes``, ``has_result``),
	the rare per-target extras (log messages, ``edge_flux``) in sparse dicts, and the arrays the device passes returned kept group by
	group (a group = the targets of one pass that share a stamp size): target ``i`` is row ``pos[i]`` of group ``group[i]``.  Nothing is
	allocated per target until a per-target view is asked for: ``result[i]`` builds the dict the list-based API used to return.
	"""

	def __init__(self, n):
		self.n = int(n)
		self.status = np.zeros(n, dtype='int32')
		self.stamp = np.full((n, 4), -1, dtype='int64')
		self.stamp_resizes = np.zeros(n, dtype='int32')
		self.has_result = np.zeros(n, dtype=bool)
		self.group = np.full(n, -1, dtype='int32')
		self.pos = np.zeros(n, dtype='int32')
		self.errors = {}       # target -> list of "LEVEL: message" strings (targets without messages have no entry)
		self.edge_flux = {}    # target -> flux on the stuck edges (haloswitch quick break)
		#: per device pass: dict of host arrays.  They are READ-ONLY VIEWS of the page-locked block the pass was downloaded into and
		#: live as long as this result: :meth:`release` (and the result's deletion) hands the blocks to the next job's download.  Copy
		#: what has to outlive the result (``result[i]`` does: the per-target dict holds copies).
		self.groups = []
		self._pinned = []      # (context, block): returned to the context's pool when the result goes
		self._job = None       # the native job (pipeline.FramesEngine) whose page-locked blocks the group arrays view

	def release(self):
		"""Give the page-locked blocks behind the group arrays back to their pool: the group arrays must not be used afterwards
		(they are dropped from ``groups``; a view a caller kept would read the next job's data)."""
		for ctx, blk in self._pinned:
			ctx.pinned_release(blk)
		self._pinned = []
		self.groups = []
		job, self._job = self._job, None
		if job is not None:
			job.release()

	def __del__(self):
		try:
			self.release()
		except Exception: # noqa: B902
			pass

	def __len__(self):
		return self.n

	def column(self, name, fill=np.nan):
		"""A per-target column gathered from the groups (``contamination``, ``mask_size``, or a ``DIAGNOSTICS_COLUMNS`` name)."""
		out = np.full(self.n, fill, dtype='float64')
		for g, grp in enumerate(self.groups):
			sel = np.flatnonzero(self.has_result & (self.group == g))
			if len(sel) == 0:
				continue
			rows = self.pos[sel]
			if name == 'contamination':
				out[sel] = grp['contamination'][rows]
			elif name == 'mask_size':
				out[sel] = grp['mask'][rows].reshape(len(rows), -1).sum(axis=1)
			else:
				out[sel] = grp['diagnostics'][rows, engine.DIAGNOSTICS_COLUMNS.index(name)]
		return out

	def __getitem__(self, i):
		i = int(i)
		if i < 0:
			i += self.n
		if not 0 <= i < self.n:
			raise IndexError(i)
		d = {'status': int(self.status[i]), 'errors': list(self.errors.get(i, ())), 'stamp_resizes': int(self.stamp_resizes[i])}
		if self.stamp[i, 1] >= 0:
			d['stamp'] = tuple(int(v) for v in self.stamp[i])
		if i in self.edge_flux:
			d['edge_flux'] = self.edge_flux[i]
		if self.has_result[i]:
			grp, j = self.groups[self.group[i]], int(self.pos[i])
			a, b = grp['cat_offsets'][j], grp['cat_offsets'][j + 1]
			inside = grp['cat_in_mask'][a:b].astype(bool)
			lc = grp['lc']   # the light-curve block (5, m, T): flux, flux_err, flux_background, centroid column, centroid row
			# (copies: the group arrays live in page-locked memory that goes back to the context's pool with this result)
			d.update(mask=grp['mask'][j].astype(bool), sumimage=np.array(grp['sumimage'][j]), flux=np.array(lc[0][j]), flux_err=np.array(lc[1][j]),
				flux_background=np.array(lc[2][j]), pos_centroid=np.stack((lc[3][j], lc[4][j]), axis=-1),   # (T, 2): column then row (BasePhotometry.py:428)
				contamination=float(grp['contamination'][j]),
				skip_targets=[int(s) for s in grp['cat_starid'][a:b][inside] if s != grp['target_starid'][j]],
				diagnostics=dict(zip(engine.DIAGNOSTICS_COLUMNS, grp['diagnostics'][j])))
		return d

	def __iter__(self):
		return (self[i] for i in range(self.n))


#: what the native engine reports as codes (csrc/frames.cpp) in the words of the reference's plugin / of the Python rounds below
_EVENT_TEXT = {1: 'ERROR: No flux above threshold.', 2: 'WARNING: No masks found. Using minimum aperture.',
	3: 'WARNING: No mask found for main target. Using minimum aperture.', 4: 'ERROR: Too many masks.',
	6: 'WARNING: Could not resize stamp any further.', 7: 'ERROR: Stamp resize hit limit. Haloswitch quick break.',
	8: 'ERROR: Too many stamp resizes.', 9: 'ERROR: No targets in mask.', 12: 'ValueError: Invalid stamp selected'}


class _NativeCatalog(object):
	"""The catalogue of a region inside the library (``tp_frames_catalog``: copied and binned into cells once)."""
	def __init__(self, lib, catalog):
		self.lib = lib
		sid = np.ascontiguousarray(catalog['starid'], dtype='int64')
		tm = np.ascontiguousarray(catalog['tmag'], dtype='float32')
		row = np.ascontiguousarray(catalog['row'], dtype='float64')
		col = np.ascontiguousarray(catalog['column'], dtype='float64')
		h = ctypes.c_void_p()
		rc = lib.tp_frames_catalog_create(len(sid), sid.ctypes.data, tm.ctypes.data, row.ctypes.data, col.ctypes.data, ctypes.byref(h))
		if rc != 0:
			raise TessphotError(rc, (lib.tp_last_error(None) or b'').decode())
		self.handle = h.value

	def __del__(self):
		h, self.handle = getattr(self, 'handle', None), None
		if h:
			self.lib.tp_frames_catalog_destroy(h)


class FramesJob(object):
	"""A batch in flight in the native engine: :meth:`collect` waits for it and returns its :class:`FramesResult`."""
	def __init__(self, engine, handle, n, T, keep):
		self.engine, self.handle, self.n, self.T = engine, handle, int(n), int(T)
		self._keep = keep      # the stack and the catalogue: alive while the job runs

	def release(self):
		h, self.handle = self.handle, None
		if h and self.engine.handle:
			self.engine.lib.tp_frames_release(h)

	def __del__(self):
		try:
			self.release()
		except Exception: # noqa: B902
			pass

	def done(self):
		"""Has the job's worker finished (:meth:`collect` would not wait)?"""
		d = ctypes.c_int32(0)
		self.engine.lib.tp_frames_poll(self.handle, ctypes.byref(d))
		return bool(d.value)

	def collect(self):
		from . import comm as tpcomm
		lib, h, n, T = self.engine.lib, self.handle, self.n, self.T
		rc = lib.tp_frames_wait(h)
		if rc != 0:
			msg = (lib.tp_last_error(None) or b'').decode()
			self.release()
			raise TessphotError(rc, msg)
		self._keep = None
		out = FramesResult(n)
		out._job = self
		ng, ne = ctypes.c_int32(), ctypes.c_int64()
		lib.tp_frames_counts(h, ctypes.byref(ng), ctypes.byref(ne))
		has = np.zeros(n, dtype='uint8')
		lib.tp_frames_targets(h, out.status.ctypes.data, out.stamp.ctypes.data, out.stamp_resizes.ctypes.data, has.ctypes.data,
			out.group.ctypes.data, out.pos.ctypes.data)
		out.has_result = has.astype(bool)
		for g in range(ng.value):
			m, H, W, cap, ncat, blk, nb = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_void_p(), ctypes.c_uint64()
			lib.tp_frames_group(h, g, ctypes.byref(m), ctypes.byref(H), ctypes.byref(W), ctypes.byref(cap), ctypes.byref(ncat), ctypes.byref(blk), ctypes.byref(nb))
			# the fields of the packed block (comm.packed_block_layout with the catalogue flags and the extras), as read-only views of the
			# page-locked block: offsets written out here -- this loop is the host's share of a batch (it was 1.0 ms of a 7.7 ms call
			# through packed_block_layout / unpack_block and one ctypes array TYPE per block size)
			mm, Hh, Ww, cc = m.value, H.value, W.value, cap.value
			P = Hh * Ww
			host = _view_bytes(blk.value, nb.value)
			grp, off = {}, 0
			for name, shape, dtype, size in (('lc', (5, mm, T), 'float64', 40 * mm * T), ('contamination', (mm,), 'float64', 8 * mm), ('status', (mm,), 'int32', 4 * mm),
				('flags', (mm,), 'int32', 4 * mm), ('mask', (mm, Hh, Ww), 'uint8', mm * P), ('cat_in_mask', (cc,), 'uint8', cc),
				('sumimage', (mm, Hh, Ww), 'float64', 8 * mm * P), ('diagnostics', (mm, 10), 'float64', 80 * mm)):
				grp[name] = host[off:off + size].view(dtype).reshape(shape)
				off = -(-(off + size) // 256) * 256
			if off != nb.value:
				raise RuntimeError('the packed block of the native engine does not have the layout of comm.packed_block_layout')
			offs = np.empty(mm + 1, dtype='int64')
			cat_ids = np.empty(max(ncat.value, 0), dtype='int64')
			tids = np.empty(mm, dtype='int64')
			lib.tp_frames_group_lists(h, g, offs.ctypes.data, cat_ids.ctypes.data, tids.ctypes.data)
			grp.update(cat_offsets=offs, cat_starid=cat_ids, target_starid=tids)
			out.groups.append(grp)
		if ne.value:
			k = ne.value
			tgt, code, a, b, txt = (np.empty(k, dtype='int32') for _ in range(5))
			val = np.empty(k, dtype='float64')
			lib.tp_frames_events(h, tgt.ctypes.data, code.ctypes.data, a.ctypes.data, b.ctypes.data, val.ctypes.data, txt.ctypes.data)
			from .plugins import _MASK_EXCEPTIONS
			for e in range(k):
				i, c = int(tgt[e]), int(code[e])
				if c == 5:
					msg = 'RuntimeError: ' + _MASK_EXCEPTIONS[int(a[e])]
				elif c == 10:
					msg = 'ERROR: Device pass failed for a %dx%d stamp: %s' % (a[e], b[e], lib.tp_frames_text(h, int(txt[e])).decode())
				elif c == 11:
					msg = 'ERROR: Device pass failed while the light curves were copied: ' + lib.tp_frames_text(h, int(txt[e])).decode()
				else:
					msg = _EVENT_TEXT[c]
				out.errors.setdefault(i, []).append(msg)
				if c == 7:
					out.edge_flux[i] = float(val[e])
		return out


class FramesEngine(object):
	"""
	The native job engine of the batched drop-in entry (``tp_frames_*``, ``csrc/frames.cpp``): ``slots`` jobs in flight, each
	driven by a worker thread of the library on three streams of its own.  ``submit`` copies the batch's host arrays and
	returns at once; ``FramesJob.collect`` waits.  One engine per context (:meth:`of`), closed with it.
	"""
	def __init__(self, ctx, slots=4):
		self.ctx, self.lib, self.slots = ctx, ctx.lib, int(slots)
		h = ctypes.c_void_p()
		rc = self.lib.tp_frames_engine_create(ctx.device, self.slots, ctypes.byref(h))
		if rc != 0:
			raise TessphotError(rc, (self.lib.tp_last_error(None) or b'').decode())
		self.handle = h.value
		hbm = ctypes.c_uint64()
		self.lib.tp_frames_engine_info(self.handle, None, None, ctypes.byref(hbm))
		self.hbm_bytes = hbm.value

	#: jobs in flight the engine of a context is made for -- the most that pays (five measure the same as four, and every slot is five
	#: streams of the process's two dozen hardware queues; aperture_frames_pipelined clamps to it); the engine
	#: is made ONCE with all of them: remaking it for more slots would strand the jobs and results that point at the old one
	MAX_SLOTS = 4

	@classmethod
	def of(cls, ctx, slots=4):
		"""The engine of ``ctx`` (made on first use with :attr:`MAX_SLOTS` slots and kept until the context closes)."""
		if slots > cls.MAX_SLOTS:
			raise ValueError('the frames engine has %d slots' % cls.MAX_SLOTS)
		eng = ctx.__dict__.get('_frames_engine')
		if eng is None or eng.handle is None:
			eng = ctx.__dict__['_frames_engine'] = cls(ctx, slots=cls.MAX_SLOTS)
		return eng

	def catalog(self, catalog):
		"""The catalogue of a region copied into the library and binned (once per run over the region: the pipelined entry)."""
		return _NativeCatalog(self.lib, catalog)

	def submit(self, stack, targets, catalog, time, quality, settings=None, datasource='ffi', budget_share=1.0):
		from . import stamps as st
		from .plugins import load_settings, mag2flux
		settings = load_settings() if settings is None else settings
		tmag_limit = settings.getfloat('haloswitch', 'tmag_limit')
		flux_limit = settings.getfloat('haloswitch', 'flux_limit')
		sid = np.ascontiguousarray(targets['starid'], dtype='int64')
		tm = np.ascontiguousarray(targets['tmag'], dtype='float64')
		row = np.ascontiguousarray(targets['row'], dtype='float64')
		col = np.ascontiguousarray(targets['column'], dtype='float64')
		n = len(sid)
		first, valid = st.default_stamps(row, col, tm, stack.limits)
		first = np.ascontiguousarray(first, dtype='int64')
		valid8 = np.ascontiguousarray(valid, dtype='uint8')
		attempts = np.where(tm < 6, 10, 5).astype('int32')                    # photometry.py:70-73 (stamps.retry_limit)
		bright = (tm <= tmag_limit) & (not datasource.startswith('tpf:'))      # photometry.py:146
		budget_flux = np.where(bright, flux_limit * mag2flux(tm), np.nan).astype('float64')
		t = np.ascontiguousarray(time, dtype='float64')
		q = np.ascontiguousarray(quality, dtype='int32')
		if len(t) != stack.n_cad or len(q) != stack.n_cad:
			raise ValueError('time and quality must have one entry per frame of the stack')
		# an FFI target's sum image is a crop of the region's (BasePhotometry.py:1001-1006); a postage-stamp target ('tpf:...') sums its
		# own stamp (:1007-1019): no region sum image is handed over and every pass forms the stamps' own (tp_sumimage)
		sumimage = None if datasource.startswith('tpf:') else stack.sumimage_for(q)
		tm_stacks = None if sumimage is None else stack.time_major()
		sdesc = _lib.tp_frames_stack(stack.dev['images'].ptr, stack.dev['images_err'].ptr, stack.dev['backgrounds'].ptr,
			stack.n_cad, stack.n_rows, stack.n_cols, stack.row0, stack.col0, None if sumimage is None else sumimage.ptr,
			None if tm_stacks is None else tm_stacks[0]['images'].ptr, None if tm_stacks is None else tm_stacks[0]['images_err'].ptr,
			None if tm_stacks is None else tm_stacks[0]['backgrounds'].ptr, 0 if tm_stacks is None else tm_stacks[1])
		budget = float(os.environ.get('TESSPHOT_FRAMES_BUDGET_GB', 0)) * 1e9 or self.hbm_bytes / 4.0
		h = ctypes.c_void_p()
		rc = self.lib.tp_frames_submit(self.handle, ctypes.byref(sdesc), catalog.handle, n, sid.ctypes.data, tm.ctypes.data, row.ctypes.data, col.ctypes.data,
			first.ctypes.data, valid8.ctypes.data, attempts.ctypes.data, budget_flux.ctypes.data, t.ctypes.data, q.ctypes.data, budget * budget_share, ctypes.byref(h))
		if rc != 0:
			raise TessphotError(rc, (self.lib.tp_last_error(None) or b'').decode())
		return FramesJob(self, h.value, n, stack.n_cad, (stack, catalog, sumimage))

	def close(self):
		h, self.handle = self.handle, None
		if h:
			self.lib.tp_frames_engine_destroy(h)


def aperture_frames(ctx, stack, targets, catalog, time, quality, settings=None, cadence_s=1800, datasource='ffi', engine='native'):
	"""
	``AperturePhotometry.do_photometry`` INCLUDING its stamp-resize loop (photometry.py:75-170) for every target of a CCD region
	held in a :class:`FrameStack`: round after round, the targets still in play are grouped by stamp size, their stamps are cut
	out of the frames on the device, one fused pass per group produces sum image, mask and light curve, and the edge flags
	decide -- with the plugin's own rules (:mod:`photometry_amd.stamps`) -- who is finished, who gets a bigger stamp, and who
	gives up ("Too many stamp resizes." / "Stamp resize hit limit. Haloswitch quick break.").

	``targets``: dict of arrays ``starid, tmag, row, column`` (CCD positions, float64); ``catalog``: the same columns for every
	star of the region.  Returns a :class:`FramesResult` (columnar; ``result[i]`` is the per-target dict: ``status, stamp,
	stamp_resizes, errors`` and, unless the target ended in an error before the extraction, ``mask, sumimage, flux, flux_err,
	flux_background, pos_centroid, contamination, skip_targets, diagnostics``).  The common case -- the mask does not touch an
	edge and nothing is logged -- is decided for a whole group with array operations; only the targets that resize, warn or fail
	take the per-target path.

	``engine``: ``'native'`` (default) -- the rounds are driven by a worker thread of the library (``csrc/frames.cpp``,
	:class:`FramesEngine`): the host submits the batch and collects it; ``'python'`` -- the same rounds as a Python generator
	(:func:`_frames_job`, the implementation the native engine is held to in ``tests/test_gpu_resize.py``).
	"""
	if engine == 'python':
		job = _frames_job(ctx, stack, targets, catalog, time, quality, settings, cadence_s, datasource, [ctx] + ctx.side_contexts(2))
		while True:
			try:
				next(job)
			except StopIteration as done:
				return done.value
	eng = FramesEngine.of(ctx)
	return eng.submit(stack, targets, eng.catalog(catalog), time, quality, settings=settings, datasource=datasource).collect()


def aperture_frames_pipelined(ctx, stack, batches, catalog, time, quality, settings=None, cadence_s=1800, datasource='ffi', in_flight=4, engine='native'):
	"""
	:func:`aperture_frames` over consecutive batches of targets of one CCD region (``batches``: an iterable of ``targets`` dicts),
	``in_flight`` of them at a time, each on its own streams: the rounds of a batch are a strict chain -- queue the passes, wait,
	decide, queue the next round -- whose later links are a few latency-bound passes over the resized stamps that leave most of
	the chip idle, so the first round of the next batch runs under them and the host decides one batch while the device works on
	the other.  Yields one :class:`FramesResult` per batch, in order; every batch gives what a call of its own would.
	With the native engine (default) every batch is a job of :class:`FramesEngine`: submitted, driven by its own worker thread of
	the library, collected in order -- the host touches a batch twice.
	"""
	if engine != 'python':
		from collections import deque
		# four streams per slot (three of the engine's pool and a copy stream): beyond ~24 streams in the process its hardware queues
		# are oversubscribed and time-sliced -- measured in round 5: 3.8-4.2 x 10^5 targets/s with four or five jobs in flight, 2.4-2.9 x 10^5 with six
		in_flight = max(1, min(int(in_flight), FramesEngine.MAX_SLOTS))
		eng = FramesEngine.of(ctx, slots=in_flight)
		cat = eng.catalog(catalog)
		# results are yielded in the order of the batches, but a slot is handed on as soon as ANY job is done: the jobs of a run take
		# 5 - 15 ms each (three to ten groups, one to five rounds), and waiting for the oldest left finished ones holding their slots
		queue, finished, serial, next_out = deque(), {}, 0, 0
		for targets in batches:
			while len(queue) >= in_flight:
				# (at most in_flight results wait for an older batch: beyond that the oldest job is waited for)
				k = 0 if len(finished) >= in_flight else next((k for k, (_, job) in enumerate(queue) if job.done()), 0)
				s, job = queue[k]
				del queue[k]
				finished[s] = job.collect()
				while next_out in finished:
					yield finished.pop(next_out)
					next_out += 1
			queue.append((serial, eng.submit(stack, targets, cat, time, quality, settings=settings, datasource=datasource, budget_share=1.0 / in_flight)))
			serial += 1
		while queue:
			s, job = queue.popleft()
			finished[s] = job.collect()
			while next_out in finished:
				yield finished.pop(next_out)
				next_out += 1
		return
	catalog = {k: np.asarray(v) for k, v in catalog.items()}
	cat_index = _CatalogIndex(catalog)
	every = [ctx] + ctx.side_contexts(3 * in_flight - 1)
	free_slots = list(range(in_flight))[::-1]
	source = iter(batches)
	running, finished, order, exhausted = [], {}, 0, False   # running: [serial, slot, generator]
	next_out = 0
	while True:
		while not exhausted and free_slots:
			try:
				targets = next(source)
			except StopIteration:
				exhausted = True
				break
			slot = free_slots.pop()
			job = _frames_job(ctx, stack, targets, catalog, time, quality, settings, cadence_s, datasource, every[3 * slot:3 * slot + 3], cat_index, 1.0 / in_flight)
			running.append([order, slot, job])
			order += 1
			try:
				next(job)   # the first round of the new batch is queued before the older ones are waited for
			except StopIteration as done:
				serial, slot, _ = running.pop()
				finished[serial] = done.value
				free_slots.append(slot)
		if not running and exhausted:
			break
		for entry in list(running):
			try:
				next(entry[2])
			except StopIteration as done:
				running.remove(entry)
				finished[entry[0]] = done.value
				free_slots.append(entry[1])
		while next_out in finished:
			yield finished.pop(next_out)
			next_out += 1
	while next_out in finished:
		yield finished.pop(next_out)
		next_out += 1


def _frames_job(ctx, stack, targets, catalog, time, quality, settings, cadence_s, datasource, streams, cat_index=None, budget_share=1.0):
	"""The rounds of :func:`aperture_frames` as a generator: it yields wherever the host would wait for the device (after the
	passes of a round are queued; before the last light curves have arrived) and returns the :class:`FramesResult`.  ``streams``:
	the contexts of this job (the first one takes the large groups); ``cat_index``: a prebuilt index of ``catalog``."""
	from . import stamps as st
	from .plugins import load_settings, mag2flux, mask_outcome
	from ._lib import TessphotError
	settings = load_settings() if settings is None else settings
	tmag_limit = settings.getfloat('haloswitch', 'tmag_limit')
	flux_limit = settings.getfloat('haloswitch', 'flux_limit')
	n = len(targets['starid'])
	catalog = {k: np.asarray(v) for k, v in catalog.items()}
	time = np.asarray(time, dtype='float64')
	quality = np.asarray(quality, dtype='int32')
	tmags = np.asarray(targets['tmag'], dtype='float64')
	out = FramesResult(n)
	full_sumimage = None if datasource.startswith('tpf:') else stack.sumimage_for(quality)   # (BasePhotometry.py:1001-1019: crop for FFI targets, own sum for stamps)
	log = {}
	def logger_of(i):
		if i not in log:
			log[i] = _Messages()
		return log[i]
	first, valid = st.default_stamps(targets['row'], targets['column'], targets['tmag'], stack.limits)
	cur = np.asarray(first, dtype='int64').copy()
	for i in np.flatnonzero(~valid): # BasePhotometry.py:671-672: the constructor raises -> STATUS.ERROR through tessphot
		out.status[i] = 2
		out.errors[int(i)] = ['ValueError: Invalid stamp selected']
		out.stamp[i] = (-1, -2, -1, -2)
	cat_index = _CatalogIndex(catalog) if cat_index is None else cat_index
	attempts_left = np.where(tmags < 6, 10, 5).astype('int64')   # photometry.py:70-73 (stamps.retry_limit)
	active = np.flatnonzero(valid)
	edge_bits = sum(bit for _name, bit, _idx, _sign in st.SIDES)

	def finish(i, status):
		out.status[i] = int(status)
		out.stamp[i] = cur[i]
		if i in log and log[i].items:
			out.errors[i] = out.errors.get(i, []) + log[i].items
			log[i].items = []

	from . import comm as tpcomm
	events, pending = [], []
	if '_hbm_bytes' not in ctx.__dict__:
		ctx.__dict__['_hbm_bytes'] = ctx.info()['hbm_bytes']
	budget = (float(os.environ.get('TESSPHOT_FRAMES_BUDGET_GB', 0)) * 1e9 or ctx.__dict__['_hbm_bytes'] / 4.0) * budget_share   # (jobs in flight share it)
	while len(active):
		heights, widths = cur[active, 1] - cur[active, 0], cur[active, 3] - cur[active, 2]
		keys = heights * 100000 + widths
		still = []
		# the groups of this round (targets that share a stamp size), cut into parts whose cubes and output blocks fit a budget of
		# device memory: a full CCD has tens of thousands of targets beside 65 GB of resident frame stacks, and an out-of-memory
		# error would cost every target of the part its result (a quarter of the HBM by default, TESSPHOT_FRAMES_BUDGET_GB)
		jobs = []
		for key in np.unique(keys):
			idx_all = active[keys == key]
			H, W = int(key // 100000), int(key % 100000)
			per_target = 3 * H * W * round_up(stack.n_cad, 32) * 4 + 5 * stack.n_cad * 8 + H * W * 9 + 256
			nmax = max(1, int(budget // per_target))
			for a0 in range(0, len(idx_all), nmax):
				jobs.append((idx_all[a0:a0 + nmax], H, W, per_target * len(idx_all[a0:a0 + nmax])))
		parts, acc = [[]], 0
		for idx_j, H, W, nbytes_j in jobs:
			if parts[-1] and acc + nbytes_j > budget:
				parts.append([])
				acc = 0
			parts[-1].append((idx_j, H, W))
			acc += nbytes_j
		for part in parts:
			# ---- the device passes of all groups of this round are queued first, round-robin on a few streams (a group of a few large
			# stamps is a latency-bound pass of ~1 ms that hides under the pass of the 15 x 15 group), each pass ending with ONE
			# download of its packed output block into page-locked memory; then the results are decided group by group
			launched = []
			for gi, (idx, H, W) in enumerate(part):
				g = streams[gi % len(streams)] if len(idx) < 256 or gi == 0 else streams[0]
				cubes = host = None
				try:
					if H * W > 32767:
						raise TessphotError(1, f'a {H}x{W} stamp is beyond the 32 767 pixels of the mask builder')
					# the stamps are cut while the host selects the catalogue stars of the group
					cut = stack.cut_lazy(g, cur[idx], H, W)
					cubes = {k: cut[k] for k in stack.names}
					cat_offsets, cat_arrays = _catalogs_of_stamps(cat_index, cur[idx])
					scene = _GroupScene(stack, time, quality, cadence_s, cur[idx], cat_offsets, cat_arrays, targets, idx)
					batch = ApertureBatch(g, scene, cubes=cut)
					work = ApertureWork(g, batch, packed=True, cat_capacity=max(int(cat_offsets[-1]), 1), extras=True)
					# a small group is a latency-bound pass: the three stand-alone kernels spread A1 and A6 over the chip where the fused launch
					# gives each target one wavefront (8 targets of 25 x 25: 1.03 against 1.47 ms; bit-identical outputs)
					# the sum images: crops of the region's (BasePhotometry.py:1001-1006), as in the native engine
					if full_sumimage is not None:
						engine.crop_sumimage(g, full_sumimage, cut['_stamps'], H, W, stack.row0, stack.col0, out=work.sumimage)
					aperture_step(g, batch, work, fused=len(idx) >= 1024, sumimage_given=full_sumimage is not None)
					aperture_diagnostics(g, batch, work)
					host = g.pinned_block(work.block.nbytes)
					# two copies: what the decisions of this round read (flags, masks, sum images ...: everything behind the light curves in
					# the block) first, with an event; the light curves (97 % of the bytes) travel while the next round is decided and queued
					lc_bytes = work.block_layout['contamination'][0]
					g.download_async(host, device_view(g, work.block.ptr + lc_bytes, (work.block.nbytes - lc_bytes,), 'uint8'), host_offset=lc_bytes)
					ev = events.pop() if events else g.event()
					g.record(ev)
					g.download_async(host, device_view(g, work.block.ptr, (lc_bytes,), 'uint8'))
					launched.append((g, idx, H, W, scene, cat_offsets, cat_arrays, cubes, batch, work, host, ev))
				except TessphotError as e:
					# e.g. a stamp beyond 32 767 pixels (the mask builder's signed 16-bit labels): Halo territory upstream
					try:
						g.sync()
					except TessphotError:
						pass
					if cubes is not None:
						for c in cubes.values():
							c.free()
					if host is not None:
						g.pinned_release(host)
					for i in idx:
						logger_of(int(i)).error('Device pass failed for a %dx%d stamp: %s', H, W, str(e))
						finish(int(i), 2)
			yield   # the passes of this part run: the caller may serve another job meanwhile
			failed = None
			for (g, idx, H, W, scene, cat_offsets, cat_arrays, cubes, batch, work, host, ev) in launched:
				try:
					g.event_sync(ev)      # the small part of the group's block is on the host (its light curves may still be on their way)
				except TessphotError as e:   # a device error surfaces here: every group of the round is lost
					failed = e
				events.append(ev)
				for c in cubes.values():
					c.free()
				pending.append((g, batch, work))   # alive until the light curves have arrived
				if failed is not None:
					g.pinned_release(host)
					for i in idx:
						logger_of(int(i)).error('Device pass failed for a %dx%d stamp: %s', H, W, str(failed))
						finish(int(i), 2)
					continue
				res = tpcomm.unpack_block(host.array[:work.block.nbytes], work.block_layout)
				out._pinned.append((g, host))
				grp = dict(res, cat_offsets=cat_offsets, cat_starid=cat_arrays['starid'], target_starid=scene.target_starid)
				gid = len(out.groups)
				out.groups.append(grp)
				attempts_left[idx] -= 1
				flags = res['flags'].astype('int64')
				kind = flags >> 8
				# ---- the common case, for the whole group at once: nothing to log, no edge touched -> the attempt stands
				simple = ((flags & (1 | 32 | edge_bits)) == 0) & (kind == 0)
				done = idx[simple]
				out.status[done] = res['status'][simple]
				out.stamp[done] = cur[done]
				out.has_result[done] = True
				out.group[done] = gid
				out.pos[done] = np.flatnonzero(simple)
				for i in done:   # what an earlier round logged for the target (it went on with a bigger stamp) stays in its details
					if int(i) in log and log[int(i)].items:
						finish(int(i), int(out.status[i]))
				# ---- the others, one by one with the plugin's rules
				for j in np.flatnonzero(~simple):
					i = int(idx[j])
					fl = int(flags[j])
					try:
						if mask_outcome(fl, logger_of(i)) == 'error':
							finish(i, 2)
							continue
					except RuntimeError as e: # an uncaught exception of the reference's plugin -> STATUS.ERROR (tessphot.py:37-49)
						out.errors[i] = out.errors.get(i, []) + ['RuntimeError: ' + str(e)]
						finish(i, 2)
						continue
					wanted = st.edge_requests(fl)
					if wanted:
						before = tuple(int(v) for v in cur[i])
						new = st.moved(before, stack.limits, **wanted)
						if new == before:
							logger_of(i).warning("Could not resize stamp any further.")
						else:
							out.stamp_resizes[i] += 1
							cur[i] = new
							mask = res['mask'][j].astype(bool)
							bright = tmags[i] <= tmag_limit and not datasource.startswith('tpf:')
							stuck = st.quick_break_flux(res['sumimage'][j], mask, before, new, wanted) if bright else None
							if stuck is not None and stuck > flux_limit * mag2flux(tmags[i]):
								logger_of(i).error('Stamp resize hit limit. Haloswitch quick break.')
								out.edge_flux[i] = stuck
								finish(i, 2)
							elif attempts_left[i] == 0:
								logger_of(i).error('Too many stamp resizes.')
								finish(i, 2)
							else:
								still.append(i)
							continue
					# this attempt stands
					if fl >> 8 == 6:
						logger_of(i).error("No targets in mask.")
					out.has_result[i] = True
					out.group[i], out.pos[i] = gid, j
					finish(i, int(res['status'][j]))
		active = np.asarray(sorted(still), dtype='int64')
	# the light curves of every round have arrived
	marks = []
	for g in streams:
		ev = events.pop() if events else g.event()
		g.record(ev)
		marks.append((g, ev))
	yield
	for g, ev in marks:
		try:
			g.event_sync(ev)
		except TessphotError as e:   # a copy of light curves failed: nothing that was extracted can be trusted
			for i in np.flatnonzero(out.has_result):
				out.has_result[i] = False
				out.errors[int(i)] = out.errors.get(int(i), []) + ['ERROR: Device pass failed while the light curves were copied: ' + str(e)]
				out.status[i] = 2
	for g, batch, _work in pending:
		batch.release_host()
	return out


#--------------------------------------------------------------------------------------------------
# PSF photometry of many targets of one CCD region (the batched counterparts of the LinPSF / PSF plugins)
#--------------------------------------------------------------------------------------------------
class PSFFramesResult(object):
	"""
	Columnar results of :func:`linpsf_frames` / :func:`psf_frames`: ``status`` int32 (values of ``STATUS``), ``stamp`` int64
	``(n, 4)``, ``flux`` / ``flux_err`` float64 ``(n, T)``, ``contamination`` float64 (LinPSF), ``pos_centroid`` float64
	``(n, T, 2)`` as (row, column) (PSF), ``errors``: target index -> messages.  ``result[i]`` is the per-target dict.
	"""
	def __init__(self, n, T, method):
		self.n, self.method = n, method
		self.status = np.zeros(n, dtype='int32')
		self.stamp = np.zeros((n, 4), dtype='int64')
		if method == 'linpsf':
			# (filled group by group; the rows no group covers -- invalid stamps -- get their NaN at the end, finish(): two arrays of
			# 21 MB per 2 000 targets are not written twice)
			self.flux, self.flux_err = np.empty((n, T)), np.empty((n, T))
			self._covered = np.zeros(n, dtype=bool)
		else:
			self.flux = np.full((n, T), np.nan)
			self.flux_err = np.full((n, T), np.nan)
			self._covered = None
		self.contamination = np.full(n, np.nan)
		self.pos_centroid = np.full((n, T, 2), np.nan) if method == 'psf' else None
		self.errors = {}

	def finish(self):
		"""NaN for the rows no group has written (see ``__init__``)."""
		if self._covered is not None:
			rest = ~self._covered
			if rest.any():
				self.flux[rest] = np.nan
				self.flux_err[rest] = np.nan
			self._covered = None
		return self

	def __len__(self):
		return self.n

	def __getitem__(self, i):
		i = int(i)
		r = {'status': int(self.status[i]), 'stamp': tuple(int(v) for v in self.stamp[i]), 'errors': list(self.errors.get(i, [])),
			'flux': self.flux[i], 'flux_err': self.flux_err[i]}
		if self.method == 'linpsf':
			r['contamination'] = float(self.contamination[i])
		else:
			r['pos_centroid'] = self.pos_centroid[i]
		return r


def _psf_frame_groups(stack, targets):
	"""Default stamps of the targets (BasePhotometry.py:541-564, no resizing: neither PSF plugin resizes) grouped by size."""
	from . import stamps as st
	first, valid = st.default_stamps(targets['row'], targets['column'], targets['tmag'], stack.limits)
	cur = np.asarray(first, dtype='int64')
	active = np.flatnonzero(valid)
	keys = (cur[active, 1] - cur[active, 0]) * 100000 + (cur[active, 3] - cur[active, 2])
	groups = [(int(key // 100000), int(key % 100000), active[keys == key]) for key in np.unique(keys)]
	return cur, valid, groups


def _jitter32(jitter, T):
	"""Per-cadence (column, row) shifts as the plugins apply them (``catalog_attime``: float32 additions)."""
	if jitter is None:
		return np.zeros(T, dtype='float32'), np.zeros(T, dtype='float32')
	j = np.asarray(jitter, dtype='float64')
	return j[:, 0].astype('float32'), j[:, 1].astype('float32')


def linpsf_frames(ctx, stack, targets, catalog, time, quality, prf_model, jitter=None, cutoff_radius=5):
	"""
	``LinPSFPhotometry.do_photometry`` (linpsf_photometry.py:79-219) for every target of a CCD region held in a
	:class:`FrameStack`: default stamps grouped by size and cut on the device, the stars fitted beside each target selected as
	the plugin does (:93-104), their positions at every cadence = catalogue position + ``jitter[k]`` (``(T, 2)`` column / row
	shifts: what ``catalog_attime`` returns for a translation), one ``tp_linpsf_prf`` + ``tp_linpsf_fit`` per group.
	Status and messages follow the plugin: ERROR "All target flux values are NaN.", WARNING "High contamination" above 0.1.
	Returns a :class:`PSFFramesResult`.
	"""
	from . import psf as hpsf
	n, T = len(targets['starid']), stack.n_cad
	catalog = {k: np.asarray(v) for k, v in catalog.items()}
	out = PSFFramesResult(n, T, 'linpsf')
	cur, valid, groups = _psf_frame_groups(stack, targets)
	out.stamp[:] = cur
	for i in np.flatnonzero(~valid):
		out.status[i] = 2
		out.errors[int(i)] = ['ValueError: Invalid stamp selected']
		out.stamp[i] = (-1, -2, -1, -2)
	cat_index = _CatalogIndex(catalog)
	jc, jr = _jitter32(jitter, T)
	d_jc, d_jr = ctx.array(jc), ctx.array(jr)
	base_coef, tx, ty = ctx.array(prf_model.base_coef), ctx.array(prf_model.tx), ctx.array(prf_model.ty)
	for H, W, idx in groups:
		cat_offsets, cat = _catalogs_of_stamps(cat_index, cur[idx])
		sel, star_offsets, target_index = hpsf.select_stars(cat, cat_offsets, np.asarray(targets['starid'], dtype='int64')[idx])
		# positions = catalogue position + the cadence's shift, summed in float32 like the plugin's catalogue: formed on the device
		# (on the host the two arrays -- 75 MB for 2 000 targets -- were most of this entry's time)
		pos_row = engine.star_positions(ctx, ctx.array(np.ascontiguousarray(cat['row_stamp'][sel], dtype='float32')), d_jr)
		pos_col = engine.star_positions(ctx, ctx.array(np.ascontiguousarray(cat['column_stamp'][sel], dtype='float32')), d_jc)
		cube = engine.cut_stamps(ctx, stack.dev['images'], ctx.array(cur[idx].astype('int32')), H, W, stack.row0, stack.col0)
		try:
			coef = engine.linpsf_prf(ctx, base_coef, ctx.array(prf_model.weights(cur[idx])))
			fit = engine.linpsf_fit(ctx, cube, coef, tx, ty, ctx.array(star_offsets), ctx.array(target_index), pos_row, pos_col,
				max(int(np.diff(star_offsets).max()), 1), cutoff_radius=cutoff_radius)
			res = fit.to_host(('contamination', 'status'))
			if len(idx) == n and np.array_equal(idx, np.arange(n)) and fit.flux.shape == out.flux.shape:
				# one group holds every target in order (the usual case: default stamps of one size): straight into the result arrays
				fit.flux.to_host(out=out.flux)
				fit.flux_err.to_host(out=out.flux_err)
			else:
				out.flux[idx] = fit.flux.to_host()[:, :T]
				out.flux_err[idx] = fit.flux_err.to_host()[:, :T]
			out._covered[idx] = True
		finally:
			ctx.sync()
			cube.free()
		out.contamination[idx] = res['contamination']
		# linpsf_photometry.py:198-200, 214-219: ERROR when every flux is NaN, WARNING above 10 % contamination (NaN compares false)
		failed = res['status'] == 2
		with np.errstate(invalid='ignore'):
			high = ~failed & (res['contamination'] > 0.1)
		out.status[idx] = np.where(failed, 2, np.where(high, 3, 1))
		for i in idx[failed]:
			out.errors[int(i)] = ['All target flux values are NaN.']
		for i in idx[high]:
			out.errors[int(i)] = ['High contamination']
	return out.finish()


def psf_frames(ctx, stack, targets, catalog, time, quality, prf_model, readnoise=10.0, gain=100.0, n_readout=720, cutoff_radius=5):
	"""
	``PSFPhotometry.do_photometry`` (psf_photometry.py:111-196) for every target of a CCD region held in a :class:`FrameStack`:
	default stamps grouped by size, images and backgrounds cut on the device, per target the up to five stars the plugin fits
	(:117-130) with their catalogue positions and fluxes as the first guess, the 3 x 3 minimum aperture of the finite sum-image
	pixels (:29-41), one ``tp_psf_fit`` per group (the cadences of a target are a warm-start chain, the targets run side by side).
	Returns a :class:`PSFFramesResult` (status OK, as the plugin: NaN fluxes are only logged there, :190-194).
	"""
	from .plugins import psf_star_selection, mag2flux
	n, T = len(targets['starid']), stack.n_cad
	catalog = {k: np.asarray(v) for k, v in catalog.items()}
	out = PSFFramesResult(n, T, 'psf')
	cur, valid, groups = _psf_frame_groups(stack, targets)
	out.stamp[:] = cur
	for i in np.flatnonzero(~valid):
		out.status[i] = 2
		out.errors[int(i)] = ['ValueError: Invalid stamp selected']
		out.stamp[i] = (-1, -2, -1, -2)
	cat_index = _CatalogIndex(catalog)
	base_coef, tx, ty = ctx.array(prf_model.base_coef), ctx.array(prf_model.tx), ctx.array(prf_model.ty)
	full_sumimage = stack.sumimage_for(quality)
	trow, tcol, ttmag = (np.asarray(targets[k], dtype='float64') for k in ('row', 'column', 'tmag'))
	for H, W, idx in groups:
		cat_offsets, cat = _catalogs_of_stamps(cat_index, cur[idx])
		stamps_dev = ctx.array(cur[idx].astype('int32'))
		images = engine.cut_stamps(ctx, stack.dev['images'], stamps_dev, H, W, stack.row0, stack.col0)
		backgrounds = engine.cut_stamps(ctx, stack.dev['backgrounds'], stamps_dev, H, W, stack.row0, stack.col0)
		try:
			# bit 1 of the aperture image (BasePhotometry.py:1033-1074) from the sum image -- for an FFI target a crop of the region's (:1001-1006)
			finite = np.isfinite(engine.crop_sumimage(ctx, full_sumimage, stamps_dev, H, W, stack.row0, stack.col0).to_host()).reshape(len(idx), H, W)
			offsets = [0]
			params0 = []
			mini = np.zeros((len(idx), H, W), dtype='uint8')
			for j, i in enumerate(idx):
				lo, hi = int(cat_offsets[j]), int(cat_offsets[j + 1])
				r1, c1 = cur[i, 0], cur[i, 2]
				sel = psf_star_selection(cat['row_stamp'][lo:hi], cat['column_stamp'][lo:hi], cat['tmag'][lo:hi], trow[i] - r1, tcol[i] - c1, ttmag[i])
				params0.append(np.column_stack((np.asarray(cat['row_stamp'][lo:hi][sel], dtype='float64'), np.asarray(cat['column_stamp'][lo:hi][sel], dtype='float64'),
					mag2flux(np.asarray(cat['tmag'][lo:hi][sel], dtype='float64')))))
				offsets.append(offsets[-1] + len(sel))
				cols, rows = np.meshgrid(np.arange(c1 + 1, cur[i, 3] + 1, dtype='int32'), np.arange(r1 + 1, cur[i, 1] + 1, dtype='int32'))
				mini[j] = (np.abs(cols - tcol[i] - 1) <= 1) & (np.abs(rows - trow[i] - 1) <= 1) & finite[j]
			coef = engine.linpsf_prf(ctx, base_coef, ctx.array(prf_model.weights(cur[idx])))
			res = engine.psf_fit(ctx, images, backgrounds, coef, tx, ty, ctx.array(np.asarray(offsets, dtype='int64')), ctx.array(np.concatenate(params0, axis=0)),
				ctx.array(mini), variance_floor=n_readout * readnoise**2 / gain**2, cutoff_radius=cutoff_radius)
			flux, flux_err = res['flux'].to_host(), res['flux_err'].to_host()
			crow, ccol = res['centroid_row'].to_host(), res['centroid_col'].to_host()
		finally:
			ctx.sync()
			images.free()
			backgrounds.free()
		out.flux[idx] = flux[:, :T]
		out.flux_err[idx] = flux_err[:, :T]
		out.pos_centroid[idx, :, 0] = crow[:, :T]
		out.pos_centroid[idx, :, 1] = ccol[:, :T]
		out.status[idx] = 1
	return out
