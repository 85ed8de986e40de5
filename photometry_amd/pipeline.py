# -*- coding: utf-8 -*-
"""
Batched per-target pipelines over a stamp cube resident in HBM.

``run_aperture``: A1 sum image -> A2..A5b K2P2 masks + A7 contamination -> A6 extraction,
i.e. ``AperturePhotometry.do_photometry`` (photometry/AperturePhotometry/photometry.py:44-257)
for every target of the batch at once.
"""

import numpy as np
from . import engine
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
		q = np.asarray(scene.quality, dtype='int32')
		self.quality = ctx.array(q)
		self.time = ctx.array(np.asarray(scene.time, dtype='float64'))
		self.stamps = ctx.array(np.asarray(scene.stamps, dtype='int32'))
		self.n_targets = self.images.n_targets
		self.n_cad = self.images.n_cad
		self.height, self.width = self.images.height, self.images.width
		# catalog (ragged) and target metadata for the mask kernel
		c = scene.catalog
		self.cat_offsets = ctx.array(np.asarray(scene.cat_offsets, dtype='int64'))
		self.cat_starid = ctx.array(np.asarray(c['starid'], dtype='int64'))
		self.cat_tmag = ctx.array(np.asarray(c['tmag'], dtype='float32'))
		self.cat_row = ctx.array(np.asarray(c['row'], dtype='float32'))
		self.cat_column = ctx.array(np.asarray(c['column'], dtype='float32'))
		self.cat_row_stamp = ctx.array(np.asarray(c['row_stamp'], dtype='float32'))
		self.cat_column_stamp = ctx.array(np.asarray(c['column_stamp'], dtype='float32'))
		self.target_pos_row = ctx.array(np.asarray(scene.target_pos_row, dtype='float64'))
		self.target_pos_column = ctx.array(np.asarray(scene.target_pos_column, dtype='float64'))
		self.target_tmag = ctx.array(np.asarray(scene.target_tmag, dtype='float64'))
		self.target_starid = ctx.array(np.asarray(scene.target_starid, dtype='int64'))
		ap = getattr(scene, 'aperture', None)
		if ap is None:
			ap = np.ones((self.n_targets, self.height, self.width), dtype='int32')
		self.aperture = ctx.array(np.asarray(ap, dtype='int32'))

	#: per-target arrays (sliced by :meth:`chunk`); the flat catalogue arrays are shared because the
	#: CSR offsets are absolute
	_PER_TARGET = ('images', 'images_err', 'backgrounds', 'stamps', 'target_pos_row', 'target_pos_column',
		'target_tmag', 'target_starid', 'aperture')

	def chunk(self, start, count):
		"""Non-owning view of the targets ``[start, start+count)`` (same HBM)."""
		v = ApertureBatch.__new__(ApertureBatch)
		v.__dict__.update(self.__dict__)
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
	block of a rank) -- out of ONE allocation (``self.block``), so that the per-step gather of a multi-GPU run is a
	single message.
	"""

	def __init__(self, ctx, batch, packed=False):
		Nt, T, H, W = batch.n_targets, batch.n_cad, batch.height, batch.width
		self.sumimage = ctx.empty((Nt, H, W), 'float64')
		self.block = None
		if packed:
			sizes = [5 * Nt * T * 8, Nt * 8, Nt * 4, Nt * 4, Nt * H * W]
			offs = [0]
			for n in sizes:
				offs.append(round_up(offs[-1] + n, 256))
			self.block = ctx.zeros((offs[-1],), 'uint8')
			b = self.block
			self.lc = engine.LightCurves(ctx, Nt, T, block=device_view(ctx, b.ptr + offs[0], (5, Nt, T), 'float64', base=b))
			self.contamination = device_view(ctx, b.ptr + offs[1], (Nt,), 'float64', base=b)
			self.status = device_view(ctx, b.ptr + offs[2], (Nt,), 'int32', base=b)
			self.flags = device_view(ctx, b.ptr + offs[3], (Nt,), 'int32', base=b)
			self.mask = device_view(ctx, b.ptr + offs[4], (Nt, H, W), 'uint8', base=b)
			self.block_layout = dict(zip(('lc', 'contamination', 'status', 'flags', 'mask'), offs[:5]))
		else:
			self.mask = ctx.zeros((Nt, H, W), 'uint8')
			self.status = ctx.zeros((Nt,), 'int32')
			self.flags = ctx.zeros((Nt,), 'int32')
			self.contamination = ctx.zeros((Nt,), 'float64')
			self.lc = engine.LightCurves(ctx, Nt, T)
		self.diag = ctx.zeros((Nt, 8), 'float64')
		self.cat_in_mask = ctx.zeros((max(int(batch.scene.cat_offsets[-1]), 1),), 'uint8')
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


def aperture_step(ctx, batch, work, masks_from=None, fused=True):
	"""
	One pass of the aperture hot path over the batch; everything stays in HBM.

	``fused``: one launch with a wavefront per target (``tp_aperture_photometry``); ``False`` runs the three
	stand-alone kernels A1, K2P2, A6 back to back (bit-identical outputs).
	``masks_from``: optional ``(mask uint8 DeviceArray, status int32 DeviceArray)`` to bypass the
	on-device K2P2 (used by tests that inject the oracle's masks; implies the stand-alone kernels).
	"""
	subtract, backgrounds = None, batch.backgrounds
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

	def __init__(self, ctx, scene, prf_model, images=None, subtract=None):
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
		self.out = engine.LinPSFResult(ctx, scene.n_targets, len(rs), T)
		self.n_fit_stars = len(rs)


def linpsf_step(ctx, batch, cutoff_radius=5.0):
	"""One pass of the LinPSF hot path: P1 table blend, P2-P4 fit + contamination."""
	engine.linpsf_prf(ctx, batch.base_coef, batch.weights, out=batch.coef)
	engine.linpsf_fit(ctx, batch.images, batch.coef, batch.tx, batch.ty, batch.star_offsets, batch.target_index,
		batch.pos_row, batch.pos_col, batch.max_stars, cutoff_radius=cutoff_radius, subtract=batch.subtract, out=batch.out)
	return batch.out
