# -*- coding: utf-8 -*-
"""
Stamp sources: where the plugin classes get their per-target inputs from.

The reference's ``BasePhotometry.__init__`` (photometry/BasePhotometry.py:100-486) pulls these
from HDF5 / SQLite / SPICE / FITS files through h5py, astropy and spiceypy -- none of which
exists in this image, and that I/O is outside the hot path.  A
``StampSource`` hands over the same information as plain numpy arrays:

* light-curve time base: ``time, timecorr, cadenceno, quality`` (BasePhotometry.py:234-241)
* the target: ``starid, tmag, target_pos_row, target_pos_column`` (CCD pixels, :440-447)
* a cut-out function ``cutout(starid, stamp) -> dict(images, images_err, backgrounds)`` returning
  float32 ``(rows, cols, times)`` cubes exactly like ``_load_cube`` (:720-751)
* the catalogue of the stamp (``starid, tmag, row, column``; :1094-1181) and optionally per-cadence
  positions (``catalog_attime``, :1224-1258)
* frame limits (``max_stamp``), ``n_readout``, sector / camera / ccd, the pixel-flag free
  ``backgrounds_pixels_used`` image.
"""

import numpy as np


class MemoryStampSource(object):
	"""
	Source backed by in-memory full "frames" ``(R, C, T)`` (a small synthetic CCD region) or by
	per-target stamp cubes that cannot grow.

	Parameters:
		frames: dict with float32 ``(R, C, T)`` arrays ``images, images_err, backgrounds`` covering
			CCD rows ``[row0, row0+R)`` and columns ``[col0, col0+C)``.
		row0, col0: CCD coordinates of frame pixel (0, 0).
		time, timecorr, cadenceno, quality: light-curve base arrays ``(T,)``.
		catalog: dict of arrays ``starid, tmag, row, column`` (CCD coordinates) of ALL stars of the region.
		jitter: optional ``(T, 2)`` per-cadence (column, row) shift used by ``catalog_attime``.
		frames may also hold ``pixel_flags`` (uint8 ``(R, C, T)``: the ``pixel_flags/%04d`` images of the prepare stage).
		backgrounds_pixels_used: optional bool ``(R, C)`` image of the prepare stage (BasePhotometry.py:1052-1061).
	"""

	def __init__(self, frames, row0, col0, time, timecorr, cadenceno, quality, catalog, sector=1, camera=1, ccd=1,
		cadence=1800, n_readout=720, jitter=None, prf=None, targets=None, backgrounds_pixels_used=None):
		self.frames = {k: np.asarray(v, dtype='uint8' if k == 'pixel_flags' else 'float32') for k, v in frames.items()}
		self.backgrounds_pixels_used = None if backgrounds_pixels_used is None else np.asarray(backgrounds_pixels_used, dtype=bool)
		R, C, T = self.frames['images'].shape
		self.row0, self.col0 = int(row0), int(col0)
		self.max_stamp = (self.row0, self.row0 + R, self.col0, self.col0 + C)
		self.time = np.asarray(time, dtype='float64')
		self.timecorr = np.asarray(timecorr, dtype='float64')
		self.cadenceno = np.asarray(cadenceno, dtype='int32')
		self.quality = np.asarray(quality, dtype='int32')
		self.catalog = {k: np.asarray(v) for k, v in catalog.items()}
		self.sector, self.camera, self.ccd, self.cadence, self.n_readout = sector, camera, ccd, cadence, n_readout
		self.jitter = None if jitter is None else np.asarray(jitter, dtype='float64')
		self.prf = prf
		#: optional float64 positions of the main targets (the reference projects ra/dec through the WCS,
		#: BasePhotometry.py:440-447; the catalogue columns are only float32)
		self.targets = targets

	def target(self, starid):
		if self.targets is not None:
			idx = np.flatnonzero(np.asarray(self.targets['starid']) == starid)
			if len(idx):
				i = idx[0]
				return {'starid': int(starid), 'tmag': float(self.targets['tmag'][i]),
					'row': float(self.targets['row'][i]), 'column': float(self.targets['column'][i])}
		idx = np.flatnonzero(self.catalog['starid'] == starid)
		if len(idx) == 0:
			raise RuntimeError(f"Star could not be found in catalog: {starid:d}") # BasePhotometry.py:414
		i = idx[0]
		return {'starid': int(starid), 'tmag': float(self.catalog['tmag'][i]),
			'row': float(self.catalog['row'][i]), 'column': float(self.catalog['column'][i])}

	def cutout(self, stamp):
		"""float32 ``(rows, cols, T)`` cubes of the stamp (row_min, row_max, col_min, col_max)."""
		r1, r2, c1, c2 = stamp
		sl = (slice(r1 - self.row0, r2 - self.row0), slice(c1 - self.col0, c2 - self.col0))
		return {k: np.ascontiguousarray(v[sl[0], sl[1], :]) for k, v in self.frames.items()}

	def catalog_in_stamp(self, stamp, buffer_size=5):
		"""Stars inside the stamp plus a 5-pixel buffer (catalog_sqlite_search_footprint, buffer_size=5)."""
		r1, r2, c1, c2 = stamp
		row, col = self.catalog['row'], self.catalog['column']
		sel = (row >= r1 - 0.5 - buffer_size) & (row < r2 - 0.5 + buffer_size) & (col >= c1 - 0.5 - buffer_size) & (col < c2 - 0.5 + buffer_size)
		return {k: v[sel] for k, v in self.catalog.items()}


def source_from_scene(scene, i=None, margin=0):
	"""
	Build a :class:`MemoryStampSource` for target ``i`` of a ``simulate.Scene`` (the frame is just the
	target's own stamp, optionally with the cubes of a larger ``margin`` -- the stamp cannot grow
	beyond the frame, which is the fixed-cube situation of the batched engine).
	"""
	i = 0 if i is None else i
	st = scene.stamps[i]
	cat = scene.catalog_of(i)
	frames = {'images': scene.images[i], 'images_err': scene.images_err[i], 'backgrounds': scene.backgrounds[i]}
	return MemoryStampSource(frames, st[0], st[2], scene.time, scene.timecorr, np.arange(scene.n_cad) + 1, scene.quality,
		{'starid': cat['starid'], 'tmag': cat['tmag'], 'row': cat['row'], 'column': cat['column']},
		cadence=int(scene.cadence_s), jitter=scene.jitter,
		targets={'starid': scene.target_starid[i:i+1], 'tmag': scene.target_tmag[i:i+1],
			'row': scene.target_pos_row[i:i+1], 'column': scene.target_pos_column[i:i+1]})
