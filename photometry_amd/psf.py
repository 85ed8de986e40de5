# -*- coding: utf-8 -*-
"""
Host side of the PRF model (P1): what ``photometry.psf.PSF.__init__`` does per target
(photometry/psf.py:35-119), re-arranged for a batch.

The reference reads the SPOC PRF ``.mat`` file of the camera/CCD (``prfStruct`` with ``values``,
``ccdRow``, ``ccdColumn``, ``prfRow``, ``prfColumn``, psf.py:81-104), blends the PRF samples with
inverse-distance weights to the stamp centre (:101-113), normalises (:116) and fits an interpolating
bicubic spline with ``scipy.interpolate.RectBivariateSpline`` (:119) -- once per target.

The spline fit is linear in the data, therefore::

    coef(blend) = sum_i w_i * coef(PRF_i) / (sum_i w_i * nansum(PRF_i) * cdelt1p * cdelt2p)

so this class fits the samples ONCE (with the same scipy call) and the per-target table is a
25-term weighted sum done on the device (``tp_linpsf_prf``).
"""

import glob
import os
import numpy as np
from scipy.interpolate import RectBivariateSpline

MINIMUM_PRF_WEIGHT = 1e-6 #: psf.py:74


def prf_file(psf_dir, sector, camera, ccd):
	"""
	The SPOC PRF file the reference opens for (sector, camera, CCD), psf.py:47-72: sectors 1-3 use the characterisation
	that starts at sector 1, later sectors the one that starts at sector 4; same input checks and messages.
	``psf_dir`` is the directory that holds ``start_s0001`` / ``start_s0004`` (upstream: ``photometry/data/psf``).
	"""
	if sector < 1:
		raise ValueError("Sector number must be greater than zero")
	if camera not in (1, 2, 3, 4):
		raise ValueError("Camera must be 1, 2, 3 or 4.")
	if ccd not in (1, 2, 3, 4):
		raise ValueError("CCD must be 1, 2, 3 or 4.")
	sector_dir = 'start_s0004' if sector >= 4 else 'start_s0001'
	found = glob.glob(os.path.join(psf_dir, sector_dir, f'tess*-{camera:d}-{ccd:d}-characterized-prf.mat'))
	if not found:
		raise FileNotFoundError(f"no PRF file for camera {camera}, CCD {ccd} under {os.path.join(psf_dir, sector_dir)}")
	return found[0]


class PRFModel(object):
	"""
	Parameters:
		values: ``(n_hdu, xdim, ydim)`` PRF sample images (``mat['values']``; first axis = prfColumn).
		ccd_column, ccd_row: ``(n_hdu,)`` CCD positions of the samples (``ccdColumn``, ``ccdRow``).
		prf_x, prf_y: sub-pixel sample coordinates (``prfColumn``, ``prfRow``); the SPOC files hold 9 samples per pixel.
	"""

	def __init__(self, values, ccd_column, ccd_row, prf_x, prf_y):
		values = np.asarray(values, dtype='float64')
		self.prf_x = np.asarray(prf_x, dtype='float64').flatten()
		self.prf_y = np.asarray(prf_y, dtype='float64').flatten()
		if values.ndim != 3 or values.shape[1:] != (len(self.prf_x), len(self.prf_y)):
			raise ValueError("PRF values must be (n_hdu, len(prf_x), len(prf_y))")
		if not np.all(np.isfinite(values)):
			raise ValueError("non-finite PRF samples are not supported")
		for g in (self.prf_x, self.prf_y):
			if len(g) < 4 or not np.all(np.diff(g) > 0):
				raise ValueError("PRF sample coordinates must be strictly increasing, at least 4 per axis (RectBivariateSpline's own rule)")
		# (any spacing is taken: the SPOC layout -- evenly spaced, 9 samples per pixel -- runs on the fast kernels, anything else on the
		# general ones with the FITPACK box integral; the library decides from the knots, tp_linpsf_fit in include/tessphot_hip.h)
		self.n_hdu = values.shape[0]
		self.ccd_column = np.asarray(ccd_column, dtype='float64').flatten()
		self.ccd_row = np.asarray(ccd_row, dtype='float64').flatten()
		self.cdelt1p = np.median(np.diff(self.prf_x)) # psf.py:94-95
		self.cdelt2p = np.median(np.diff(self.prf_y))
		self.sums = np.array([np.nansum(v) for v in values])
		coefs = []
		for i in range(self.n_hdu):
			spl = RectBivariateSpline(self.prf_x, self.prf_y, values[i]) # psf.py:119
			tx, ty, c = spl.tck
			coefs.append(np.asarray(c, dtype='float64'))
		self.tx, self.ty = np.asarray(tx, dtype='float64'), np.asarray(ty, dtype='float64')
		self.n, self.ny = len(self.tx) - 4, len(self.ty) - 4   # (different lengths: the any-grid kernels, tp_linpsf_fit_xy / tp_psf_fit_xy)
		self.base_coef = np.ascontiguousarray(np.stack(coefs)) # (n_hdu, n*ny)

	@classmethod
	def from_mat(cls, path):
		"""
		A SPOC ``*-characterized-prf.mat`` file, unpacked like psf.py:81-104: ``prfStruct`` with one entry per PRF sample;
		``prfColumn`` / ``prfRow`` are taken from the first entry ("assuming they are all the same"), ``values`` /
		``ccdColumn`` / ``ccdRow`` from every entry.
		"""
		from scipy.io import loadmat
		mat = loadmat(path)['prfStruct']
		prf_x = np.asarray(mat['prfColumn'][0][0], dtype='float64').flatten()
		prf_y = np.asarray(mat['prfRow'][0][0], dtype='float64').flatten()
		n_hdu = len(mat['values'][0])
		values = np.stack([np.asarray(mat['values'][0][i], dtype='float64') for i in range(n_hdu)])
		ccd_column = np.array([float(np.asarray(mat['ccdColumn'][0][i]).ravel()[0]) for i in range(n_hdu)])
		ccd_row = np.array([float(np.asarray(mat['ccdRow'][0][i]).ravel()[0]) for i in range(n_hdu)])
		self = cls(values, ccd_column, ccd_row, prf_x, prf_y)
		self.path = path
		return self

	@classmethod
	def for_ccd(cls, psf_dir, sector, camera, ccd):
		"""The model of a CCD from the reference's data directory layout (psf.py:66-72)."""
		return cls.from_mat(prf_file(psf_dir, sector, camera, ccd))

	@classmethod
	def from_spline(cls, spline):
		"""A single, already normalised spline (tests with the golden vectors)."""
		self = cls.__new__(cls)
		tx, ty, c = spline.tck
		self.tx, self.ty = np.asarray(tx, dtype='float64'), np.asarray(ty, dtype='float64')
		self.n, self.ny = len(self.tx) - 4, len(self.ty) - 4
		self.n_hdu = 1
		self.base_coef = np.ascontiguousarray(np.asarray(c, dtype='float64')[None, :])
		self.ccd_column = self.ccd_row = np.zeros(1)
		self.sums = np.ones(1)
		self.cdelt1p = self.cdelt2p = 1.0
		self._unit = True
		return self

	def weights(self, stamps):
		"""
		Per-target blend weights divided by the normalisation (psf.py:77-78, 101-116).
		``stamps``: ``(Nt, 4)`` = (row_min, row_max, col_min, col_max).  Returns ``(Nt, n_hdu)`` float64.
		"""
		stamps = np.asarray(stamps, dtype='float64')
		if getattr(self, '_unit', False):
			return np.ones((stamps.shape[0], 1))
		ref_column = 0.5*(stamps[:, 3] + stamps[:, 2])
		ref_row = 0.5*(stamps[:, 1] + stamps[:, 0])
		w = np.sqrt((ref_column[:, None] - self.ccd_column[None, :])**2 + (ref_row[:, None] - self.ccd_row[None, :])**2)
		w = 1.0 / np.maximum(w, MINIMUM_PRF_WEIGHT)
		norm = (w * self.sums[None, :]).sum(axis=1) * self.cdelt1p * self.cdelt2p
		return w / norm[:, None]


def select_stars(catalog, cat_offsets, target_starid):
	"""
	linpsf_photometry.py:87-104 for a batch: stars closer than 5 pixels to the main target and not
	more than 5 magnitudes fainter.  Returns ``(sel bool over the flat catalog, star_offsets (Nt+1),
	target_index (Nt))`` for the compacted list of fitted stars.
	"""
	off = np.asarray(cat_offsets, dtype='int64')
	Nt = len(off) - 1
	counts = np.diff(off)
	tgt = np.repeat(np.arange(Nt), counts)
	starid = np.asarray(catalog['starid'])
	is_main = starid == np.asarray(target_starid)[tgt]
	main_idx = np.full(Nt, -1, dtype='int64')
	main_idx[tgt[is_main]] = np.flatnonzero(is_main)
	if np.any(main_idx < 0):
		raise ValueError("main target missing from its catalog")
	rs, cs, tm = (np.asarray(catalog[k]) for k in ('row_stamp', 'column_stamp', 'tmag'))
	dist = np.sqrt((rs[main_idx][tgt] - rs)**2 + (cs[main_idx][tgt] - cs)**2)
	sel = (dist < 5) & (tm[main_idx][tgt] - tm > -5)
	fit_counts = np.bincount(tgt[sel], minlength=Nt)
	star_offsets = np.concatenate(([0], np.cumsum(fit_counts))).astype('int64')
	# index of the main target inside its fitted stars
	pos_in_fit = np.cumsum(sel) - 1
	target_index = (pos_in_fit[main_idx] - star_offsets[:-1]).astype('int32')
	return sel, star_offsets, target_index
