# -*- coding: utf-8 -*-
"""
Deterministic synthetic stamp-cube generator (host side, numpy).

The data model follows the reference's own simulator (``simulation/simulateFITS.py:338-405``:
stars with ``mag2flux`` fluxes spread by a pixel-integrated PSF, a smooth background and
Gaussian noise) and ``utilities.mag2flux`` / ``utilities.integratedGaussian``
(``photometry/utilities.py:100-149``), restricted to a per-target stamp and extended with
per-cadence jitter, NaN pixels and quality flags as laid out in SURVEY.md section 8(d).

It is used by the tests (small sizes) and as the *scene* generator of ``bench.py`` -- the
bench fills the big pixel cubes on the device (``tp_synth_fill``) from the same scene
parameters, so no 35 GB host array is ever needed.

Layout: every cube is ``(Nt, H, W, T)`` float32, i.e. per target the reference's
``(rows, cols, times)`` C-order cube with time as the fastest axis (BasePhotometry.py:732).
"""

import numpy as np
from scipy.special import erf

ZP = 20.451 #: photometry/utilities.py:134


def mag2flux(mag, zp=ZP):
	return np.clip(10**(-0.4*(mag - zp)), 0, None)


class Scene(object):
	"""Container for one batch of synthetic targets (plain attributes, numpy arrays)."""

	def subset(self, sel):
		"""Return a Scene holding only the targets ``sel`` (1-D index array / slice)."""
		sel = np.arange(self.n_targets)[sel]
		s = Scene()
		s.__dict__.update(self.__dict__)
		s.n_targets = len(sel)
		for name in ('images', 'images_err', 'backgrounds', 'raw', 'raw_err', 'stamps', 'target_pos_row', 'target_pos_column',
			'target_tmag', 'target_starid', 'aperture', 'star_params', 'sigma_psf', 'bkg_level', 'bkg_phase'):
			v = getattr(self, name, None)
			if v is not None:
				setattr(s, name, v[sel])
		# ragged catalog
		offs = [0]
		cols = {k: [] for k in self.catalog}
		for i in sel:
			a, b = self.cat_offsets[i], self.cat_offsets[i+1]
			for k in self.catalog:
				cols[k].append(self.catalog[k][a:b])
			offs.append(offs[-1] + (b - a))
		s.cat_offsets = np.asarray(offs, dtype='int64')
		s.catalog = {k: np.concatenate(v) if v else self.catalog[k][:0] for k, v in cols.items()}
		return s

	def catalog_of(self, i):
		a, b = self.cat_offsets[i], self.cat_offsets[i+1]
		return {k: v[a:b] for k, v in self.catalog.items()}


def make_scene(n_targets, n_cad, height, width, seed=0, max_neighbours=2, tmag_range=(8.0, 14.0),
	neighbour_tmag_range=(10.0, 15.0), sigma_psf=0.9, bkg_level=100.0, readnoise=10.0, cadence_s=1800.0):
	"""
	Scene parameters only (no pixel cubes): positions, magnitudes, catalog, quality, time.

	``star_params[i]`` is ``(max_neighbours+1, 3)``: (row_stamp, col_stamp, flux e/s), flux 0 for
	unused slots; slot 0 is the main target.
	"""
	rng = np.random.default_rng(seed)
	s = Scene()
	s.n_targets, s.n_cad, s.height, s.width = int(n_targets), int(n_cad), int(height), int(width)
	s.seed = seed
	s.readnoise = float(readnoise)
	Nt, T, H, W = s.n_targets, s.n_cad, s.height, s.width

	# Stamps on a fake CCD: (row_min, row_max, col_min, col_max), 0-based incl. the 44-column offset
	row0 = rng.integers(0, 2048 - H, Nt)
	col0 = rng.integers(44, 44 + 2048 - W, Nt)
	s.stamps = np.column_stack((row0, row0 + H, col0, col0 + W)).astype('int32')

	nslots = max_neighbours + 1
	sp = np.zeros((Nt, nslots, 3), dtype='float64')
	# main target: centre + U(-0.5, 0.5)
	sp[:, 0, 0] = (H - 1)/2 + rng.uniform(-0.5, 0.5, Nt)
	sp[:, 0, 1] = (W - 1)/2 + rng.uniform(-0.5, 0.5, Nt)
	tmag = rng.uniform(tmag_range[0], tmag_range[1], Nt)
	sp[:, 0, 2] = mag2flux(tmag)
	nneigh = rng.integers(0, max_neighbours + 1, Nt) if max_neighbours > 0 else np.zeros(Nt, dtype=int)
	ntmag = rng.uniform(neighbour_tmag_range[0], neighbour_tmag_range[1], (Nt, max(max_neighbours, 1)))
	nrow = rng.uniform(3, H - 3, (Nt, max(max_neighbours, 1))) if H > 6 else rng.uniform(0, H, (Nt, max(max_neighbours, 1)))
	ncol = rng.uniform(3, W - 3, (Nt, max(max_neighbours, 1))) if W > 6 else rng.uniform(0, W, (Nt, max(max_neighbours, 1)))
	for j in range(max_neighbours):
		use = nneigh > j
		sp[use, j+1, 0] = nrow[use, j]
		sp[use, j+1, 1] = ncol[use, j]
		sp[use, j+1, 2] = mag2flux(ntmag[use, j])
	s.star_params = sp
	s.sigma_psf = np.full(Nt, sigma_psf)
	s.bkg_level = bkg_level * rng.uniform(0.8, 1.2, Nt)
	s.bkg_phase = rng.uniform(0, 2*np.pi, Nt)

	s.target_tmag = tmag
	s.target_starid = (np.arange(Nt, dtype='int64') + 1) * 10
	# CCD position of main target (float64), consistent with stamp-relative position:
	s.target_pos_row = sp[:, 0, 0] + s.stamps[:, 0]
	s.target_pos_column = sp[:, 0, 1] + s.stamps[:, 2]

	# Ragged catalog (float32 columns as in BasePhotometry.py:1153-1178)
	offs = [0]
	starid, ctmag, crow, ccol, crs, ccs = [], [], [], [], [], []
	for i in range(Nt):
		n = 1 + int(nneigh[i])
		for j in range(n):
			starid.append(s.target_starid[i] + j)
			ctmag.append(tmag[i] if j == 0 else ntmag[i, j-1])
			crs.append(sp[i, j, 0])
			ccs.append(sp[i, j, 1])
			crow.append(sp[i, j, 0] + s.stamps[i, 0])
			ccol.append(sp[i, j, 1] + s.stamps[i, 2])
		offs.append(offs[-1] + n)
	s.cat_offsets = np.asarray(offs, dtype='int64')
	s.catalog = {
		'starid': np.asarray(starid, dtype='int64'),
		'tmag': np.asarray(ctmag, dtype='float32'),
		'row': np.asarray(crow, dtype='float32'),
		'column': np.asarray(ccol, dtype='float32'),
		'row_stamp': np.asarray(crs, dtype='float32'),
		'column_stamp': np.asarray(ccs, dtype='float32'),
	}

	# Time axis / quality: 2 % of the cadences carry bit 32 (Desat)
	s.time = 1325.0 + np.arange(T) * cadence_s/86400.0
	s.timecorr = np.zeros(T)
	q = np.zeros(T, dtype='int32')
	q[rng.random(T) < 0.02] = 32
	s.quality = q
	s.cadence_s = cadence_s
	# per-cadence jitter (pixels) and multiplicative variability
	s.jitter = rng.normal(0, 0.02, (T, 2))
	s.variability_seed = int(rng.integers(0, 2**31))
	return s


def _gauss_int(edges_lo, centre, sigma):
	"""Integral of a unit 1-D Gaussian over [x-0.5, x+0.5] for pixel centres ``edges_lo``."""
	d = np.sqrt(2) * sigma
	return 0.5*(erf((edges_lo - centre + 0.5)/d) - erf((edges_lo - centre - 0.5)/d))


def fill_cubes(s, nan_fraction=1e-3, with_raw=False, dtype='float32'):
	"""
	Fill ``s.images`` (background-subtracted), ``s.images_err``, ``s.backgrounds`` on the host.

	noise sigma = sqrt(signal + bkg + readnoise**2); ``nan_fraction`` of the pixel-cadences are NaN
	in both images and errors.  With ``with_raw`` also ``s.raw`` = images + backgrounds (before NaN).
	"""
	Nt, T, H, W = s.n_targets, s.n_cad, s.height, s.width
	rng = np.random.default_rng([s.seed, 12345])
	rows = np.arange(H, dtype='float64')
	cols = np.arange(W, dtype='float64')
	images = np.empty((Nt, H, W, T), dtype=dtype)
	errs = np.empty((Nt, H, W, T), dtype=dtype)
	bkgs = np.empty((Nt, H, W, T), dtype=dtype)
	raw = np.empty((Nt, H, W, T), dtype=dtype) if with_raw else None
	vrng = np.random.default_rng(s.variability_seed)
	tt = np.arange(T)
	for i in range(Nt):
		signal = np.zeros((H, W, T), dtype='float64')
		var = 1.0 + vrng.normal(0, 1e-3, T)
		for j in range(s.star_params.shape[1]):
			r0, c0, f = s.star_params[i, j]
			if f <= 0:
				continue
			gr = _gauss_int(rows[:, None], r0 + s.jitter[None, :, 1], s.sigma_psf[i]) # (H, T)
			gc = _gauss_int(cols[:, None], c0 + s.jitter[None, :, 0], s.sigma_psf[i]) # (W, T)
			fj = f * (var if j == 0 else 1.0)
			signal += gr[:, None, :] * gc[None, :, :] * fj
		bkg_t = s.bkg_level[i] * (1 + 0.05*np.sin(2*np.pi*tt/max(T, 1)*3 + s.bkg_phase[i]))
		bkg = np.broadcast_to(bkg_t[None, None, :], (H, W, T))
		sigma = np.sqrt(signal + bkg + s.readnoise**2)
		noise = rng.standard_normal((H, W, T)) * sigma
		img = signal + noise
		nanmask = rng.random((H, W, T)) < nan_fraction
		if with_raw:
			r = (img + bkg).astype(dtype)
			r[nanmask] = np.nan
			raw[i] = r
		img = img.astype(dtype)
		err = sigma.astype(dtype)
		img[nanmask] = np.nan
		err[nanmask] = np.nan
		images[i] = img
		errs[i] = err
		bkgs[i] = bkg.astype(dtype)
	s.images, s.images_err, s.backgrounds = images, errs, bkgs
	if with_raw:
		s.raw = raw
		s.raw_err = errs.copy()
	# Aperture bit image (BasePhotometry.py:1033-1061): 1 = collected; CCD outputs A-D by 1-based column
	ap = np.ones((Nt, H, W), dtype='int32')
	cgrid = (s.stamps[:, 2][:, None] + 1 + np.arange(W)[None, :])[:, None, :] * np.ones((1, H, 1), dtype='int64')
	ap[(45 <= cgrid) & (cgrid <= 556)] |= 32
	ap[(557 <= cgrid) & (cgrid <= 1068)] |= 64
	ap[(1069 <= cgrid) & (cgrid <= 1580)] |= 128
	ap[(1581 <= cgrid) & (cgrid <= 2092)] |= 256
	s.aperture = ap
	return s


def synthetic_prf(n_side=5, sigma=0.9, samples_per_pixel=9, halfwidth=6.5, seed=0):
	"""
	Synthetic stand-in for a SPOC ``*-characterized-prf.mat`` file (the real ones are git-LFS objects upstream; the
	reference reads ``prfStruct.values / ccdRow / ccdColumn / prfRow / prfColumn``, psf.py:81-104): ``n_side**2`` PRF
	samples spread over the CCD, each a 117 x 117 image at 9 samples per pixel over +-6.5 px, Gaussian
	cores of slightly different widths plus a faint broad halo so that the inverse-distance blend (psf.py:101-113)
	matters.  Returns ``dict(values, ccdColumn, ccdRow, prfColumn, prfRow)``.
	"""
	rng = np.random.default_rng(seed)
	n = int(round(2 * halfwidth * samples_per_pixel))
	grid = (np.arange(n) - (n - 1) / 2) / samples_per_pixel
	cols, rows = np.meshgrid(np.linspace(45, 2092, n_side), np.linspace(1, 2048, n_side))
	values = np.empty((n_side**2, n, n))
	for i in range(n_side**2):
		sx, sy = sigma * (1 + 0.08 * rng.standard_normal(2))
		core = np.outer(np.exp(-0.5 * (grid / sx)**2), np.exp(-0.5 * (grid / sy)**2))
		halo = np.outer(np.exp(-0.5 * (grid / (3 * sx))**2), np.exp(-0.5 * (grid / (3 * sy))**2))
		values[i] = core + 1e-4 * halo
	return {'values': values, 'ccdColumn': cols.ravel(), 'ccdRow': rows.ravel(), 'prfColumn': grid.copy(), 'prfRow': grid.copy()}
