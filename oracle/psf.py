# -*- coding: utf-8 -*-
"""
ORACLE (test infrastructure only) -- P1 / P2: the PRF model.

Restates ``photometry/psf.py``: ``PSF.__init__`` (:35-119, the inverse-distance blend
of the SPOC PRF samples, normalisation and the interpolating bicubic spline) and
``PSF.integrate_to_image`` (:122-148).

Third-party arithmetic: ``scipy.interpolate.RectBivariateSpline`` (psf.py:119) is FITPACK
``regrid`` with ``s=0`` and its ``.integral`` (psf.py:146) is FITPACK ``dblint`` /
``fpintb`` (scipy is pinned to 1.7.3 in requirements.txt:9 and is not under
/root/reference; FITPACK itself has been unchanged for decades).  ``fpintb``/``dblint`` are
restated here from the published FITPACK algorithm (Gaffney's formulae for the indefinite
integral of a B-spline) and pinned against the scipy installed in this image by the tests.

The box integral is separable:  ``integral = sum_ij wx[i] * wy[j] * c[i, j]`` with
``wx = fpintb(tx, xa, xb)``, ``wy = fpintb(ty, ya, yb)``; limits clip to the knot span.
"""

import numpy as np
from scipy.interpolate import RectBivariateSpline

MINIMUM_PRF_WEIGHT = 1e-6 #: psf.py:74


def fpintb(t, nk1, x, y):
	"""
	FITPACK ``fpintb``: integrals of the normalised B-splines N_{j,k+1} of degree
	``k = len(t) - nk1 - 1`` over ``[x, y]``.  Returns ``bint`` of length ``nk1``.
	"""
	t = np.asarray(t, dtype='float64')
	n = len(t)
	k1 = n - nk1
	ak = float(k1)
	k = k1 - 1
	bint = np.zeros(nk1, dtype='float64')
	a, b = x, y
	neg = False
	if a == b:
		return bint
	if a > b:
		a, b = y, x
		neg = True
	# 1-based FITPACK indices are kept below and shifted by -1 on access.
	if a < t[k1-1]:
		a = t[k1-1]
	if b > t[nk1]:
		b = t[nk1]
	if a > b:
		return bint
	l = k1
	l0 = l + 1
	arg = a
	ia = 0
	aint = np.zeros(6)
	h = np.zeros(6)
	h1 = np.zeros(6)
	for it in (1, 2):
		while not (arg < t[l0-1] or l == nk1):
			l = l0
			l0 = l + 1
		aint[:k1] = 0.0
		aint[0] = (arg - t[l-1])/(t[l] - t[l-1])
		h1[0] = 1.0
		for j in range(1, k+1):
			h[0] = 0.0
			for i in range(1, j+1):
				li = l + i
				lj = li - j
				f = h1[i-1]/(t[li-1] - t[lj-1])
				h[i-1] = h[i-1] + f*(t[li-1] - arg)
				h[i] = f*(arg - t[lj-1])
			j1 = j + 1
			for i in range(1, j1+1):
				li = l + i
				lj = li - j1
				aint[i-1] = aint[i-1] + h[i-1]*(arg - t[lj-1])/(t[li-1] - t[lj-1])
				h1[i-1] = h[i-1]
		if it == 1:
			lk = l - k
			ia = lk
			for i in range(1, k1+1):
				bint[lk-1] = -aint[i-1]
				lk += 1
			arg = b
	lk = l - k
	ib = lk - 1
	for i in range(1, k1+1):
		bint[lk-1] = bint[lk-1] + aint[i-1]
		lk += 1
	if ib >= ia:
		for i in range(ia, ib+1):
			bint[i-1] = bint[i-1] + 1.0
	f = 1.0/ak
	for i in range(1, nk1+1):
		j = i + k1
		bint[i-1] = bint[i-1]*(t[j-1] - t[i-1])*f
	if neg:
		bint = -bint
	return bint


def dblint(tx, ty, c, kx, ky, xb, xe, yb, ye):
	"""FITPACK ``dblint``: double integral of the bivariate spline (tx, ty, c)."""
	nkx1 = len(tx) - kx - 1
	nky1 = len(ty) - ky - 1
	wx = fpintb(tx, nkx1, xb, xe)
	wy = fpintb(ty, nky1, yb, ye)
	C = np.asarray(c, dtype='float64').reshape(nkx1, nky1)
	res = 0.0
	for i in range(nkx1):
		if wx[i] == 0.0:
			continue
		res += float(np.sum(wx[i] * wy * C[i]))
	return res


class PSF(object):
	"""
	Restatement of ``photometry.psf.PSF`` working from in-memory PRF samples instead of
	the SPOC ``.mat`` files (those are git-LFS objects that are absent here).

	Parameters:
		prf_values: ``(n_hdu, xdim, ydim)`` PRF sample images (``mat['values']``).
		prf_ccd_column, prf_ccd_row: ``(n_hdu,)`` CCD positions of the samples.
		PRFx, PRFy: sub-pixel grid coordinates (``prfColumn``, ``prfRow``).
		stamp: ``(row_min, row_max, col_min, col_max)``.
	"""

	def __init__(self, prf_values, prf_ccd_column, prf_ccd_row, PRFx, PRFy, stamp):
		self.stamp = stamp
		self.shape = (int(stamp[1] - stamp[0]), int(stamp[3] - stamp[2])) # psf.py:65
		self.ref_column = 0.5*(stamp[3] + stamp[2]) # psf.py:77
		self.ref_row = 0.5*(stamp[1] + stamp[0]) # psf.py:78
		PRFx = np.asarray(PRFx, dtype='float64').flatten()
		PRFy = np.asarray(PRFy, dtype='float64').flatten()
		cdelt1p = np.median(np.diff(PRFx))
		cdelt2p = np.median(np.diff(PRFy))
		prf = np.zeros((len(PRFx), len(PRFy)), dtype='float64')
		for i in range(len(prf_values)): # psf.py:101-113
			prfWeight = np.sqrt((self.ref_column - float(prf_ccd_column[i]))**2 + (self.ref_row - float(prf_ccd_row[i]))**2)
			prfWeight = max(prfWeight, MINIMUM_PRF_WEIGHT)
			prf += prf_values[i] / prfWeight
		prf /= (np.nansum(prf) * cdelt1p * cdelt2p) # psf.py:116
		self.prf = prf
		self.splineInterpolation = RectBivariateSpline(PRFx, PRFy, prf) # psf.py:119
		self.tx, self.ty, c = self.splineInterpolation.tck
		self.coeffs = np.asarray(c).reshape(len(self.tx) - 4, len(self.ty) - 4)

	@classmethod
	def from_spline(cls, spline, shape):
		"""Build directly from a ``RectBivariateSpline`` (golden tests)."""
		self = cls.__new__(cls)
		self.shape = tuple(shape)
		self.stamp = (0, shape[0], 0, shape[1])
		self.splineInterpolation = spline
		self.tx, self.ty, c = spline.tck
		self.coeffs = np.asarray(c).reshape(len(self.tx) - 4, len(self.ty) - 4)
		return self

	def integrate_to_image(self, params, cutoff_radius=5):
		"""psf.py:122-148, with the FITPACK box integral written out (separable form)."""
		H, W = self.shape
		img = np.zeros(self.shape, dtype='float64')
		nkx1 = len(self.tx) - 4
		nky1 = len(self.ty) - 4
		for star in params:
			star_row, star_column, star_flux = star[0], star[1], star[2]
			# basis integrals for every pixel column / row of the stamp:
			WX = np.array([fpintb(self.tx, nkx1, (j - star_column) - 0.5, (j - star_column) + 0.5) for j in range(W)])
			WY = np.array([fpintb(self.ty, nky1, (i - star_row) - 0.5, (i - star_row) + 0.5) for i in range(H)])
			full = WY @ (WX @ self.coeffs).T # (H, W): sum_ab wx_a c_ab wy_b
			for i in range(H):
				for j in range(W):
					if cutoff_radius is None or np.sqrt((j-star_column)**2 + (i-star_row)**2) < cutoff_radius:
						img[i, j] += star_flux * full[i, j]
		return img

	def integrate_to_image_scipy(self, params, cutoff_radius=5):
		"""Literal psf.py:136-146 (calls scipy's FITPACK ``integral``); slow, used for pinning."""
		img = np.zeros(self.shape, dtype='float64')
		for i in range(self.shape[0]):
			for j in range(self.shape[1]):
				for star in params:
					star_row = star[0]
					star_column = star[1]
					if cutoff_radius is None or np.sqrt((j-star_column)**2 + (i-star_row)**2) < cutoff_radius:
						star_flux = star[2]
						column_cen = j - star_column
						row_cen = i - star_row
						img[i, j] += star_flux * self.splineInterpolation.integral(column_cen-0.5, column_cen+0.5, row_cen-0.5, row_cen+0.5)
		return img


def synthetic_prf(n_hdu_side=5, sigma=0.9, nsub=9, halfwidth=6.5, seed=0):
	"""
	Synthetic stand-in for a SPOC ``*-characterized-prf.mat`` (psf.py:81-104): ``n_hdu_side**2``
	PRF samples on a grid over the CCD, each a ``(117, 117)`` image sampled at 9 sub-pixels per
	pixel over +-6.5 px; slightly different widths/ellipticities per sample so that the blend
	(psf.py:101-113) is exercised.

	Returns dict(values, ccdColumn, ccdRow, prfColumn, prfRow).
	"""
	rng = np.random.default_rng(seed)
	n = int(round(2*halfwidth*nsub))
	x = (np.arange(n) - (n-1)/2) / nsub
	n_hdu = n_hdu_side**2
	cc, rr = np.meshgrid(np.linspace(45, 2092, n_hdu_side), np.linspace(1, 2048, n_hdu_side))
	values = np.empty((n_hdu, n, n))
	for i in range(n_hdu):
		sx = sigma * (1 + 0.08*rng.standard_normal())
		sy = sigma * (1 + 0.08*rng.standard_normal())
		gx = np.exp(-0.5*(x/sx)**2)
		gy = np.exp(-0.5*(x/sy)**2)
		values[i] = np.outer(gx, gy) + 1e-4*np.outer(np.exp(-0.5*(x/(3*sx))**2), np.exp(-0.5*(x/(3*sy))**2))
	return {'values': values, 'ccdColumn': cc.flatten(), 'ccdRow': rr.flatten(), 'prfColumn': x.copy(), 'prfRow': x.copy()}
