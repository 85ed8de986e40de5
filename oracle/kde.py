# -*- coding: utf-8 -*-
"""
ORACLE (test infrastructure only) -- Gaussian KDE as used by K2P2 (A2).

The reference calls statsmodels 0.13.2 (requirements.txt:29), which is NOT under
/root/reference and is not installable in this image:

* ``select_bandwidth(flux_cut, bw='scott', kernel='gau')``  (k2p2v2.py:410)
* ``KDEUnivariate(flux_cut).fit(kernel='gau', bw=..., fft=True, gridsize=100)`` (k2p2v2.py:414-415)
* ``kernel.support[np.argmax(kernel.density)]`` (k2p2v2.py:420)
* ``kernel.evaluate(x)`` (k2p2v2.py:419)

This module restates the published algorithms of that version
(``statsmodels/nonparametric/bandwidths.py: _select_sigma, bw_scott, select_bandwidth``;
``kde.py: kdensityfft, KDEUnivariate.evaluate``; ``kdetools.py: forrt, revrt,
silverman_transform``; ``linbin.pyx: fast_linbin``; ``kernels.py: Gaussian``).
statsmodels itself cannot be run here, so the restatement is pinned INDEPENDENTLY in
``tests/test_oracle_pins.py``: ``evaluate`` against ``scipy.stats.gaussian_kde`` with the same kernel width (1e-11), the
FFT density's argmax against the argmax of the direct Gaussian sum on the same 128-point grid (the only thing the reference
reads from it, k2p2v2.py:420), linear binning against its defining properties and a hand-computed case, ``_select_sigma`` /
Scott's rule against numpy's percentile definition.  What stays unpinned is only agreement with statsmodels where that
package deviates from the published algorithm in a way these properties cannot see.
"""

import numpy as np
from scipy.stats import scoreatpercentile


def select_sigma(x):
	"""bandwidths._select_sigma: min(std(ddof=1), IQR/1.349), or std if IQR == 0."""
	normalize = 1.349
	IQR = (scoreatpercentile(x, 75) - scoreatpercentile(x, 25)) / normalize
	std_dev = np.std(x, axis=0, ddof=1)
	if IQR > 0:
		return np.minimum(std_dev, IQR)
	return std_dev


def bw_scott(x):
	"""bandwidths.bw_scott: 1.059 * A * n**(-1/5)."""
	A = select_sigma(x)
	n = len(x)
	return 1.059 * A * n ** (-0.2)


def normal_reference_constant():
	"""
	kernels.CustomKernel.normal_reference_constant for ``kernels.Gaussian`` (order 2, L2Norm = 1/(2 sqrt(pi)), second moment 1):
	``C = sqrt(pi) * 2!**3 * L2Norm / (2 * 2 * 4! * 1**2); C = 2 * C**(1/5)`` = 1.0592238...
	"""
	C = np.pi**(.5) * 2.0**3 * (1.0/(2.0*np.sqrt(np.pi)))
	C /= (2 * 2 * 24.0 * 1.0**2)
	return 2*C**(1.0/(2*2+1))


def bw_normal_reference(x):
	"""bandwidths.bw_normal_reference (the default of ``KDEUnivariate.fit``): C * A * n**(-1/5)."""
	A = select_sigma(x)
	n = len(x)
	return normal_reference_constant() * A * n ** (-0.2)


def select_bandwidth(x, bw='scott', kernel='gau'):
	"""bandwidths.select_bandwidth (raises RuntimeError on zero bandwidth)."""
	if bw.lower() not in ('scott', 'normal_reference'):
		raise ValueError("only 'scott' and 'normal_reference' are restated")
	bandwidth = bw_scott(np.asarray(x)) if bw.lower() == 'scott' else bw_normal_reference(np.asarray(x))
	if np.any(bandwidth == 0):
		raise RuntimeError("Selected KDE bandwidth is 0. Cannot estimate density. "
			"Either provide the bandwidth during initialization or use an alternative method.")
	return bandwidth


def fast_linbin(X, a, b, M):
	"""linbin.pyx fast_linbin (Fan & Marron linear binning), including its ``li > 1`` guard."""
	M = int(M)
	gcnts = np.zeros(M, dtype='float64')
	delta = (b - a) / (M - 1)
	lxi = (np.asarray(X, dtype='float64') - a) / delta
	li = lxi.astype(int)
	rem = lxi - li
	for i in range(len(lxi)):
		li_i = li[i]
		if li_i > 1 and li_i < M:
			gcnts[li_i] = gcnts[li_i] + 1 - rem[i]
			gcnts[li_i+1] = gcnts[li_i+1] + rem[i]
		# (statsmodels' ``if li_i > M: gcnts[M] += 1`` branch cannot trigger: b = max + cut*bw)
	return gcnts


def fast_linbin_vec(X, a, b, M):
	"""``fast_linbin`` without the Python loop (``np.add.at`` accumulates in input order too: same sums, same rounding)."""
	M = int(M)
	gcnts = np.zeros(M + 1, dtype='float64')
	delta = (b - a) / (M - 1)
	lxi = (np.asarray(X, dtype='float64') - a) / delta
	li = lxi.astype(int)
	rem = lxi - li
	ok = (li > 1) & (li < M)
	idx = np.empty(2 * int(np.sum(ok)), dtype=int)
	w = np.empty(idx.size)
	idx[0::2] = li[ok]; idx[1::2] = li[ok] + 1
	w[0::2] = 1 - rem[ok]; w[1::2] = rem[ok]
	np.add.at(gcnts, idx, w)
	return gcnts[:M]


def fast_linbin_fixed(X, a, b, M, bits=40):
	"""
	``fast_linbin`` with the weights accumulated as ``bits``-bit fixed-point integers: ``q = rint(rem * 2**bits)`` goes to the
	upper grid point, ``2**bits - q`` to the lower one.  Integer sums do not depend on the order of the samples -- this is the
	arithmetic of the device kernel (``csrc/radial.hip``), whose parallel accumulation must be reproducible; it deviates from
	the float accumulation by at most ``n * 2**-(bits + 1)`` per grid point.
	"""
	M = int(M)
	delta = (b - a) / (M - 1)
	lxi = (np.asarray(X, dtype='float64') - a) / delta
	li = lxi.astype(int)
	rem = lxi - li
	ok = (li > 1) & (li < M)
	q = np.rint(rem[ok] * 2.0**bits).astype(np.uint64)
	g = np.zeros(M + 1, dtype=np.uint64)
	np.add.at(g, li[ok], np.uint64(2**bits) - q)
	np.add.at(g, li[ok] + 1, q)
	return g[:M].astype('float64') * (1.0 / 2.0**bits)


def forrt(X, m=None):
	"""kdetools.forrt: RFFT in Munro (1976) FORRT ordering."""
	if m is None:
		m = len(X)
	y = np.fft.rfft(X, m) / m
	return np.r_[y.real, y[1:-1].imag]


def revrt(X, m=None):
	"""kdetools.revrt: inverse of forrt."""
	if m is None:
		m = len(X)
	i = int(m // 2 + 1)
	y = X[:i] + np.r_[0, X[i:], 0] * 1j
	return np.fft.irfft(y) * m


def silverman_transform(bw, M, RANGE):
	"""kdetools.silverman_transform: FFT of the Gaussian kernel (Silverman AS 176)."""
	J = np.arange(M / 2 + 1)
	FAC1 = 2 * (np.pi * bw / RANGE)**2
	JFAC = J**2 * FAC1
	BC = 1 - 1. / 3 * (J * 1. / M * np.pi)**2
	FAC = np.exp(-JFAC) / BC
	kern_est = np.r_[FAC, FAC[1:-1]]
	return kern_est


def kdensityfft(x, bw, gridsize=100, cut=3, binning='float'):
	"""kde.kdensityfft for a user-given bandwidth.  Returns ``(density, grid, bw)``.  ``binning='fixed'``: :func:`fast_linbin_fixed`."""
	x = np.asarray(x, dtype='float64')
	bw = float(bw)
	nobs = len(x)
	gridsize = 2 ** np.ceil(np.log2(gridsize)) # round to next power of 2 -> 128.0
	a = np.min(x) - cut * bw
	b = np.max(x) + cut * bw
	grid, delta = np.linspace(a, b, int(gridsize), retstep=True)
	RANGE = b - a
	if binning == 'fixed':
		binned = fast_linbin_fixed(x, a, b, gridsize) / (delta * nobs)
	else:
		binned = (fast_linbin if nobs < 2000 else fast_linbin_vec)(x, a, b, gridsize) / (delta * nobs)
	y = forrt(binned)
	zstar = silverman_transform(bw, gridsize, RANGE) * y
	f = revrt(zstar)
	return f, grid, bw


class KDE(object):
	"""Minimal ``KDEUnivariate`` (kde.py): ``fit`` (fft path) and ``evaluate`` (direct sum)."""

	def __init__(self, endog):
		self.endog = np.ascontiguousarray(endog, dtype='float64')

	def fit(self, kernel='gau', bw=None, fft=True, gridsize=None, cut=3, binning='float'):
		if kernel != 'gau' or not fft:
			raise NotImplementedError
		if bw is None:
			bw = 'normal_reference'     # the default of KDEUnivariate.fit
		if isinstance(bw, str):
			bw = select_bandwidth(self.endog, bw)
		if gridsize is None:
			gridsize = max(len(self.endog), 512.0)
		self.density, self.support, self.bw = kdensityfft(self.endog, bw, gridsize=gridsize, cut=cut, binning=binning)
		return self

	def evaluate(self, point):
		"""kernels.CustomKernel.density with the Gaussian shape
		``0.3989422804014327*exp(-x**2/2)``: ``1/(h*n) * sum(K((xs - x)/h))``."""
		xs = self.endog
		n = len(xs)
		h = self.bw
		point = np.atleast_1d(np.asarray(point, dtype='float64'))
		z = (xs[:, None] - point[None, :]) / h
		return 1. / (h * n) * np.sum(0.3989422804014327 * np.exp(-z**2 / 2.0), axis=0)
