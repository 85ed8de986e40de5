# -*- coding: utf-8 -*-
"""
ORACLE (test infrastructure only) -- 1-parameter Powell minimiser as used by K2P2 (A2).

The reference computes ``MODE = minimize(kernel_opt, max_guess, method='Powell').x``
(k2p2v2.py:418-421) with scipy 1.7.3 (requirements.txt:9; scipy is not under
/root/reference).  The KDE has several local maxima inside Powell's initial 1.0-wide
bracket, so "MODE" is *what scipy's Powell returns*, and the restatement replays
``scipy/optimize/optimize.py`` step for step in float64:

* ``bracket(func, xa=0.0, xb=1.0)``  (Numerical-Recipes mnbrak, gold 1.618034,
  grow_limit 110, verysmall 1e-21, maxiter 1000)
* ``Brent.optimize`` (tol = xtol*100 = 1e-2, mintol 1e-11, cg 0.3819660, maxiter 500)
* ``_linesearch_powell`` / ``_minimize_powell`` (xtol = ftol = 1e-4, maxiter = maxfev = N*1000)

Version note: scipy >= 1.10 validates the bracket and, when invalid, returns the best of
the three bracket points instead of running Brent (``_recover_from_bracket_error``).
scipy 1.7.3 (the reference pin) ran Brent regardless.  ``validate_bracket=False`` (default)
is the 1.7.3 behaviour; ``validate_bracket=True`` reproduces the scipy installed in this
image so the restatement can be pinned against it (tests/test_oracle_k2p2.py).
"""

import numpy as np

_gold = 1.618034
_verysmall_num = 1e-21


class _MaxFev(Exception):
	pass


def bracket(func, xa=0.0, xb=1.0, grow_limit=110.0, maxiter=1000):
	"""scipy.optimize.bracket.  Returns ``(xa, xb, xc, fa, fb, fc, valid)``."""
	fa = func(xa)
	fb = func(xb)
	if (fa < fb):
		xa, xb = xb, xa
		fa, fb = fb, fa
	xc = xb + _gold * (xb - xa)
	fc = func(xc)
	it = 0
	while (fc < fb):
		tmp1 = (xb - xa) * (fb - fc)
		tmp2 = (xb - xc) * (fb - fa)
		val = tmp2 - tmp1
		if np.abs(val) < _verysmall_num:
			denom = 2.0 * _verysmall_num
		else:
			denom = 2.0 * val
		w = xb - ((xb - xc) * tmp2 - (xb - xa) * tmp1) / denom
		wlim = xb + grow_limit * (xc - xb)
		if it > maxiter:
			raise RuntimeError("Too many iterations.")
		it += 1
		if (w - xc) * (xb - w) > 0.0:
			fw = func(w)
			if (fw < fc):
				xa = xb
				xb = w
				fa = fb
				fb = fw
				break
			elif (fw > fb):
				xc = w
				fc = fw
				break
			w = xc + _gold * (xc - xb)
			fw = func(w)
		elif (w - wlim)*(wlim - xc) >= 0.0:
			w = wlim
			fw = func(w)
		elif (w - wlim)*(xc - w) > 0.0:
			fw = func(w)
			if (fw < fc):
				xb = xc
				xc = w
				w = xc + _gold * (xc - xb)
				fb = fc
				fc = fw
				fw = func(w)
		else:
			w = xc + _gold * (xc - xb)
			fw = func(w)
		xa = xb
		xb = xc
		xc = w
		fa = fb
		fb = fc
		fc = fw

	cond1 = (fb < fc and fb <= fa) or (fb < fa and fb <= fc)
	cond2 = (xa < xb < xc or xc < xb < xa)
	cond3 = np.isfinite(xa) and np.isfinite(xb) and np.isfinite(xc)
	return xa, xb, xc, fa, fb, fc, bool(cond1 and cond2 and cond3)


def brent(func, tol=1.48e-8, maxiter=500, validate_bracket=False):
	"""``Brent(func, tol).optimize()`` with ``brack=None``.  Returns ``(xmin, fval)``."""
	_mintol = 1.0e-11
	_cg = 0.3819660
	xa, xb, xc, fa, fb, fc, valid = bracket(func)
	if validate_bracket and not valid:
		xs, fs = [xa, xb, xc], [fa, fb, fc]
		if np.any(np.isnan(xs)) or np.any(np.isnan(fs)):
			return np.nan, np.nan
		imin = int(np.argmin(fs))
		return xs[imin], fs[imin]

	x = w = v = xb
	fw = fv = fx = fb
	if (xa < xc):
		a = xa
		b = xc
	else:
		a = xc
		b = xa
	deltax = 0.0
	rat = 0.0
	it = 0
	while (it < maxiter):
		tol1 = tol * np.abs(x) + _mintol
		tol2 = 2.0 * tol1
		xmid = 0.5 * (a + b)
		if np.abs(x - xmid) < (tol2 - 0.5 * (b - a)):
			break
		if (np.abs(deltax) <= tol1):
			if (x >= xmid):
				deltax = a - x
			else:
				deltax = b - x
			rat = _cg * deltax
		else:
			tmp1 = (x - w) * (fx - fv)
			tmp2 = (x - v) * (fx - fw)
			p = (x - v) * tmp2 - (x - w) * tmp1
			tmp2 = 2.0 * (tmp2 - tmp1)
			if (tmp2 > 0.0):
				p = -p
			tmp2 = np.abs(tmp2)
			dx_temp = deltax
			deltax = rat
			if ((p > tmp2 * (a - x)) and (p < tmp2 * (b - x)) and
					(np.abs(p) < np.abs(0.5 * tmp2 * dx_temp))):
				rat = p * 1.0 / tmp2
				u = x + rat
				if ((u - a) < tol2 or (b - u) < tol2):
					if xmid - x >= 0:
						rat = tol1
					else:
						rat = -tol1
			else:
				if (x >= xmid):
					deltax = a - x
				else:
					deltax = b - x
				rat = _cg * deltax

		if (np.abs(rat) < tol1):
			if rat >= 0:
				u = x + tol1
			else:
				u = x - tol1
		else:
			u = x + rat
		fu = func(u)

		if (fu > fx):
			if (u < x):
				a = u
			else:
				b = u
			if (fu <= fw) or (w == x):
				v = w
				w = u
				fv = fw
				fw = fu
			elif (fu <= fv) or (v == x) or (v == w):
				v = u
				fv = fu
		else:
			if (u >= x):
				a = x
			else:
				b = x
			v = w
			w = x
			x = u
			fv = fw
			fw = fx
			fx = fu
		it += 1
	return x, fx


def minimize_powell_1d(func, x0, xtol=1e-4, ftol=1e-4, validate_bracket=False, return_nfev=False):
	"""
	``scipy.optimize.minimize(func, x0, method='Powell').x[0]`` for a scalar parameter.

	``func`` takes and returns a Python/numpy float.
	"""
	maxiter = 1000
	maxfun = 1000
	ncalls = [0]

	def f(x):
		ncalls[0] += 1
		return float(func(float(x)))

	x = float(x0)
	direc = 1.0
	fval = f(x)
	x1 = x
	it = 0

	def linesearch(p, xi, fval):
		# _linesearch_powell, unbounded branch
		if xi == 0.0:
			return fval, p, xi
		alpha_min, fret = brent(lambda alpha: f(p + alpha*xi), tol=xtol*100, validate_bracket=validate_bracket)
		xi = alpha_min * xi
		return fret, p + xi, xi

	while True:
		fx = fval
		delta = 0.0
		# for i in range(N): (N = 1)
		direc1 = direc
		fx2 = fval
		fval, x, direc1 = linesearch(x, direc1, fval)
		if (fx2 - fval) > delta:
			delta = fx2 - fval
		it += 1
		bnd = ftol * (np.abs(fx) + np.abs(fval)) + 1e-20
		if 2.0 * (fx - fval) <= bnd:
			break
		if ncalls[0] >= maxfun:
			break
		if it >= maxiter:
			break
		if np.isnan(fx) and np.isnan(fval):
			break

		# Construct the extrapolated point
		direc1 = x - x1
		x1 = x
		x2 = x + 1 * direc1
		fx2 = f(x2)

		if (fx > fx2):
			t = 2.0*(fx + fx2 - 2.0*fval)
			temp = (fx - fval - delta)
			t *= temp*temp
			temp = fx - fx2
			t -= delta*temp*temp
			if t < 0.0:
				fval, x, direc1 = linesearch(x, direc1, fval)
				if direc1 != 0.0:
					direc = direc1

	if return_nfev:
		return x, ncalls[0]
	return x
