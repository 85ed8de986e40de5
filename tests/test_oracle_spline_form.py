# -*- coding: utf-8 -*-
"""
The identity the matrix-core LinPSF fit rests on (photometry_amd/csrc/linpsf_mfma.hip, tp_linpsf_coef_kernel), checked on the
CPU against the reference's own expression for the pixel-integrated PRF (``RectBivariateSpline.integral`` over the pixel,
photometry/psf.py:136-146):

as a function of the star's position the integral over one pixel is a tensor-product QUARTIC SPLINE in the two sub-pixel
phases with simple knots on the PRF sample grid, so over the ``na x nb`` knot intervals a star visits it equals

    F(X, Y) = sum_ED C[E][D] m_E(X) m_D(Y),    m = {1, t, t^2, t^3, t^4, (t-1)+^4, (t-2)+^4},

where ``C[e][d]`` (e, d <= 4) is the biquartic of interval (0, 0), ``C[4+a][d] = K(a,0)[4][d] - K(a-1,0)[4][d]``,
``C[e][4+b] = K(0,b)[e][4] - K(0,b-1)[e][4]`` and ``C[4+a][4+b]`` the second difference of ``K[4][4]`` -- K(a, b) being the
biquartic of interval (a, b) in ITS OWN local phases (what the vector-ALU path evaluates per table origin).
"""
import numpy as np
from oracle import psf as opsf


def _setup():
	prf = opsf.synthetic_prf(seed=7)
	p = opsf.PSF(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'], (0, 15, 0, 15))
	return p


def _position(kn, X, jstar):
	"""The star coordinate whose pixel-edge phase, counted in knot intervals from the first interior knot, is X (the inverse of
	the device's ``X = ((jstar - pos - 0.5) - kn[4]) / h + 1 - 9 jstar``)."""
	h = kn[5] - kn[4]
	return jstar - 0.5 - kn[4] - (X - 1.0 + 9.0 * jstar) * h


def test_pixel_integral_is_one_quartic_spline_over_the_visited_intervals():
	p = _setup()
	exact = lambda i, j, row, col: p.splineInterpolation.integral((j - col) - 0.5, (j - col) + 0.5, (i - row) - 0.5, (i - row) + 0.5)
	rng = np.random.default_rng(5)
	ph = np.array([0.08, 0.3, 0.5, 0.72, 0.93])
	V = np.vander(ph, 5, increasing=True)                 # phases -> monomials
	for (i, j, jr, jc, na, nb) in [(7, 9, 7, 7, 3, 2), (4, 6, 7, 7, 2, 3), (8, 8, 7, 8, 3, 3), (10, 5, 8, 7, 1, 2)]:
		hx, hy = p.tx[5] - p.tx[4], p.ty[5] - p.ty[4]
		# the global phase of the pixel-centre position (jr, jc) and the intervals around it
		X0 = np.floor(((jc - (jc - 0.2) - 0.5) - p.tx[4]) / hx + 1 - 9 * jc)
		Y0 = np.floor(((jr - (jr + 0.1) - 0.5) - p.ty[4]) / hy + 1 - 9 * jr)
		# the biquartic of every interval from 5 x 5 exact integrals (local phases), as the per-origin path has it
		K = np.empty((na, nb, 5, 5))
		for a in range(na):
			for b in range(nb):
				vals = np.array([[exact(i, j, _position(p.ty, Y0 + b + fy, jr), _position(p.tx, X0 + a + fx, jc)) for fy in ph] for fx in ph])
				K[a, b] = np.linalg.solve(V, np.linalg.solve(V, vals.T).T)      # vals[x, y] = sum_ed K[e, d] fx^e fy^d
		C = np.zeros((7, 7))
		C[:5, :5] = K[0, 0]
		for a in range(1, na):
			C[4 + a, :4] = K[a, 0][4, :4] - K[a - 1, 0][4, :4]
		for b in range(1, nb):
			C[:4, 4 + b] = K[0, b][:4, 4] - K[0, b - 1][:4, 4]
		k44 = lambda a, b: K[a, b][4, 4] if (a >= 0 and b >= 0) else 0.0
		for a in range(na):
			for b in range(nb):
				C[4 + a, 4 + b] = k44(a, b) - k44(a - 1, b) - k44(a, b - 1) + k44(a - 1, b - 1)
		basis = lambda t: np.array([1.0, t, t**2, t**3, t**4, max(t - 1.0, 0.0)**4, max(t - 2.0, 0.0)**4])
		scale = max(abs(exact(i, j, jr, jc)), 1e-6)
		worst = 0.0
		for _ in range(200):
			X, Y = rng.uniform(0, na), rng.uniform(0, nb)
			F = basis(X) @ C @ basis(Y)
			ref = exact(i, j, _position(p.ty, Y0 + Y, jr), _position(p.tx, X0 + X, jc))
			worst = max(worst, abs(F - ref) / scale)
		# the residual is the conditioning of the 5 x 5 fits above (the device contracts the table exactly instead of fitting, and
		# is held to 1e-8 against the oracle by tests/test_gpu_linpsf.py); a wrong rule is off by 1e-2 and more
		assert worst < 1e-7, (i, j, na, nb, worst)
