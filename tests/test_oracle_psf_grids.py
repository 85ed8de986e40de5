# -*- coding: utf-8 -*-
"""
The oracle's FITPACK box integral (``oracle.psf.fpintb`` / ``integrate_to_image``) on PRF grids other than the SPOC layout and
with cut-off radii beyond the grid -- the cases the general device kernels are checked against (tests/test_gpu_linpsf.py,
tests/test_gpu_psfphot.py) -- pinned against the reference's own expression: scipy's ``RectBivariateSpline.integral`` over every
pixel (photometry/psf.py:136-146), called literally.
"""
import numpy as np
import pytest
from oracle import psf as opsf
from prf_common import general_prf


@pytest.mark.parametrize("kind,cutoff", [('warped', 5), ('warped', None), ('nsub7', 6.5), ('coarse', None), ('spoc', 7.5), ('spoc', None), ('rect', 5), ('rect', None)])
def test_box_integral_on_any_grid_equals_scipy(kind, cutoff):
	prf = general_prf(kind)
	stamp = (100, 117, 300, 316)      # 17 x 16: wider than every grid here, so pixels beyond the PRF's support are included
	p = opsf.PSF(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'], stamp)
	params = np.array([[8.3, 7.6, 1000.0], [2.1, 12.4, 250.0], [15.9, 0.2, 80.0]])
	img = p.integrate_to_image(params, cutoff_radius=cutoff)
	ref = p.integrate_to_image_scipy(params, cutoff_radius=cutoff)
	np.testing.assert_allclose(img, ref, rtol=1e-11, atol=1e-15 * np.abs(ref).max())
	np.testing.assert_array_equal(img == 0, ref == 0)
	if cutoff is None:
		# the whole PRF of the first star lies on the stamp: its integral is the normalisation of psf.py:116 (sum * cdelt^2 = 1) to
		# the accuracy of a Riemann sum against the spline's integral
		one = p.integrate_to_image(params[:1] * [1, 1, 0] + [0, 0, 1.0], cutoff_radius=None)
		assert abs(one.sum() - 1.0) < 0.05
