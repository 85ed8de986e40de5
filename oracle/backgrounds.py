# -*- coding: utf-8 -*-
"""
ORACLE (test infrastructure only) -- B* / B2 / B3: background estimation on stamps.

B2 and B3 restate ``photometry/prepare.py``: the time smoothing of the backgrounds
(:258, :317-335) and the subtraction + manual-exclude masking (:419-425); both are pinned bit for bit by
``tests/golden/golden_background.npz``, produced by executing exactly those statements of the reference
(tests/golden/make_golden.py:golden_background).

B* is BUILD-DEFINED.  The reference's estimator (``photometry/backgrounds.py:52-211``) only
exists for full 2048x2048 frames: photutils ``Background2D`` on 64x64 tiles (:200-206) plus a
radial component for TESS FFIs; a 15x15 stamp is smaller than one tile.  The stamp-level
analogue stated here keeps every ingredient that still has a meaning on a single tile
(SURVEY.md section 8a, row B*):

* pixel mask exactly as ``backgrounds.py:89-94``: non-finite, > flux_cutoff (8e4), < 0, plus an
  optional manual-exclude image;
* ONE mesh cell = the whole stamp; a cell with more than ``exclude_percentile`` = 50 % masked
  pixels has no estimate (photutils raises for a frame without any usable cell) -> NaN;
* ``SigmaClip(sigma=3, maxiters=5)`` with median centre and population std (astropy 5.1);
* ``SExtractorBackground``: ``std == 0 -> mean``; ``|mean - median| / std < 0.3 ->
  2.5*median - 1.5*mean``; else ``median`` (photutils 1.3.0);
* the 3x3 median filter and the bicubic zoom of a 1x1 mesh are identities -> the background is
  constant over the stamp for each cadence.

Being build-defined, B* is defined DOWN TO THE LAST BIT, once, here (:func:`bstar_frames`), and
``csrc/background.hip`` implements the same arithmetic: which float64 operations, in which order.
Every decision of the estimator -- is a pixel clipped, which branch of the mode estimator -- is a
comparison of float64 numbers; were the two sides free to sum in different orders, a value within
rounding of ``median +- 3 std`` (or of ``|mean - median| = 0.3 std``) could be clipped on one side
only and move the estimate -- and through the sum image a pixel across K2P2's threshold -- by far
more than rounding.  The definition:

1. ``n`` kept pixels; the float64 sums ``s1 = sum x`` and ``s2 = sum x*x`` of the kept values run over EIGHT
   interleaved accumulators (pixel ``p`` of the row-major stamp goes to accumulator ``p % 8``, in pixel order; masked pixels
   add +0.0), combined as ``((a0+a1)+(a2+a3))+((a4+a5)+(a6+a7))`` (:func:`_tree8`).
2. The kept values sorted ascending, ``k[0..n)``; the kept set of the clipping is a rank range ``[lo, hi)``.
3. At most five clipping passes: ``m = hi - lo``; ``med = (k[lower middle] + k[upper middle]) * 0.5``;
   ``q9 = 9 * (m*s2 - s1*s1)``, negative or NaN -> 0;  a value is clipped iff ``d = (x - med) * m`` has ``d*d > q9``
   (above for ``d > 0``, below for ``d < 0``): the 3-sigma test ``|x - med| > 3 std`` with ``std**2 = (m s2 - s1**2) / m**2``,
   without division or square root.  The clipped values are the ``na`` highest and ``nb`` lowest ranks of the range; their
   sums again run over eight accumulators: accumulator ``g`` takes the ``(g + 8 s)``-th value from the top, then the
   ``(g + 8 s)``-th from the bottom, for ``s = 0, 1, ...``; ``s1 -= tree8(r1)``, ``s2 -= tree8(r2)``.
4. The SExtractor rule on the sums of the ``m`` kept values, with one division and no square root: ``q = m*s2 - s1*s1``
   (``= m**2 var``), ``e = s1 - m*med`` (``= m (mean - med)``), ``mean = s1 / m``;  ``q <= 0`` (or NaN) -> ``mean``
   (``std == 0``);  ``e*e < 0.09*q`` -> ``2.5*med - 1.5*mean`` (``|mean - med| / std < 0.3``);  else ``med``;  the result
   rounded to float32.

Against the LITERAL astropy / photutils statements (:func:`fit_background_stamp_literal`: ``np.median``, ``np.std``,
``np.mean`` -- numpy's pairwise sums, the two-pass variance) the defined arithmetic agrees to float32 rounding wherever no
decision is within rounding of its threshold (``tests/test_oracle_background.py``).  **Parity unpinned** against
photutils/astropy (not installable here); pinned by the reference's own known answer
``tests/test_background.py:36-54`` (constant image 1000 -> background 1000, nothing masked).
"""

import numpy as np
from .quality import pixel_filter


def stamp_mask(img, flux_cutoff=8e4, exclude=None):
	"""backgrounds.py:89-97 -- True where the pixel is NOT used."""
	img = np.asarray(img)
	with np.errstate(invalid='ignore'):
		mask = ~np.isfinite(img)
		mask |= (img > flux_cutoff)
		mask |= (img < 0)
	if exclude is not None:
		mask |= np.asarray(exclude, dtype=bool)
	return mask


def sigma_clip(data, sigma=3.0, maxiters=5):
	"""astropy.stats.SigmaClip(sigma, maxiters) on 1-D data (cenfunc=median, stdfunc=std)."""
	data = np.asarray(data, dtype='float64')
	for _ in range(maxiters):
		if data.size == 0:
			break
		med = np.median(data)
		std = np.std(data)
		keep = (data >= med - sigma*std) & (data <= med + sigma*std)
		if np.all(keep):
			break
		data = data[keep]
	return data


def sextractor_background(data):
	"""photutils.SExtractorBackground.calc_background on already clipped 1-D data."""
	if data.size == 0:
		return np.nan
	med = np.median(data)
	mean = np.mean(data)
	std = np.std(data)
	if std == 0:
		return mean
	if np.abs(mean - med) / std < 0.3:
		return 2.5*med - 1.5*mean
	return med


def fit_background_stamp_literal(img, flux_cutoff=8e4, exclude=None, exclude_percentile=50.0):
	"""
	B* in the literal astropy / photutils statements (numpy's own summation orders): what :func:`bstar_frames` defines bit for
	bit, up to rounding.  Returns ``(background scalar float64, mask bool (H, W))``.
	"""
	img = np.asarray(img)
	mask = stamp_mask(img, flux_cutoff, exclude)
	if np.all(mask):
		return np.nan, mask
	if np.sum(mask) > exclude_percentile/100.0 * img.size:
		return np.nan, mask
	data = sigma_clip(img[~mask])
	return sextractor_background(data), mask


def _tree8(a):
	"""``((a0+a1)+(a2+a3))+((a4+a5)+(a6+a7))`` along the last axis (length 8)."""
	return ((a[..., 0] + a[..., 1]) + (a[..., 2] + a[..., 3])) + ((a[..., 4] + a[..., 5]) + (a[..., 6] + a[..., 7]))


def _sums8(Z):
	"""float64 ``[F, P]`` -> (sum, sum of squares) per frame over eight interleaved accumulators, see the module header."""
	F, P = Z.shape
	J = -(-P // 8)
	Zp = np.zeros((F, J*8), dtype='float64')
	Zp[:, :P] = Z
	Zp = Zp.reshape(F, J, 8)
	a1 = np.zeros((F, 8), dtype='float64')
	a2 = np.zeros((F, 8), dtype='float64')
	for j in range(J):
		z = Zp[:, j, :]
		a1 = a1 + z
		a2 = a2 + z*z            # z is a widened float32: the product is exact, so a fused multiply-add gives the same
	return _tree8(a1), _tree8(a2)


def bstar_frames(X, flux_cutoff=8e4, exclude=None, exclude_percentile=50.0, full=False):
	"""
	B*, THE DEFINITION (module header): ``X`` float32 ``(F, P)`` -- F frames of P pixels in row-major stamp order -> float32
	``(F,)``.  ``exclude``: optional bool ``(F, P)``.  With ``full`` also a dict of the clipping's end state (``lo``, ``hi``,
	``n``, ``passes``).  Vectorised over the frames; every float64 operation is written out in the order the header gives.
	"""
	X = np.ascontiguousarray(X, dtype='float32')
	F, P = X.shape
	with np.errstate(invalid='ignore'):
		ok = (X >= np.float32(0)) & (X <= np.float32(flux_cutoff))      # backgrounds.py:91-94; NaN fails both
	if exclude is not None:
		ok &= ~np.asarray(exclude, dtype=bool).reshape(F, P)
	n = ok.sum(axis=1).astype('int64')
	frac = np.float32(exclude_percentile / 100.0)
	usable = (n > 0) & ~((P - n).astype('float32') > frac * np.float32(P))
	s1, s2 = _sums8(np.where(ok, X, np.float32(0)).astype('float64'))
	K = np.sort(np.where(ok, X, np.float32(np.inf)), axis=1).astype('float64')   # kept values first, ascending
	rows = np.arange(F)
	idx = np.arange(P)[None, :]
	lo = np.zeros(F, dtype='int64')
	hi = np.where(usable, n, 1)
	passes = np.zeros(F, dtype='int64')
	g8 = np.arange(8)[None, :]
	with np.errstate(invalid='ignore', over='ignore', divide='ignore'):
		for it in range(6):
			m = hi - lo
			m1 = lo + (m >> 1)
			m0 = np.where(m & 1, m1, m1 - 1)
			med = (K[rows, m0] + K[rows, m1]) * 0.5
			if it == 5:
				break
			mm = m.astype('float64')
			q9 = 9.0 * (mm*s2 - s1*s1)
			q9 = np.where(q9 > 0.0, q9, 0.0)
			D = (K - med[:, None]) * mm[:, None]
			inrange = (idx >= lo[:, None]) & (idx < hi[:, None]) & usable[:, None]
			out = inrange & (D*D > q9[:, None])
			na = np.sum(out & (D > 0.0), axis=1)
			nb = np.sum(out & (D < 0.0), axis=1)
			if not np.any(na | nb):
				break
			passes += ((na | nb) != 0)
			r1 = np.zeros((F, 8), dtype='float64')
			r2 = np.zeros((F, 8), dtype='float64')
			for s in range(int(-(-max(na.max(), nb.max()) // 8))):
				t = g8 + 8*s
				for cnt, pos in ((na, hi[:, None] - 1 - t), (nb, lo[:, None] + t)):        # from the top, then from the bottom
					take = t < cnt[:, None]
					x = np.where(take, K[rows[:, None], np.clip(pos, 0, P - 1)], 0.0)
					r1 = r1 + x
					r2 = r2 + x*x
			s1 = s1 - _tree8(r1)
			s2 = s2 - _tree8(r2)
			lo = lo + nb
			hi = hi - na
		mm = (hi - lo).astype('float64')
		q = mm*s2 - s1*s1
		e = s1 - mm*med
		mean = s1 / mm
		bkg = np.where(q > 0.0, np.where(e*e < 0.09*q, 2.5*med - 1.5*mean, med), mean)
	result = np.where(usable, bkg, np.nan).astype('float32')
	if full:
		return result, {'lo': lo, 'hi': hi, 'n': n, 'passes': passes, 'usable': usable}
	return result


def fit_background_stamp(img, flux_cutoff=8e4, exclude=None, exclude_percentile=50.0):
	"""
	B*: one cadence of one stamp.  Returns ``(background scalar float64, mask bool (H, W))``; the value is the float32 of
	:func:`bstar_frames`.
	"""
	img = np.asarray(img, dtype='float32')
	mask = stamp_mask(img, flux_cutoff, exclude)
	ex = None if exclude is None else np.asarray(exclude, dtype=bool).reshape(1, -1)
	return float(bstar_frames(img.reshape(1, -1), flux_cutoff, ex, exclude_percentile)[0]), mask


def background_series(raw, flux_cutoff=8e4, exclude=None):
	"""B* for a ``(H, W, T)`` cube -> float32 ``(T,)`` (stored like the reference's float32 blocks, prepare.py:327)."""
	H, W, T = raw.shape
	X = np.moveaxis(np.asarray(raw, dtype='float32'), 2, 0).reshape(T, H*W)
	ex = None if exclude is None else np.moveaxis(np.asarray(exclude, dtype=bool), 2, 0).reshape(T, H*W)
	return bstar_frames(X, flux_cutoff, ex)


def time_smooth_width(cadence):
	"""prepare.py:258"""
	return {1800: 3, 600: 9}[int(cadence)]


def smooth_time(bkg_raw, time_smooth=3):
	"""
	B2 (prepare.py:317-335): ``bck[k] = nanmean(block[k-w : k+w+1])``, ``w = time_smooth//2``,
	float32 block, bottleneck nanmean = sequential float32 accumulation / count.
	``bkg_raw``: float32 ``(..., T)`` smoothing along the last axis.
	"""
	x = np.asarray(bkg_raw, dtype='float32')
	N = x.shape[-1]
	w = time_smooth // 2
	out = np.empty_like(x)
	for k in range(N):
		indx1 = max(k - w, 0)
		indx2 = min(k + w + 1, N)
		asum = np.zeros(x.shape[:-1], dtype='float32')
		cnt = np.zeros(x.shape[:-1], dtype='int64')
		for n in range(indx1, indx2):
			v = x[..., n]
			ok = ~np.isnan(v)
			asum = np.where(ok, (asum + np.where(ok, v, np.float32(0))).astype('float32'), asum)
			cnt += ok
		with np.errstate(invalid='ignore', divide='ignore'):
			out[..., k] = np.where(cnt > 0, asum / cnt.astype('float32'), np.float32(np.nan))
	return out


def subtract_background(raw, raw_err, bkg, pixel_flags=None, backapp=False):
	"""
	B3 (prepare.py:419-425): ``flux0 -= backgrounds`` (unless BACKAPP) and manual-exclude pixels
	-> NaN in image and error.  ``bkg`` broadcastable to ``raw`` (float32).
	"""
	img = np.array(raw, dtype='float32', copy=True)
	err = np.array(raw_err, dtype='float32', copy=True)
	if not backapp:
		img -= np.asarray(bkg, dtype='float32')
	if pixel_flags is not None:
		excl = ~pixel_filter(np.asarray(pixel_flags))
		img[excl] = np.nan
		err[excl] = np.nan
	return img, err


#--------------------------------------------------------------------------------------------------
# B1: the full-frame estimator (backgrounds.py:52-211), branch taken for a plain ndarray image
#--------------------------------------------------------------------------------------------------
def mesh_statistics(img, mask, box=64, sigma=3.0, maxiters=5):
	"""
	The low-resolution mesh of ``photutils.Background2D(img, (box, box), sigma_clip=SigmaClip(3, maxiters=5),
	bkg_estimator=SExtractorBackground, mask=mask)`` (backgrounds.py:200-206), photutils 1.3.0 as published: the image is cut
	into ``box x box`` cells (the reference's 2048 x 2048 frames divide evenly; other sizes are padded with masked pixels,
	``edge_method='pad'``), the unmasked pixels of every cell are sigma-clipped ONCE (``Background2D`` applies its own
	``sigma_clip`` and switches the estimator's off) and reduced by the SExtractor estimate.  Returns ``(mesh float64
	(ny, nx), nmasked int (ny, nx))``; a cell without any unmasked pixel is NaN.

	``nmasked`` counts what photutils' SECOND mesh selection counts (``Background2D._calc_bkg_bkgrms``: "perform mesh rejection
	on sigma-clipped data (i.e., for any newly-masked pixels)", ``np.ma.count_masked(data_sigclip, axis=1)``; its
	``mesh_nmasked`` property is the same number): the input mask, the padding AND the pixels the sigma clip rejected.  The
	first selection (input mask + padding only) is implied by it: the clip only adds masked pixels.
	"""
	img = np.asarray(img)
	R, C = img.shape
	ny, nx = -(-R // box), -(-C // box)
	mesh = np.full((ny, nx), np.nan)
	nmasked = np.zeros((ny, nx), dtype='int64')
	for j in range(ny):
		for i in range(nx):
			cell = img[j*box:(j+1)*box, i*box:(i+1)*box]
			m = mask[j*box:(j+1)*box, i*box:(i+1)*box]
			data = sigma_clip(cell[~m], sigma, maxiters)
			nmasked[j, i] = box*box - data.size           # masked + padded + clipped pixels
			mesh[j, i] = sextractor_background(data)
	return mesh, nmasked


def finish_mesh(mesh, nmasked, box=64, exclude_percentile=50.0, filter_size=3):
	"""The low-resolution half of :func:`mesh_to_background`: rejected cells filled, then the 3 x 3 NaN-ignoring median filter."""
	from scipy import ndimage
	mesh = np.array(mesh, dtype='float64', copy=True)
	good = (nmasked <= exclude_percentile / 100.0 * box * box) & np.isfinite(mesh)
	if not np.any(good):
		raise ValueError("All meshes contain > %s masked pixels" % exclude_percentile)
	if not np.all(good):
		yy, xx = np.nonzero(good)
		vals = mesh[good]
		for (y, x) in zip(*np.nonzero(~good)):
			d = np.hypot(yy - y, xx - x)
			near = np.argsort(d, kind='stable')[:10]
			w = 1.0 / d[near]
			mesh[y, x] = np.sum(w * vals[near]) / np.sum(w)
	if filter_size > 1:
		mesh = ndimage.generic_filter(mesh, np.nanmedian, size=filter_size, mode='constant', cval=np.nan)
	return mesh


def mesh_to_background(mesh, nmasked, shape, box=64, exclude_percentile=50.0, filter_size=3):
	"""
	From the mesh to the full-resolution background, photutils 1.3.0 as published: cells with more than
	``exclude_percentile`` % masked pixels (``nmasked <= exclude_percentile / 100 * box**2`` keeps a cell, ``_select_meshes``;
	``nmasked`` after the sigma clip, see :func:`mesh_statistics`) are dropped and filled by inverse-distance weighting from
	the 10 nearest kept cells (``ShepardIDWInterpolator``, power 1, weights ``1 / distance`` in mesh-index units; which of
	several EQUIDISTANT cells make the ten is cKDTree's traversal order upstream and index order here: unpinned); 3 x 3 median filter (``generic_filter(nanmedian, mode='constant',
	cval=nan)``); ``BkgZoomInterpolator``: cubic-spline ``scipy.ndimage.zoom(mesh, box, order=3, mode='reflect',
	grid_mode=True)``, clipped to the range of the mesh.  scipy is called directly, as photutils does.
	"""
	from scipy import ndimage
	mesh = finish_mesh(mesh, nmasked, box, exclude_percentile, filter_size)
	if mesh.shape == (1, 1):
		return np.full(shape, mesh[0, 0])
	bkg = ndimage.zoom(mesh, box, order=3, mode='reflect', cval=0.0, grid_mode=True)
	bkg = np.clip(bkg, np.min(mesh), np.max(mesh))
	return bkg[:shape[0], :shape[1]]


def fit_background(image, flux_cutoff=8e4, exclude=None):
	"""
	``fit_background(image)`` for a plain 2-D array (backgrounds.py:52-211): not a TESS FFIImage, so ``bkgiters = 1`` and no
	radial component (:156-157).  Returns ``(background float64, mask bool)``.  **Parity unpinned** against photutils /
	astropy (not installable here) except for the reference's own known answer: a constant image comes back as that
	constant with nothing masked (tests/test_background.py:36-54).
	"""
	img = np.asarray(image)
	mask = stamp_mask(img, flux_cutoff, exclude)          # backgrounds.py:89-97
	if np.all(mask):
		return np.full(img.shape, np.nan), mask
	mesh, nmasked = mesh_statistics(img, mask)
	return mesh_to_background(mesh, nmasked, img.shape), mask


#--------------------------------------------------------------------------------------------------
# TESS branch of fit_background: the radial component (backgrounds.py:104-197)
#--------------------------------------------------------------------------------------------------
#: pixel coordinates of the camera centre relative to every CCD (backgrounds.py:118-135: averages of the Sector 1 WCS)
CAMERA_CENTRE = {
	(1, 1): (2158.222313, 2099.523364), (1, 2): (-5.653058, 2098.018608), (1, 3): (2141.511437, 2099.868226), (1, 4): (-22.406442, 2100.116443),
	(2, 1): (2148.588316, 2094.033024), (2, 2): (-16.806140, 2095.810070), (2, 3): (2151.351646, 2105.747100), (2, 4): (-13.118570, 2105.982211),
	(3, 1): (2152.175481, 2092.337442), (3, 2): (-10.494413, 2093.108135), (3, 3): (2145.029218, 2107.883573), (3, 4): (-17.374782, 2105.296746),
	(4, 1): (2149.259760, 2091.433315), (4, 2): (-12.906931, 2093.350054), (4, 3): (2148.906766, 2110.730620), (4, 4): (-14.629676, 2111.341670),
}


def reduce_mode(x, binning='float'):
	"""``_reduce_mode`` (backgrounds.py:20-32): mode of the KDE (statsmodels default bandwidth, 2000 -> 2048 grid points)."""
	from .kde import KDE
	if len(x) == 0:
		return np.nan
	x = np.asarray(x, dtype='float64')
	kde = KDE(x)
	try:
		with np.errstate(all='ignore'):
			kde.fit(gridsize=2000, binning=binning)
	except RuntimeError as err:
		if str(err).startswith('Selected KDE bandwidth is 0.'):
			return np.median(x)
		raise
	return kde.support[np.argmax(kde.density)]


def binned_callable(x, values, statistic, bins):
	"""
	``scipy.stats.binned_statistic(x, values, statistic=<callable>, bins=<edges>)`` (scipy 1.7.3, _binned_statistic.py):
	``np.digitize`` bin numbers, samples within rounding (``decimal = int(-log10(min edge step)) + 6``) of the last edge moved
	into the last bin, the callable applied to the values of every occupied bin in input order, ``statistic([])`` elsewhere.
	"""
	bins = np.asarray(bins, dtype='float64')
	nbin = len(bins) - 1
	number = np.digitize(x, bins)
	decimal = int(-np.log10(np.diff(bins).min())) + 6
	on_edge = np.where(np.around(x, decimal) == np.around(bins[-1], decimal))[0]
	number[on_edge] -= 1
	try:
		null = statistic([])
	except Exception:
		null = np.nan
	result = np.full(nbin + 2, null, dtype='float64')
	for i in np.unique(number):
		result[i] = statistic(values[number == i])
	return result[1:-1]


def move_median_central(x, width_points):
	"""utilities.move_median_central (utilities.py:52-62): the restatement in ``oracle/utilities.py`` (pinned by ``golden_misc.npz``)."""
	from .utilities import move_median_central as mmc
	import warnings
	with warnings.catch_warnings():
		warnings.simplefilter('ignore', RuntimeWarning)    # all-NaN windows
		with np.errstate(all='ignore'):
			return mmc(np.asarray(x, dtype='float64'), width_points)


def radial_geometry(shape, camera, ccd, radial_cutoff=2400, radial_pixel_step=15):
	"""backgrounds.py:137-149: distance image (float64), ring edges and ring centres."""
	xycen = CAMERA_CENTRE[(camera, ccd)]
	xx, yy = np.meshgrid(np.arange(44, shape[1]+44, 1), np.arange(0, shape[0], 1))
	r = np.sqrt((xx - xycen[0])**2 + (yy - xycen[1])**2)
	radial_max = np.max(r) + radial_pixel_step
	bins = np.arange(radial_cutoff, radial_max, radial_pixel_step)
	bin_center = bins[1:] - radial_pixel_step/2
	return r, bins, bin_center


def fit_background_tess(image, camera, ccd, flux_cutoff=8e4, exclude=None, bkgiters=3, radial_cutoff=2400, radial_pixel_step=15,
	radial_smooth=3, full=False, device_arithmetic=False):
	"""
	``fit_background`` for a TESS FFIImage (backgrounds.py:52-211 with ``is_tess``): ``bkgiters`` rounds of the radial
	component (ring modes of log10(img - square + zeropoint), 3-point median, interpolating cubic spline, :162-197) and the
	Background2D mesh of ``img0 - img_bkg_radial`` (:199-206).  ``image``: float32 ``(R, C)`` = ``FFIImage.data`` (the 2048
	science columns; the distance image starts at column 44, :144).  The numpy 1.21 type promotion the reference runs under
	is written out: in the first round ``img0 - 0`` is float32, so ``zeropoint = -min + 1.0`` is a float64 made from a float32
	minimum, ``pix + zeropoint`` and ``log10`` are float32; from the second round on everything is float64.
	Returns ``(background float64, mask)`` (+ a dict of intermediates with ``full``).  **Parity unpinned** against statsmodels /
	photutils (not installable here); the scipy parts (binned statistic, spline, zoom) are the real scipy.

	``device_arithmetic``: the same algorithm with the three roundings the device makes, for a comparison that is not decided by
	them (a ring mode is the argmax over a KDE grid: where two grid points tie within rounding, ANY last-bit difference moves the
	mode by a whole grid step).  (1) First round: the float32 ``log10`` is the correctly rounded one -- numpy's float32
	``log10`` is a libm / SVML routine with 1-ulp errors on ~40 % of its arguments, different between numpy builds and CPUs, so
	the reference's own ring modes are not reproducible across machines at exactly these ties; (2) the two background components
	are stored as float32 images between the rounds and added in float32 storage at the end (the reference keeps float64);
	(3) the linear binning accumulates fixed-point integers (:func:`oracle.kde.fast_linbin_fixed`).
	"""
	from scipy.interpolate import InterpolatedUnivariateSpline
	img0 = np.asarray(image, dtype='float32')
	mask = stamp_mask(img0, flux_cutoff, exclude)
	if np.all(mask):
		return (np.full(img0.shape, np.nan), mask, {}) if full else (np.full(img0.shape, np.nan), mask)
	r, bins, bin_center = radial_geometry(img0.shape, camera, ccd, radial_cutoff, radial_pixel_step)
	img_bkg_radial = np.asarray(0)
	img_bkg_square = None
	inter = {'s2': [], 'zeropoint': [], 'radial': []}
	for iters in range(bkgiters):
		if img_bkg_square is None:
			pix = img0[~mask].flatten()                                # float32
			zeropoint = -np.float64(np.min(pix)) + 1.0                  # numpy 1.x: float32 scalar + Python float -> float64
			logpix = np.log10(pix + np.float32(zeropoint))              # float32 array + scalar stays float32
			if device_arithmetic:
				logpix = np.log10((pix + np.float32(zeropoint)).astype('float64')).astype('float32')
		else:
			img = img0.astype('float64') - (img_bkg_square.astype('float32').astype('float64') if device_arithmetic else img_bkg_square)
			pix = img[~mask].flatten()
			zeropoint = -np.min(pix) + 1.0
			logpix = np.log10(pix + zeropoint)
		s2 = binned_callable(r[~mask].flatten(), logpix, (lambda v: reduce_mode(v, binning='fixed')) if device_arithmetic else reduce_mode, bins)
		inter['s2'].append(s2.copy())
		inter['zeropoint'].append(float(zeropoint))
		if radial_smooth:
			s2 = move_median_central(s2, radial_smooth)
		indx = ~np.isnan(s2)
		Ngood = np.sum(indx)
		if Ngood >= 3:
			try:
				intp = InterpolatedUnivariateSpline(bin_center[indx], s2[indx], k=3, ext=3)
				img_bkg_radial = 10**intp(r) - zeropoint
			except ValueError:
				img_bkg_radial = 0
		else:
			img_bkg_radial = 0
		inter['radial'].append(np.array(img_bkg_radial, dtype='float64', copy=True))
		work = img0.astype('float64') - (np.asarray(img_bkg_radial, dtype='float32').astype('float64') if device_arithmetic else img_bkg_radial)
		mesh, nmasked = mesh_statistics(work, mask)
		img_bkg_square = mesh_to_background(mesh, nmasked, img0.shape)
	if device_arithmetic:
		img_bkg = (img_bkg_radial + img_bkg_square.astype('float32').astype('float64')).astype('float32').astype('float64')
	else:
		img_bkg = img_bkg_radial + img_bkg_square
	return (img_bkg, mask, inter) if full else (img_bkg, mask)


#--------------------------------------------------------------------------------------------------
# Pixel flags: "background shenanigans" (pixel_flags.py:61-79, prepare.py:515-622)
#--------------------------------------------------------------------------------------------------
def pixel_manual_exclude(data, is_tess=False, camera=None, ccd=None, cadenceno=None, tstart=np.nan, tstop=np.nan):
	"""
	pixel_flags.pixel_manual_exclude (pixel_flags.py:13-58) with the header cards as arguments (``cadenceno`` = FFIINDEX, None
	when the card is missing): Mars in camera 1 CCD 4 (columns >= 1536), Earth-shine in camera 1 (everything), a TESS image
	that is zero everywhere (everything).  Returns the bool mask.
	"""
	data = np.asarray(data)
	mask = np.zeros(data.shape, dtype='bool')
	if is_tess:
		time = 0.5*(tstart + tstop)
		cadenceno = np.inf if cadenceno is None else cadenceno
	else:
		time = np.nan
		cadenceno = np.inf
	if is_tess and camera == 1 and ccd == 4 and (cadenceno <= 4724 or tstart <= 1325.881282301840):
		mask[:, 1536:] = True
	elif is_tess and camera == 1 and (11354 <= cadenceno <= 11366 or 1464.0158778 <= time <= 1464.265871):
		mask[:, :] = True
	if is_tess and np.all(data == 0):
		mask[:, :] = True
	return mask


def prepare_pixel_flags(raw, masks, manexcl):
	"""
	The pixel flags the prepare stage stores per frame (prepare.py:296-297, 406-408): NotUsedForBackground (1) where
	``fit_background`` masked the pixel, ManualExclude (2) where ``pixel_manual_exclude`` did; and ``backgrounds_pixels_used``
	(:435, :464-466): the pixel was used for the background in more than half of the frames.
	"""
	flags = np.where(masks, 1, 0).astype('uint8')
	flags[manexcl] |= 2
	used = np.sum((flags & 1) == 0, axis=0, dtype='int32')
	return flags, (used / flags.shape[0] > 0.5)


def pixel_background_shenanigans(img, SumImage=None):
	"""pixel_flags.py:61-79"""
	from scipy.ndimage import median_filter
	flux0 = (img - SumImage) if SumImage is not None else img
	return median_filter(flux0, size=15)


def shenanigans_block_frames(indicies, k, block=25):
	"""
	The frames whose median makes up the block that starts at position ``k`` of the shuffled list (prepare.py:562-571).  The
	reference fills ONE ``(R, C, block)`` buffer block after block and takes ``nanmedian`` over ALL its slots: a short last block
	overwrites only its first ``L`` slots, the others still hold frames ``L .. block-1`` of the previous block -- they take part.
	(With fewer than ``block`` frames in total the untouched slots are uninitialised memory upstream: undefined; here they are
	left out.)  Pinned by ``tests/golden/golden_shenanigans.npz`` (those statements executed).
	"""
	own = list(indicies[k:k + block])
	if len(own) < block and k >= block:
		own += list(indicies[k - block + len(own):k])
	return own


def background_shenanigans_flags(images, SumImage, pixel_flags, bkgshe_threshold=40, block=25, flag=4):
	"""
	prepare.py:515-622 on in-memory arrays: ``images`` float32 ``(T, R, C)``, ``pixel_flags`` integer ``(T, R, C)`` (a
	modified copy is returned together with the float32 indicator stack and the float64 mean image).
	"""
	images = np.asarray(images)
	numfiles = images.shape[0]
	pixel_flags_ind = np.empty(images.shape[1:] + (numfiles,), dtype='float32')
	for k in range(numfiles):
		pixel_flags_ind[:, :, k] = pixel_background_shenanigans(images[k], SumImage=SumImage)
	mean_shenanigans = np.zeros_like(SumImage, dtype='float64')
	indicies = list(range(numfiles))
	np.random.seed(0)
	np.random.shuffle(indicies)
	for k in range(0, numfiles, block):
		blockdata = np.stack([pixel_flags_ind[:, :, i].astype('float64') for i in shenanigans_block_frames(indicies, k, block)], axis=2)
		with np.errstate(all='ignore'):
			bckshe = np.nanmedian(blockdata, axis=2)
		bckshe[np.isnan(bckshe)] = 0
		mean_shenanigans += bckshe
	mean_shenanigans /= np.ceil(numfiles/block)
	flags = np.array(pixel_flags, copy=True)
	for k in range(numfiles):
		bckshe = np.abs(pixel_flags_ind[:, :, k] - mean_shenanigans) > bkgshe_threshold
		indx = (flags[k] & flag != 0)
		flags[k][indx] -= flag
		flags[k][bckshe] |= flag
	return flags, np.moveaxis(pixel_flags_ind, 2, 0), mean_shenanigans
