# -*- coding: utf-8 -*-
"""
ORACLE (test infrastructure only) -- A2..A5: K2P2 pixel-mask creation.

Restates ``photometry/AperturePhotometry/k2p2v2.py``:
``run_DBSCAN`` (:63-86), ``k2p2WS`` (:89-288), ``k2p2_saturated`` (:291-341) and
``k2p2FixFromSum`` (:344-746, numerical part :388-623).

Third-party pieces and how they are stated here:

* scipy (installed here, called directly exactly as the reference does):
  ``stats.trim1`` (:402), ``ndimage.gaussian_filter`` (:135), ``ndimage.label`` (:197, :215,
  :328), ``ndimage.convolve`` (:550).
* scipy ``minimize(method='Powell')`` (:421): restated in :mod:`oracle.powell`
  (scipy-1.7.3 semantics; pinned against the installed scipy in the tests).
* scikit-learn ``DBSCAN`` (:79-80): on a pixel grid with ``eps = sqrt(2)+eps`` and
  ``min_samples = 4`` it is exactly: core pixel <=> at least 4 above-threshold pixels in
  its 3x3 neighbourhood (itself included); clusters = 8-connected components of core
  pixels numbered in raster order of their first core pixel; a non-core pixel gets the
  lowest label among its 8-neighbour core pixels, or -1 (noise)
  (``sklearn/cluster/_dbscan_inner.pyx``; pinned against the installed scikit-learn).
* statsmodels KDE / bandwidth (:410-420): restated in :mod:`oracle.kde` (pinned independently against scipy's
  Gaussian KDE, the direct sum and numpy percentiles: tests/test_oracle_pins.py).
* scikit-image 0.19.2 ``peak_local_max`` (:141) and ``watershed`` (:227): restated below from
  ``skimage/feature/peak.py`` and ``skimage/segmentation/_watershed(_cy.pyx)``; scikit-image itself cannot be run
  here, so both are pinned by hand-derived fixtures written out in tests/test_oracle_pins.py (two basins with a
  saddle, plateau ties broken by push order, a marker outside the mask, diagonal contact vs a one-pixel bridge; flat
  image -> no peaks, border and corner peaks kept, plateau peaks, the ``max(min, threshold_rel * max)`` threshold).
"""

import heapq
import numpy as np
from scipy import stats, ndimage
from .kde import KDE, select_bandwidth
from .powell import minimize_powell_1d
from .utilities import mad_to_sigma

#: k2p2v2.py:49
saturation_limit = 7.0


class K2P2NoFlux(Exception):
	pass


class K2P2NoStars(Exception):
	pass


def nanmedian(x):
	"""bottleneck.nanmedian (NaN for empty / all-NaN input, no warning)."""
	x = np.asarray(x, dtype='float64').ravel()
	x = x[~np.isnan(x)]
	if x.size == 0:
		return np.nan
	return np.median(x)

#--------------------------------------------------------------------------------------------------
# A2: threshold
#--------------------------------------------------------------------------------------------------
def threshold(SumImage, thresh=0.8, validate_bracket=False, full_output=False):
	"""
	k2p2v2.py:388-427.  Returns ``CUT`` (float); with ``full_output`` a dict with
	``CUT, MODE, MAD1, bandwidth, max_guess, nflux``.
	"""
	ori_mask = ~np.isnan(SumImage)
	Flux = SumImage[ori_mask].flatten()
	Flux = Flux[Flux > 0]
	if len(Flux) == 0:
		raise K2P2NoFlux("No measured flux in sum-image")

	flux_cut = stats.trim1(np.sort(Flux), 0.15)
	# trim1 (scipy/stats: uppercut = n - int(0.15 n), np.partition) keeps the n-int(0.15n)
	# smallest values; their order is unspecified -> use sorted order:
	flux_cut = np.sort(flux_cut)
	flux_cut = flux_cut[flux_cut < 70000]

	background_bandwidth = select_bandwidth(flux_cut, bw='scott', kernel='gau')

	kernel = KDE(flux_cut)
	kernel.fit(kernel='gau', bw=background_bandwidth, fft=True, gridsize=100)

	def kernel_opt(x):
		return -1*kernel.evaluate(x)[0]
	max_guess = kernel.support[np.argmax(kernel.density)]
	MODE = minimize_powell_1d(kernel_opt, max_guess, validate_bracket=validate_bracket)

	MAD1 = mad_to_sigma * nanmedian(np.abs(Flux[(Flux < MODE)] - MODE))
	CUT = MODE + thresh * MAD1
	if full_output:
		return {'CUT': CUT, 'MODE': MODE, 'MAD1': MAD1, 'bandwidth': float(background_bandwidth),
			'max_guess': max_guess, 'nflux': len(Flux), 'nflux_cut': len(flux_cut)}
	return CUT

#--------------------------------------------------------------------------------------------------
# A3: DBSCAN on the pixel grid
#--------------------------------------------------------------------------------------------------
def dbscan_grid(idx, min_for_cluster=4):
	"""
	``run_DBSCAN`` (k2p2v2.py:63-86, called :461) restated on the grid.

	Returns:
		labels (ndarray): int ``(H, W)``: -2 where ``idx`` is False, -1 noise, >= 0 cluster.
		core (ndarray): bool ``(H, W)`` core-sample flags.
	"""
	idx = np.asarray(idx, dtype='bool')
	H, W = idx.shape
	pad = np.zeros((H+2, W+2), dtype='int32')
	pad[1:-1, 1:-1] = idx
	count = np.zeros((H, W), dtype='int32')
	for dy in (0, 1, 2):
		for dx in (0, 1, 2):
			count += pad[dy:dy+H, dx:dx+W]
	core = idx & (count >= min_for_cluster)

	# 8-connected components of the core pixels, numbered in raster order of first pixel:
	comp, ncomp = ndimage.label(core, structure=np.ones((3, 3), dtype='int32'))
	labels = np.full((H, W), -2, dtype='int64')
	labels[idx] = -1
	labels[core] = comp[core] - 1
	# Border pixels: lowest label among the neighbouring core pixels
	cpad = np.full((H+2, W+2), np.iinfo('int64').max, dtype='int64')
	cpad[1:-1, 1:-1][core] = labels[core]
	best = np.full((H, W), np.iinfo('int64').max, dtype='int64')
	for dy in (0, 1, 2):
		for dx in (0, 1, 2):
			best = np.minimum(best, cpad[dy:dy+H, dx:dx+W])
	border = idx & ~core & (best != np.iinfo('int64').max)
	labels[border] = best[border]
	return labels, core

#--------------------------------------------------------------------------------------------------
# scikit-image restatements
#--------------------------------------------------------------------------------------------------
def peak_local_max(image, threshold_rel=0, footprint=None):
	"""
	``skimage.feature.peak_local_max(image, exclude_border=False, threshold_rel=..., footprint=...)``
	(0.19.2, ``min_distance=1``, ``indices=True``).  Returns ``(n, 2)`` int array of
	``(row, col)`` sorted by decreasing intensity.
	"""
	image = np.asarray(image)
	threshold = image.min()
	if threshold_rel is not None:
		threshold = max(threshold, threshold_rel * image.max())
	if footprint is None:
		footprint = np.ones((3, 3), dtype=bool)
	if footprint.size == 1 or image.size == 1:
		mask = image > threshold
	else:
		image_max = ndimage.maximum_filter(image, footprint=footprint, mode='constant')
		mask = (image == image_max)
		if np.all(mask): # no peak for a trivial image
			mask[:] = False
		mask &= image > threshold
	coord = np.nonzero(mask)
	intensities = image[coord]
	idx_maxsort = np.argsort(-intensities, kind='stable')
	coord = np.transpose(coord)[idx_maxsort]
	# ensure_spacing(spacing=1, p_norm=inf) rejects only points closer than 1 pixel: none on a grid.
	return coord


def watershed(image, markers, mask):
	"""
	``skimage.segmentation.watershed(image, markers, mask=mask)`` (0.19.2; connectivity 1,
	no compactness, no watershed lines).  Priority flood from ``_watershed_cy.pyx``:
	heap ordered by (value, age); a neighbour is labelled when it is *pushed*.
	"""
	image = np.asarray(image, dtype='float64')
	mask = np.asarray(mask).astype(bool)
	output = (np.asarray(markers) * mask).astype('int32')
	H, W = image.shape
	heap = []
	age = 1
	for index in np.flatnonzero(output):
		heapq.heappush(heap, (image.flat[index], 0, int(index)))
	out = output.ravel()
	img = image.ravel()
	msk = mask.ravel()
	while heap:
		value, _, index = heapq.heappop(heap)
		r, c = divmod(index, W)
		# neighbours in raveled-offset order: -W, -1, +1, +W
		for rr, cc in ((r-1, c), (r, c-1), (r, c+1), (r+1, c)):
			if rr < 0 or rr >= H or cc < 0 or cc >= W:
				continue
			nb = rr*W + cc
			if not msk[nb]:
				continue
			if out[nb]:
				continue
			age += 1
			out[nb] = out[index]
			heapq.heappush(heap, (img[nb], age, nb))
	return out.reshape(H, W)

#--------------------------------------------------------------------------------------------------
# A4: watershed segmentation of each cluster
#--------------------------------------------------------------------------------------------------
def k2p2WS(flux0, labels_grid, core, saturated_masks=None, ws_thres=0, ws_footprint=3, ws_blur=0.5, catalog=None):
	"""
	``k2p2WS`` (k2p2v2.py:89-288) with ``ws_alg='flux'``, on grid arrays.

	Parameters:
		flux0: sum image.
		labels_grid: output of :func:`dbscan_grid`.
		core: core flags.
		saturated_masks: ``None`` or dict label -> bool image (k2p2v2.py:486-492).
		catalog: ``(n, 3)`` array of (column, row, tmag) or ``None``.

	Returns the new label grid (``-2`` outside ``idx``, ``-1`` noise).
	"""
	outside = (labels_grid == -2)
	unique_labels_ini = sorted(set(labels_grid[~outside].tolist()))
	Labels = np.array(labels_grid, dtype='float64')
	Labels[~core] = -1 # k2p2v2.py:112 (this also resets the -2 pixels, which are never read back)
	max_label = np.max(labels_grid[~outside])

	for lab in unique_labels_ini:
		if lab == -1 or lab == -2:
			continue
		class_member = (Labels == lab)
		Z = np.zeros_like(flux0, dtype='float64')
		Z[class_member] = flux0[class_member]
		distance0 = Z

		distance = ndimage.gaussian_filter(distance0, ws_blur)
		local_maxi_loc = peak_local_max(distance, threshold_rel=ws_thres, footprint=np.ones((ws_footprint, ws_footprint)))
		if catalog is not None:
			local_maxi = np.zeros_like(flux0, dtype='bool')
			for c in catalog:
				d = np.sqrt((local_maxi_loc[:, 1] - c[0])**2 + (local_maxi_loc[:, 0] - c[1])**2)
				indx = np.argmin(d) # ValueError on empty, as in the reference (:146)
				dist_factor = 2.0 if c[2] > saturation_limit else 5.0
				if d[indx] < dist_factor*np.sqrt(2):
					local_maxi[local_maxi_loc[indx, 0], local_maxi_loc[indx, 1]] = True
		else:
			local_maxi = np.zeros_like(distance, dtype='bool')
			local_maxi[tuple(local_maxi_loc.T)] = True

		if saturated_masks is not None and lab in saturated_masks:
			saturated_pixels = saturated_masks[lab]
			sat_labels, numfeatures = ndimage.label(saturated_pixels)
			for k in range(1, numfeatures+1):
				sp = saturated_pixels & (sat_labels == k)
				if np.sum(local_maxi & sp) > 1:
					imax = np.unravel_index(np.nanargmax(distance * local_maxi * sp), distance.shape)
					local_maxi[sp] = False
					local_maxi[imax] = True

		markers = ndimage.label(local_maxi)[0]

		if np.all(local_maxi == 0):
			Labels[class_member] = -1
		else:
			labels_ws = watershed(-distance0, markers, mask=Z)
			no_labels = len(set(labels_ws.flatten().tolist()))
			Labels[class_member] = -1
			idx = (labels_ws == 1) & (Z != 0)
			Labels[idx] = lab
			for u in range(no_labels-2):
				max_label += 1
				idx = (labels_ws == u+2) & (Z != 0)
				Labels[idx] = max_label

	out = np.asarray(Labels, dtype='int64')
	out[outside] = -2
	return out

#--------------------------------------------------------------------------------------------------
# saturated columns
#--------------------------------------------------------------------------------------------------
def k2p2_saturated(SumImage, MASKS, idx):
	"""``k2p2_saturated`` (k2p2v2.py:291-341)."""
	no_masks = MASKS.shape[0]
	column_mask = np.zeros_like(SumImage, dtype='bool')
	saturated_mask = np.zeros_like(MASKS, dtype='bool')
	pixels_added = 0
	with np.errstate(invalid='ignore', divide='ignore'):
		for u in range(no_masks):
			mask = np.asarray(MASKS[u, :, :], dtype='bool')
			mask_rows, mask_columns = np.where(mask)
			mv = SumImage[mask]
			mask_max = np.nan if np.all(np.isnan(mv)) else np.nanmax(mv)
			for c in sorted(set(mask_columns.tolist())):
				column_mask[:, c] = True
				pixels = SumImage[mask & column_mask]
				pmax = np.nan if np.all(np.isnan(pixels)) else np.nanmax(pixels)
				ratio = np.abs(nanmedian(np.diff(pixels)))/pmax
				if ratio < 0.01 and nanmedian(pixels) >= mask_max/2:
					add_to_mask = (idx & column_mask)
					new_mask_labels, numfeatures = ndimage.label(add_to_mask)
					imax = np.unravel_index(np.nanargmax(SumImage * mask * column_mask), SumImage.shape)
					add_to_mask &= (new_mask_labels == new_mask_labels[imax])
					pixels_added += np.sum(add_to_mask) - np.sum(mask[column_mask])
					saturated_mask[u][add_to_mask] = True
				column_mask[:, c] = False
	return saturated_mask, pixels_added

#--------------------------------------------------------------------------------------------------
# A2..A5 driver
#--------------------------------------------------------------------------------------------------
def k2p2FixFromSum(SumImage, thresh=1, min_no_pixels_in_mask=8, min_for_cluster=4, cluster_radius=np.sqrt(2),
	segmentation=True, ws_blur=0.5, ws_thres=0.05, ws_footprint=3, extend_overflow=True, catalog=None,
	cut_override=None, validate_bracket=False, full_output=False):
	"""
	``k2p2FixFromSum`` (k2p2v2.py:344-623; plotting omitted).

	``cut_override`` replaces the KDE/Powell threshold (used to test the integer part of the
	pipeline in isolation).  Returns ``(MASKS or None, bandwidth)``, or with ``full_output``
	``(MASKS, info-dict)``.
	"""
	if not (np.sqrt(2) <= cluster_radius < 2.0):
		raise NotImplementedError("grid DBSCAN is stated for sqrt(2) <= cluster_radius < 2")
	SumImage = np.asarray(SumImage, dtype='float64')
	NY, NX = SumImage.shape

	info = {}
	if cut_override is None:
		info = threshold(SumImage, thresh, validate_bracket=validate_bracket, full_output=True)
		CUT = info['CUT']
	else:
		CUT = cut_override
		info = {'CUT': CUT, 'bandwidth': np.nan}

	idx = np.zeros_like(SumImage, dtype='bool')
	np.greater(SumImage, CUT, out=idx, where=~np.isnan(SumImage))
	info['idx'] = idx

	if np.all(~idx):
		raise K2P2NoStars("No flux above threshold")

	labels_ini, core = dbscan_grid(idx, min_for_cluster)
	info['labels_ini'] = labels_ini
	info['core'] = core

	if segmentation and np.any(labels_ini[idx] != -1):
		dummy_labels = sorted(set(labels_ini[idx].tolist()) - {-1})
		DUMMY_MASKS = np.zeros((len(dummy_labels), NY, NX), dtype='bool')
		for u, lab in enumerate(dummy_labels):
			DUMMY_MASKS[u] = (labels_ini == lab)
		smask, _ = k2p2_saturated(SumImage, DUMMY_MASKS, idx)
		if np.any(smask):
			saturated_masks = {lab: smask[u] for u, lab in enumerate(dummy_labels)}
		else:
			saturated_masks = None
		labels = k2p2WS(SumImage, labels_ini, core, saturated_masks=saturated_masks,
			ws_thres=ws_thres, ws_footprint=ws_footprint, ws_blur=ws_blur, catalog=catalog)
	else:
		labels = labels_ini
	info['labels'] = labels

	unique_labels = sorted(set(labels[idx].tolist()))
	sizes = [(int(np.sum(labels[idx] == lab)), lab) for lab in unique_labels]
	sel = [(n, lab) for n, lab in sizes if n >= min_no_pixels_in_mask and lab != -1]
	no_masks = len(sel)

	if no_masks == 0:
		MASKS = None
	else:
		# Sort by number of pixels, largest first (argsort ascending, reversed: k2p2v2.py:535-536)
		order = np.argsort(np.array([n for n, _ in sel], dtype='float64'), kind='stable')[::-1]
		sel = [sel[i] for i in order]
		MASKS = np.zeros((no_masks, NY, NX))
		for u, (n, lab) in enumerate(sel):
			MASKS[u][idx & (labels == lab)] = 1

		# Fill holes (k2p2v2.py:549-554)
		pattern = np.array([[[0, 0.25, 0], [0.25, 0, 0.25], [0, 0.25, 0]]])
		mask_holes_indx = ndimage.convolve(MASKS, pattern, mode='constant', cval=0.0)
		mask_holes_indx = (mask_holes_indx > 0.95) & (MASKS == 0)
		if np.any(mask_holes_indx):
			MASKS[mask_holes_indx] = 1

		# Extend overflow lanes (k2p2v2.py:579-623)
		if extend_overflow:
			saturated_mask, pixels_added = k2p2_saturated(SumImage, MASKS, idx)
			if catalog is not None:
				c = np.asarray(np.round(catalog[:, 0]), dtype='int32')
				r = np.asarray(np.round(catalog[:, 1]), dtype='int32')
				tmag = catalog[:, 2]
				indx = (c >= 0) & (c < SumImage.shape[1]) & (r >= 0) & (r < SumImage.shape[0])
				c = c[indx]
				r = r[indx]
				tmag = tmag[indx]
				for u in range(no_masks):
					if np.any(saturated_mask[u, :, :]):
						which_stars = np.asarray(MASKS[u, :, :][r, c], dtype='bool')
						if np.any(which_stars):
							mags_in_mask = tmag[which_stars]
							mags_total = -2.5*np.log10(np.nansum(10**(-0.4*mags_in_mask)))
							if mags_total > saturation_limit:
								saturated_mask[u, :, :] = False
						else:
							saturated_mask[u, :, :] = False
			MASKS[saturated_mask] = 1

	if full_output:
		return MASKS, info
	return MASKS, info.get('bandwidth', np.nan)
