# -*- coding: utf-8 -*-
"""
ORACLE (test infrastructure only) -- light-curve diagnostics computed right after the hot path.

Follows ``BasePhotometry.photometry`` (photometry/BasePhotometry.py:1343-1407) and
``utilities.rms_timescale`` (photometry/utilities.py:227-264): the per-target reductions over the
light curve that feed the ``diagnostics`` table of the scheduler (taskmanager.py:543-563).
Pinned by ``tests/golden/golden_diagnostics.npz`` (the reference's own ``photometry()`` executed on
the same light curves, bottleneck reductions shimmed by numpy's).
"""

import warnings
import numpy as np
from .quality import TESS_DEFAULT_BITMASK
from .utilities import rms_timescale

#: order of the columns of the device output block ``diag[n_targets][10]``
COLUMNS = ('mean_flux', 'variance', 'rms_hour', 'ptp', 'pos_centroid_col', 'pos_centroid_row', 'variability',
	'mask_size', 'edge_flux', 'flags')

FLAG_ALLNAN_FLUX = 1      # "Final lightcurve fluxes are all NaNs"   (BasePhotometry.py:1346-1347) -> ValueError upstream
FLAG_ALLNAN_ERR = 2       # "Final lightcurve errors are all NaNs"   (:1348-1349)
FLAG_BAD_TIME = 4         # rms_timescale: invalid time vector        (utilities.py:248-254) -> ValueError upstream
FLAG_NO_DETREND = 8       # "Could not detrend lightcurve for variability calculation." (:1386-1391): detrend = 0


def diagnostics(time, quality, flux, flux_err, pos_centroid, sumimage=None, mask=None, bitmask=TESS_DEFAULT_BITMASK):
	"""
	BasePhotometry.py:1343-1401 for one target with status OK / WARNING.

	Returns a dict with the keys of :data:`COLUMNS` (``flags`` as int).  Where the reference raises
	``ValueError`` the corresponding flag is set and the remaining values are NaN.
	"""
	out = {k: np.nan for k in COLUMNS}
	out['flags'] = 0
	flux = np.asarray(flux, dtype='float64')
	flux_err = np.asarray(flux_err, dtype='float64')
	time = np.asarray(time, dtype='float64')
	pos_centroid = np.asarray(pos_centroid, dtype='float64')
	if mask is not None: # :1394-1403 (independent of the light curve)
		mask = np.asarray(mask, dtype=bool)
		out['mask_size'] = float(int(np.sum(mask)))
		edge = np.zeros_like(mask, dtype=bool)
		edge[:, (0, -1)] = True
		edge[(0, -1), 1:-1] = True
		out['edge_flux'] = float(np.nansum(np.asarray(sumimage)[mask & edge]))
	if np.all(np.isnan(flux)): # :1346
		out['flags'] |= FLAG_ALLNAN_FLUX
		return out
	if np.all(np.isnan(flux_err)): # :1348
		out['flags'] |= FLAG_ALLNAN_ERR
		return out
	good = (np.asarray(quality) & bitmask) == 0 # TESSQualityFlags.filter, :1353
	gflux, gerr, gtime, gcen = flux[good], flux_err[good], time[good], pos_centroid[good]
	with warnings.catch_warnings(), np.errstate(invalid='ignore', divide='ignore'):
		warnings.simplefilter('ignore')
		mean_flux = np.nanmedian(gflux) if len(gflux) else np.nan # :1357
		out['mean_flux'] = mean_flux
		rel = (gflux / mean_flux) - 1 # :1360
		rel_err = np.abs(1/mean_flux) * gerr # :1361
		out['variance'] = np.nanvar(rel, ddof=1) # :1364
		try:
			out['rms_hour'] = rms_timescale(gtime, rel, timescale=3600/86400) # :1365
		except ValueError:
			out['flags'] |= FLAG_BAD_TIME
		out['ptp'] = np.nanmedian(np.abs(np.diff(rel))) if len(rel) > 1 else np.nan # :1366
		cen = np.nanmedian(gcen, axis=0) if len(gcen) else np.array([np.nan, np.nan]) # :1369
		out['pos_centroid_col'], out['pos_centroid_row'] = cen[0], cen[1]
		indx = np.isfinite(gtime) & np.isfinite(rel) & np.isfinite(rel_err) # :1372
		detrend = 0
		if np.any(indx):
			mintime = np.nanmin(gtime[indx])
			with warnings.catch_warnings():
				warnings.filterwarnings('error', category=np.exceptions.RankWarning)
				try:
					p = np.polyfit(gtime[indx] - mintime, rel[indx], 3, w=1/rel_err[indx]) # :1382
					detrend = np.polyval(p, gtime - mintime)
				except np.exceptions.RankWarning:
					out['flags'] |= FLAG_NO_DETREND
		else:
			out['flags'] |= FLAG_NO_DETREND
		out['variability'] = np.nanstd(rel - detrend) / np.nanmedian(rel_err) # :1393
	return out
