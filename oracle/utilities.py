# -*- coding: utf-8 -*-
"""
ORACLE (test infrastructure only) -- small utilities used on the hot path.

Follows ``photometry/utilities.py``: ``mad_to_sigma`` (:25), ``move_median_central``
(:52-62), ``integratedGaussian`` (:100-131), ``mag2flux`` (:134-149),
``rms_timescale`` (:227-264).
"""

import numpy as np
from scipy.special import erf
from scipy.stats import binned_statistic

#: photometry/utilities.py:25
mad_to_sigma = 1.482602218505602


def mag2flux(mag, zp=20.451):
	"""photometry/utilities.py:134-149"""
	return np.clip(10**(-0.4*(mag - zp)), 0, None)


def integratedGaussian(x, y, flux, x_0, y_0, sigma=1):
	"""photometry/utilities.py:100-131"""
	denom = np.sqrt(2) * sigma
	return (flux / 4 * ((erf((x - x_0 + 0.5) / denom)
		- erf((x - x_0 - 0.5) / denom)) * (erf((y - y_0 + 0.5) / denom)
		- erf((y - y_0 - 0.5) / denom))))


def _move_median(x, window, min_count):
	"""bottleneck.move_median (trailing window, NaN-aware), used by utilities.py:53."""
	x = np.asarray(x, dtype='float64')
	y = np.full_like(x, np.nan)
	for i in range(len(x)):
		w = x[max(0, i - window + 1):i + 1]
		w = w[~np.isnan(w)]
		if len(w) >= min_count:
			y[i] = np.median(w)
	return y


def _nanmedian(x):
	x = np.asarray(x, dtype='float64')
	x = x[~np.isnan(x)]
	return np.median(x) if len(x) else np.nan


def _move_median_central_1d(x, width_points):
	"""photometry/utilities.py:52-58"""
	y = _move_median(x, width_points, min_count=1)
	y = np.roll(y, -width_points//2+1)
	for k in range(width_points//2+1):
		y[k] = _nanmedian(x[:(k+2)])
		y[-(k+1)] = _nanmedian(x[-(k+2):])
	return y


def move_median_central(x, width_points, axis=0):
	"""photometry/utilities.py:61-62"""
	return np.apply_along_axis(_move_median_central_1d, axis, x, width_points)


def rms_timescale(time, flux, timescale=3600/86400):
	"""photometry/utilities.py:227-264"""
	time = np.asarray(time)
	flux = np.asarray(flux)
	if len(flux) == 0 or np.all(np.isnan(flux)):
		return np.nan
	if len(time) == 0 or np.all(np.isnan(time)):
		raise ValueError("Invalid time-vector specified. No valid timestamps.")
	time_min = np.nanmin(time)
	time_max = np.nanmax(time)
	if not np.isfinite(time_min) or not np.isfinite(time_max) or time_max - time_min <= 0:
		raise ValueError("Invalid time-vector specified")
	bins = np.arange(time_min, time_max, timescale)
	bins = np.append(bins, time_max)
	indx = np.isfinite(flux)
	flux_bin, _, _ = binned_statistic(time[indx], flux[indx], np.nanmean, bins=bins)
	return mad_to_sigma * _nanmedian(np.abs(flux_bin - _nanmedian(flux_bin)))
