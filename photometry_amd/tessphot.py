# -*- coding: utf-8 -*-
"""
``tessphot(method, **task)`` -- the dispatch function the schedulers call
(photometry/tessphot.py:20-135; call sites run_tessphot.py:153, run_tessphot_mpi.py:178), with the
same signature, return value and error conventions, plus ``tessphot_batch`` for throughput.
"""

import logging
import traceback
import numpy as np
from .status import STATUS
from .plugins import AperturePhotometry, PSFPhotometry, LinPSFPhotometry, HaloPhotometry, load_settings, mag2flux


class _PhotErrorDummy(object):
	"""tessphot.py:13-17"""
	def __init__(self, traceback, *args, **kwargs):
		self.status = STATUS.ERROR
		self.method = 'error'
		self._details = {'errors': traceback} if traceback else {}


def _try_photometry(PhotClass, *args, **kwargs):
	"""tessphot.py:20-49: any exception -> STATUS.ERROR with the traceback in the details."""
	logger = logging.getLogger(__name__)
	tbcollect = []
	pho = None
	try:
		with PhotClass(*args, **kwargs) as pho:
			pho.photometry()
			if pho.status in (STATUS.OK, STATUS.WARNING):
				pho.save_lightcurve()
	except (KeyboardInterrupt, SystemExit): # pragma: no cover
		logger.info("Stopped by user or system")
		try:
			pho._status = STATUS.ABORT
		except: # noqa: E722
			pass
	except: # noqa: E722
		logger.exception("Something happened")
		tb = traceback.format_exc().strip()
		try:
			pho._status = STATUS.ERROR
			pho.report_details(error=tb)
		except: # noqa: E722
			tbcollect.append(tb)
	if pho is None:
		return _PhotErrorDummy(tbcollect, *args, **kwargs)
	return pho


def tessphot(method=None, *args, **kwargs):
	"""
	Run the photometry pipeline on a single star (tessphot.py:52-135).

	``method``: ``'aperture'``, ``'halo'``, ``'psf'``, ``'linpsf'`` or ``None`` (aperture first, then the
	halo switch for bright targets with flux on the stamp edge, :76-109).  Raises ``ValueError`` on an
	invalid method; returns the photometry object.
	"""
	logger = logging.getLogger(__name__)
	if method is None:
		pho = _try_photometry(AperturePhotometry, *args, **kwargs)
		settings = load_settings()
		haloswitch_tmag_limit = settings.getfloat('haloswitch', 'tmag_limit')
		haloswitch_flux_limit = settings.getfloat('haloswitch', 'flux_limit')
		if not isinstance(pho, _PhotErrorDummy) and pho.target['tmag'] <= haloswitch_tmag_limit \
			and not pho.datasource.startswith('tpf:'):
			EdgeFlux = pho._details.get('edge_flux')
			errors = pho._details.get('errors', [])
			if pho.status == STATUS.ERROR \
				and ('Too many stamp resizes.' in errors or 'Stamp resize hit limit. Haloswitch quick break.' in errors):
				logger.warning("Too many stamp resizes. Let us try Halo instead.")
				pho = _try_photometry(HaloPhotometry, *args, **kwargs)
			elif EdgeFlux is not None:
				ExpectedFlux = mag2flux(pho.target['tmag'])
				if EdgeFlux/ExpectedFlux > haloswitch_flux_limit:
					logger.warning("Target is still touching the edge. Let us try Halo instead.")
					pho = _try_photometry(HaloPhotometry, *args, **kwargs)
			if isinstance(pho, HaloPhotometry):
				pho.report_details('Automatically switched to Halo photometry')
				pho._details['edge_flux'] = EdgeFlux
		if pho.status == STATUS.WARNING:
			logger.warning("Do something else?")
	else:
		try:
			PhotClass = {
				'aperture': AperturePhotometry,
				'psf': PSFPhotometry,
				'linpsf': LinPSFPhotometry,
				'halo': HaloPhotometry
			}[method]
		except KeyError:
			raise ValueError(f"Invalid method: '{method:s}'")
		pho = _try_photometry(PhotClass, *args, **kwargs)
	logger.info("Done")
	return pho


class BatchResult(object):
	"""What the scheduler consumes per target (taskmanager.py:444-563): status, method, details (+ the light curve)."""
	def __init__(self, starid, status, method, details, lightcurve, mask):
		self.starid = starid
		self.status = status
		self.method = method
		self._details = details
		self.lightcurve = lightcurve
		self.final_phot_mask = mask


def tessphot_batch(ctx, scene, cubes='host'):
	"""
	Aperture photometry of a whole batch of fixed-size stamps in one pass over the device
	(``pipeline.run_aperture``); returns one :class:`BatchResult` per target, in order.
	``scene`` carries the arrays of ``photometry_amd.simulate.Scene`` (cubes, catalogue, positions).
	"""
	from . import pipeline
	from .engine import DIAGNOSTICS_COLUMNS
	res = pipeline.run_aperture(ctx, scene, cubes=cubes)
	out = []
	for i in range(scene.n_targets):
		status = STATUS(int(res['status'][i]))
		flags = int(res['flags'][i])
		a, b = scene.cat_offsets[i], scene.cat_offsets[i+1]
		inm = res['cat_in_mask'][a:b].astype(bool)
		ids = scene.catalog['starid'][a:b][inm]
		details = {'stamp': tuple(int(v) for v in scene.stamps[i]), 'mask_size': int(res['mask'][i].sum())}
		skip = [int(s) for s in ids if s != scene.target_starid[i]]
		if skip:
			details['skip_targets'] = skip
		if not np.isnan(res['contamination'][i]):
			details['contamination'] = float(res['contamination'][i])
		if flags >> 8:
			details['errors'] = [f'ERROR: aperture mask creation failed (kind {flags >> 8})']
		if flags & 30:
			details['edge'] = flags & 30
		if status in (STATUS.OK, STATUS.WARNING):
			# BasePhotometry.py:1343-1407, computed on the device for the whole batch
			d = dict(zip(DIAGNOSTICS_COLUMNS, res['diagnostics'][i]))
			dflags = int(d['flags'])
			if dflags & 3: # the reference raises ValueError -> STATUS.ERROR through tessphot.py:37-49
				status = STATUS.ERROR
				details.setdefault('errors', []).append('ValueError: Final lightcurve fluxes are all NaNs' if dflags & 1
					else 'ValueError: Final lightcurve errors are all NaNs')
			elif dflags & 4:
				status = STATUS.ERROR
				details.setdefault('errors', []).append('ValueError: Invalid time-vector specified')
			else:
				for key in ('mean_flux', 'variance', 'rms_hour', 'ptp', 'variability', 'edge_flux'):
					details[key] = float(d[key])
				details['pos_centroid'] = np.array([d['pos_centroid_col'], d['pos_centroid_row']])
				if dflags & 8:
					details.setdefault('errors', []).append('WARNING: Could not detrend lightcurve for variability calculation.')
		lc = {k: res[k][i] for k in ('flux', 'flux_err', 'flux_background', 'pos_centroid')}
		out.append(BatchResult(int(scene.target_starid[i]), status, 'aperture', details, lc, res['mask'][i].astype(bool)))
	return out
