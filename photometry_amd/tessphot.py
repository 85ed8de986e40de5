# -*- coding: utf-8 -*-
"""
``tessphot(method, **task)`` -- the entry point the schedulers call once per task (reference: photometry/tessphot.py:52-135;
call sites run_tessphot.py:153, run_tessphot_mpi.py:178) with the contract they rely on:

* ``method`` in ``{None, 'aperture', 'psf', 'linpsf', 'halo'}``, anything else -> ``ValueError("Invalid method: '...'")``;
* the return value is always an object with ``status``, ``method`` and ``_details`` -- also when the plugin could not even be
  constructed (then ``method == 'error'``, ``status == STATUS.ERROR``, the traceback in ``_details['errors']``);
* an exception anywhere in the plugin becomes ``STATUS.ERROR`` with the traceback appended to the details; ``KeyboardInterrupt`` /
  ``SystemExit`` become ``STATUS.ABORT``; a light curve is saved only for ``OK`` / ``WARNING``;
* ``method=None`` runs aperture photometry and then asks whether a bright target (``Tmag <= haloswitch.tmag_limit``, not a
  secondary TPF target) should have been done with Halo photometry instead: either the aperture run gave up on stamp resizes, or
  the final mask still carries more than ``haloswitch.flux_limit`` of the expected flux on the stamp edge.

``tessphot_batch`` / ``tessphot_frames`` are the throughput entries (not in the reference): a whole batch per device pass.
"""

import logging
import traceback
import numpy as np
from .status import STATUS
from .plugins import AperturePhotometry, PSFPhotometry, LinPSFPhotometry, HaloPhotometry, load_settings, mag2flux

#: method keyword -> plugin class (tessphot.py:122-127)
PLUGINS = {'aperture': AperturePhotometry, 'psf': PSFPhotometry, 'linpsf': LinPSFPhotometry, 'halo': HaloPhotometry}

#: the two ways the aperture plugin reports that it ran out of stamp (photometry.py:156, :168), as they appear in the details
_RESIZE_GAVE_UP = ('Too many stamp resizes.', 'Stamp resize hit limit. Haloswitch quick break.')


class FailedTask(object):
	"""Stands in for the photometry object when its constructor raised: what the scheduler can still store."""
	method = 'error'

	def __init__(self, tracebacks):
		self.status = STATUS.ERROR
		self._details = {'errors': list(tracebacks)} if tracebacks else {}


def run_plugin(plugin, *args, before_save=None, **kwargs):
	"""Construct ``plugin``, run its photometry inside its context manager, save on success; never raises (see module doc).
	``before_save(pho)``: called after the photometry and before the light curve is written (a last word on status and details,
	so that the file on disk and the returned object agree)."""
	logger = logging.getLogger(__name__)
	pho, lost = None, []
	try:
		pho = plugin(*args, **kwargs)
		with pho:
			pho.photometry()
			if before_save is not None:
				before_save(pho)
			if pho.status in (STATUS.OK, STATUS.WARNING):
				pho.save_lightcurve()
	except (KeyboardInterrupt, SystemExit): # pragma: no cover
		logger.info("Stopped by user or system")
		if pho is not None:
			pho._status = STATUS.ABORT
	except BaseException: # noqa: B902 -- the scheduler must get a result for every task
		logger.exception("Something happened")
		text = traceback.format_exc().strip()
		if pho is not None:
			pho._status = STATUS.ERROR
			pho.report_details(error=text)
		else:
			lost.append(text)
	return FailedTask(lost) if pho is None else pho


def halo_switch_reason(pho, settings=None):
	"""
	Why the aperture result ``pho`` of a bright target should be redone with Halo photometry, or ``None``
	(the predicate of tessphot.py:81-102).
	"""
	if isinstance(pho, FailedTask):
		return None
	settings = load_settings() if settings is None else settings
	if pho.target['tmag'] > settings.getfloat('haloswitch', 'tmag_limit') or pho.datasource.startswith('tpf:'):
		return None
	messages = pho._details.get('errors', [])
	if pho.status == STATUS.ERROR and any(m in messages for m in _RESIZE_GAVE_UP):
		return "Too many stamp resizes. Let us try Halo instead."
	edge_flux = pho._details.get('edge_flux')
	if edge_flux is not None and edge_flux / mag2flux(pho.target['tmag']) > settings.getfloat('haloswitch', 'flux_limit'):
		return "Target is still touching the edge. Let us try Halo instead."
	return None


def tessphot(method=None, *args, **kwargs):
	"""Run the photometry pipeline on a single star; see the module documentation for the contract."""
	logger = logging.getLogger(__name__)
	if method is not None:
		if method not in PLUGINS:
			raise ValueError(f"Invalid method: '{method:s}'")
		pho = run_plugin(PLUGINS[method], *args, **kwargs)
	else:
		def keep_aperture_result(p):
			# No Halo photometry in this engine (third-party halophot upstream): the finished aperture result is kept, not
			# thrown away for a plugin that can only fail; a good light curve is downgraded to WARNING and the request recorded
			# BEFORE the light curve is written, so that the file and the returned status agree.
			why = halo_switch_reason(p)
			if why is not None and not HaloPhotometry.available:
				p.report_details(error='Halo switch requested (' + why + ') but Halo photometry is not available: aperture result kept')
				if p.status == STATUS.OK:
					p._status = STATUS.WARNING

		pho = run_plugin(AperturePhotometry, *args, before_save=keep_aperture_result, **kwargs)
		reason = halo_switch_reason(pho)
		if reason is not None:
			logger.warning(reason)
			edge_flux = pho._details.get('edge_flux')
			if HaloPhotometry.available:
				pho = run_plugin(HaloPhotometry, *args, **kwargs)
				if isinstance(pho, HaloPhotometry):
					# keep the diagnostics that led to the switch (tessphot.py:104-109)
					pho.report_details('Automatically switched to Halo photometry')
					pho._details['edge_flux'] = edge_flux
		if pho.status == STATUS.WARNING:
			logger.warning("Do something else?")
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


def _diagnostics_into(details, d, status):
	"""Copy the device diagnostics into ``details`` the way ``BasePhotometry.photometry`` does (BasePhotometry.py:1343-1407);
	returns the status, turned into ERROR where the reference raises ``ValueError``."""
	problems = int(d['flags'])
	if problems & 3:
		details.setdefault('errors', []).append('ValueError: Final lightcurve fluxes are all NaNs' if problems & 1
			else 'ValueError: Final lightcurve errors are all NaNs')
		return STATUS.ERROR
	if problems & 4:
		details.setdefault('errors', []).append('ValueError: Invalid time-vector specified')
		return STATUS.ERROR
	for key in ('mean_flux', 'variance', 'rms_hour', 'ptp', 'variability', 'edge_flux'):
		details[key] = float(d[key])
	details['pos_centroid'] = np.array([d['pos_centroid_col'], d['pos_centroid_row']])
	if problems & 8:
		details.setdefault('errors', []).append('WARNING: Could not detrend lightcurve for variability calculation.')
	if problems & 16:
		# the device's hourly binning gave up (unsorted or extremely sparse time axis): rms_hour is NaN -- say so
		details.setdefault('errors', []).append('WARNING: rms_hour not computed: the time axis could not be binned on the device.')
	return status


class BatchResults(object):
	"""
	What :func:`tessphot_frames` returns: the results of a whole batch in columns (``status`` int32 with the reference's STATUS
	integers -- what ``todolist.status`` stores, taskmanager.py:538-541 --, ``starid``, ``stamp``, ``stamp_resizes`` and, through
	:meth:`column`, ``mask_size`` / ``contamination`` / the diagnostics), and the per-target objects a scheduler's ``save_result``
	takes only when asked for: ``results[i]`` (or iteration) builds the :class:`BatchResult` of target ``i``.
	"""

	def __init__(self, frames_result, starid):
		self.frames = frames_result
		self.starid = np.asarray(starid, dtype='int64')
		fr = frames_result
		status = fr.status.copy()
		# BasePhotometry.photometry (BasePhotometry.py:1343-1407): the diagnostics' ValueErrors turn OK / WARNING into ERROR
		self._problems = np.zeros(len(fr), dtype='int64')
		sel = np.flatnonzero(fr.has_result & ((status == STATUS.OK.value) | (status == STATUS.WARNING.value)))
		if len(sel):
			self._problems[sel] = fr.column('flags', fill=0)[sel].astype('int64')
			status[sel[(self._problems[sel] & 7) != 0]] = STATUS.ERROR.value
		self.status = status

	def __len__(self):
		return len(self.frames)

	def column(self, name):
		"""Per-target float64 column: ``mask_size``, ``contamination`` or one of ``engine.DIAGNOSTICS_COLUMNS``; NaN without a result."""
		return self.frames.column(name)

	@property
	def stamp(self):
		return self.frames.stamp

	@property
	def stamp_resizes(self):
		return self.frames.stamp_resizes

	def __getitem__(self, i):
		r = self.frames[i]
		status = STATUS(r['status'])
		details = {'stamp': r.get('stamp'), 'stamp_resizes': r['stamp_resizes']}
		if r['errors']:
			details['errors'] = list(r['errors'])
		if 'edge_flux' in r:
			details['edge_flux'] = r['edge_flux']
		lc = mask = None
		if 'mask' in r:
			mask = r['mask']
			details['mask_size'] = int(mask.sum())
			if r['skip_targets']:
				details['skip_targets'] = r['skip_targets']
			if not np.isnan(r['contamination']):
				details['contamination'] = r['contamination']
			lc = {k: r[k] for k in ('flux', 'flux_err', 'flux_background', 'pos_centroid')}
			if status in (STATUS.OK, STATUS.WARNING):
				status = _diagnostics_into(details, r['diagnostics'], status)
		return BatchResult(int(self.starid[int(i)]), status, 'aperture', details, lc, mask)

	def __iter__(self):
		return (self[i] for i in range(len(self)))


def tessphot_frames(ctx, stack, targets, catalog, time, quality, settings=None, cadence_s=1800, engine='native'):
	"""
	Aperture photometry of every target of a CCD region resident in HBM (:class:`photometry_amd.pipeline.FrameStack`), stamp
	resizes included: what ``tessphot('aperture', ...)`` returns per target, for the whole batch in a few device passes.
	Returns a :class:`BatchResults`: columns for the whole batch, one :class:`BatchResult` per target on demand (``results[i]``).
	"""
	from . import pipeline
	res = pipeline.aperture_frames(ctx, stack, targets, catalog, time, quality, settings=settings, cadence_s=cadence_s, engine=engine)
	return BatchResults(res, targets['starid'])


def tessphot_frames_pipelined(ctx, stack, batches, catalog, time, quality, settings=None, cadence_s=1800, in_flight=4, engine='native'):
	"""
	:func:`tessphot_frames` over consecutive batches of targets of one CCD region -- what a run over a whole CCD does, a few
	thousand targets per call -- with ``in_flight`` batches on the device at a time (``pipeline.aperture_frames_pipelined``: the
	first round of a batch runs under the latency-bound resize rounds of the one before it).  ``batches``: an iterable of
	``targets`` dicts; yields one :class:`BatchResults` per batch, in order, equal to what a call of its own returns.
	"""
	from . import pipeline
	batches = list(batches) if not hasattr(batches, '__next__') else batches
	ids = []
	def feed():
		for t in batches:
			ids.append(t['starid'])
			yield t
	for k, res in enumerate(pipeline.aperture_frames_pipelined(ctx, stack, feed(), catalog, time, quality, settings=settings,
			cadence_s=cadence_s, in_flight=in_flight, engine=engine)):
		yield BatchResults(res, ids[k])


def tessphot_batch(ctx, scene, cubes='host'):
	"""
	Aperture photometry of a whole batch of FIXED-size stamp cubes in one pass over the device
	(``pipeline.run_aperture``); returns one :class:`BatchResult` per target, in order.
	``scene`` carries the arrays of ``photometry_amd.simulate.Scene`` (cubes, catalogue, positions).  A cube cannot grow:
	a mask that touches its edge is used as it is, like the reference does when ``resize_stamp`` returns False
	(BasePhotometry.py:605-612), and reported in ``details['edge']``; use :func:`tessphot_frames` when the frames the
	stamps were cut from are available, so that such targets get the reference's stamp-resize retries.
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
			details.setdefault('errors', []).append('WARNING: Could not resize stamp any further.')  # photometry.py:141-144
		if status in (STATUS.OK, STATUS.WARNING):
			status = _diagnostics_into(details, dict(zip(DIAGNOSTICS_COLUMNS, res['diagnostics'][i])), status)
		lc = {k: res[k][i] for k in ('flux', 'flux_err', 'flux_background', 'pos_centroid')}
		out.append(BatchResult(int(scene.target_starid[i]), status, 'aperture', details, lc, res['mask'][i].astype(bool)))
	return out
