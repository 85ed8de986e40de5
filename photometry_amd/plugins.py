# -*- coding: utf-8 -*-
"""
The reference's plugin API (``BasePhotometry`` / ``AperturePhotometry`` / ``LinPSFPhotometry``,
photometry/BasePhotometry.py:62-1730, AperturePhotometry/photometry.py:17-257,
linpsf_photometry.py:40-219) on top of the HIP engine.

A plugin object handles ONE target exactly like the reference's: ``pho.photometry()`` ->
``do_photometry()`` fills ``self.lightcurve[...]``, ``final_phot_mask``, ``additional_headers``,
``_details`` and returns a :class:`STATUS`.  Inside, the target is a batch of one for the same
kernels the batched pipeline uses (``photometry_amd.pipeline``); there is no CPU implementation
of the numerics in this module -- without the HIP library / a GPU ``do_photometry`` raises, which
``tessphot.run_plugin`` turns into ``STATUS.ERROR`` with the traceback in ``_details['errors']``
like the reference does for any exception (tessphot.py:37-49).

File I/O (HDF5 cut-outs, SQLite catalogues, SPICE, WCS, FITS light curves) is replaced by a
``StampSource`` (``photometry_amd.source``) and a ``.npz`` light-curve writer; see DESIGN.md.
"""

import os
import logging
import configparser
import numpy as np
from .status import STATUS
from . import engine, pipeline, stamps

#: the values of the reference's photometry/data/settings.ini, used when no settings file is given
# ([fixes] time_offset of the reference's file is not among them: the timestamp correction of the early data releases is the
# input adapter's job here -- see the warning in BasePhotometry.__init__ -- and a switch that switches nothing would mislead)
DEFAULT_SETTINGS = {'todolist': {'faint_limit': '15.0'}, 'haloswitch': {'tmag_limit': '6.0', 'flux_limit': '0.01'}}

TESS_DEFAULT_BITMASK = engine.TESS_DEFAULT_BITMASK
#: PixelQualityFlags.BackgroundShenanigans / CorrectorQualityFlags.BackgroundShenanigans (photometry/quality.py:163, :85)
PIXEL_BACKGROUND_SHENANIGANS = 4
#: PROCVER card of the light-curve files this package writes
PROCVER = 'photometry_amd-0.2'
CORRECTOR_BACKGROUND_SHENANIGANS = 256


def load_settings(path=None):
	"""
	The pipeline settings (``io.load_settings``, photometry/io.py:96-107).  ``path`` or the environment variable
	``TESSPHOT_SETTINGS`` names the reference's own ``photometry/data/settings.ini`` -- a maintainer who drops this package into
	an installation points it there; otherwise the file's shipped values (:data:`DEFAULT_SETTINGS`) apply.
	"""
	s = configparser.ConfigParser()
	s.read_dict(DEFAULT_SETTINGS)
	path = path or os.environ.get('TESSPHOT_SETTINGS')
	if path:
		if not os.path.isfile(path):
			raise FileNotFoundError(f"settings file not found: {path}")
		s.read(path)
	return s


def mag2flux(mag, zp=20.451):
	"""photometry/utilities.py:134-149"""
	return np.clip(10**(-0.4*(mag - zp)), 0, None)


_PRF_CACHE = {} #: (psf_dir, sector, camera, ccd) -> PRFModel: the spline fits of a CCD are done once per process

class Table(object):
	"""Minimal column table (stand-in for the astropy Table used for ``catalog`` / ``lightcurve``)."""

	def __init__(self, **cols):
		self.cols = {k: np.asarray(v) for k, v in cols.items()}

	def __len__(self):
		return len(next(iter(self.cols.values()))) if self.cols else 0

	def __bool__(self):
		return True

	def __contains__(self, key):
		return key in self.cols

	def keys(self):
		return self.cols.keys()

	def __getitem__(self, key):
		if isinstance(key, str):
			return self.cols[key]
		if isinstance(key, (int, np.integer)):
			return {k: v[key] for k, v in self.cols.items()}
		key = np.asarray(key)
		if key.size == 0:
			key = key.astype('int64')
		return Table(**{k: v[key] for k, v in self.cols.items()})

	def __setitem__(self, key, value):
		self.cols[key] = np.asarray(value)

	def __iter__(self):
		for i in range(len(self)):
			yield {k: v[i] for k, v in self.cols.items()}


class _DetailLog(logging.Handler):
	"""Keeps the WARNING-and-above records of this package as ``"LEVEL: message"`` strings: the reference appends exactly
	such strings to ``_details['errors']`` (BasePhotometry.py:175-179, :1409-1414) and the halo switch matches on them."""

	def __init__(self, sink):
		super().__init__(level=logging.WARNING)
		self.sink = sink

	def emit(self, record):
		self.sink.append(f"{record.levelname}: {record.getMessage()}")


#--------------------------------------------------------------------------------------------------
class BasePhotometry(object):
	"""
	Same constructor signature, properties and methods as the reference's ``BasePhotometry``
	(BasePhotometry.py:100-101, :489-506, :521-706, :881-1258, :1291-1414).  ``input_folder`` is a
	``StampSource`` (see ``photometry_amd.source``) instead of a directory of HDF5 files.
	"""

	def __init__(self, starid, input_folder, output_folder, datasource='ffi',
		sector=None, camera=None, ccd=None, cadence=None, plot=False, cache='basic', version=6, ctx=None):
		logger = logging.getLogger(__name__)
		if datasource != 'ffi' and not datasource.startswith('tpf'):
			raise ValueError(f"Invalid datasource: '{datasource:s}'") # BasePhotometry.py:133-134
		if cache not in ('basic', 'none', 'full'):
			raise ValueError("Invalid cache: '{cache:s}'")
		self.starid = starid
		self.input_folder = input_folder
		self.output_folder_base = None if output_folder is None else os.path.abspath(output_folder)
		self.output_folder = self.output_folder_base
		self.plot = plot
		self.datasource = datasource
		self.version = version
		self._ctx = ctx
		self._own_ctx = False
		src = input_folder
		if not hasattr(src, 'cutout'):
			raise FileNotFoundError("input_folder must be a StampSource (HDF5/FITS input is not part of this engine, see DESIGN.md)")
		self.source = src
		self.sector = src.sector if sector is None else sector
		self.camera = src.camera if camera is None else camera
		self.ccd = src.ccd if ccd is None else ccd
		self.cadence = src.cadence if cadence is None else cadence
		self.n_readout = src.n_readout
		self.plot_folder = None

		self._status = STATUS.UNKNOWN
		self._details = {}
		self.method = {'BasePhotometry': 'base', 'AperturePhotometry': 'aperture', 'PSFPhotometry': 'psf',
			'LinPSFPhotometry': 'linpsf', 'HaloPhotometry': 'halo'}.get(self.__class__.__name__, None)
		logger.info('STARID = %d, DATASOURCE = %s, METHOD = %s', self.starid, self.datasource, self.method)

		# collect WARNING+ log records of this package like BasePhotometry.py:175-179
		self.message_queue = []
		self._handler = _DetailLog(self.message_queue)
		logging.getLogger('photometry_amd').addHandler(self._handler)

		self._settings = None
		tgt = src.target(starid)
		# tmag plus, when the source knows them, the catalogue properties the light-curve file carries (BasePhotometry.py:416-438)
		self.target = {k: v for k, v in tgt.items() if k not in ('starid', 'row', 'column')}
		self.ticver = getattr(src, 'ticver', None)
		#: header of the input file (BasePhotometry.py:262-270, :330): cosmic-ray mitigation keywords, DATA_REL, NUM_FRM
		self.header = dict(getattr(src, 'header', None) or {})
		self.data_rel = self.header.get('DATA_REL')
		default_frames = 900 if datasource == 'ffi' else 60 # BasePhotometry.py:269, :372
		self.num_frm = self.header.get('NUM_FRM', getattr(src, 'num_frm', None) or default_frames)
		self.target_pos_row = tgt['row']
		self.target_pos_column = tgt['column']
		self._max_stamp = tuple(src.max_stamp)
		self.pixel_offset_row = 0
		self.pixel_offset_col = 0

		self.Ntimes = len(src.time)
		self.lightcurve = Table(time=np.array(src.time, dtype='float64'), timecorr=np.array(src.timecorr, dtype='float64'),
			cadenceno=np.array(src.cadenceno), quality=np.array(src.quality),
			flux=np.zeros(self.Ntimes), flux_err=np.zeros(self.Ntimes), flux_background=np.zeros(self.Ntimes),
			pos_centroid=np.zeros((self.Ntimes, 2)), pos_corr=np.zeros((self.Ntimes, 2)))
		if getattr(src, 'jitter', None) is not None:
			self.lightcurve['pos_corr'] = np.array(src.jitter, dtype='float64')
		# The timestamp offset of the early data releases (fixes/time_offset.py upstream, applied at BasePhotometry.py:244 / :384)
		# belongs to the input adapter: a source hands over corrected times.  The reference applies
		# the correction silently (a DEBUG record, fixes/time_offset.py:125), so nothing may enter details['errors'] here: a source
		# that passes one of the data releases the reference WOULD correct (DATA_REL <= 26 always, 27 / 29 depending on PROCVER:
		# fixes/time_offset.py:99-120) without saying that it did is noted at INFO level, once per source.
		if self.data_rel is not None and not (self.header.get('TIME_OFFSET_CORRECTED') or getattr(src, 'time_offset_corrected', False)):
			try:
				affected = int(self.data_rel) <= 26 or int(self.data_rel) in (27, 29)
			except (TypeError, ValueError):
				affected = False
			if affected and not getattr(src, '_time_offset_noted', False):
				logger.info("Timestamps used as the source delivered them (DATA_REL = %s, no TIME_OFFSET_CORRECTED in its header): the "
					"time-offset correction of the early data releases is the input adapter's job.", self.data_rel)
				try:
					src._time_offset_noted = True
				except AttributeError:
					pass

		self.final_phot_mask = None
		self.final_position_mask = None
		self.additional_headers = {}
		self._stamp = None
		self.target_pos_column_stamp = None
		self.target_pos_row_stamp = None
		self._set_stamp()
		self._sumimage = None
		self._cubes = None
		self._aperture = None
		self._catalog = None
		self._psf = None
		self._settings = None

	# -- context manager / lifetime (BasePhotometry.py:489-506) -----------------------------------
	def __enter__(self):
		return self

	def __exit__(self, *args):
		self.close()

	def close(self):
		h = getattr(self, '_handler', None)
		if h is not None:
			logging.getLogger('photometry_amd').removeHandler(h)
			self._handler = None
		if getattr(self, '_own_ctx', False) and self._ctx is not None:
			self._ctx.close()
			self._ctx = None

	@property
	def ctx(self):
		"""The device context (created on first use; raises if there is no HIP library / GPU)."""
		if self._ctx is None:
			from .device import Context
			self._ctx = Context(0)
			self._own_ctx = True
		return self._ctx

	@property
	def status(self):
		return self._status

	# -- stamp logic (BasePhotometry.py:521-706): geometry in photometry_amd.stamps --------------------
	@property
	def _stamp_limits(self):
		"""The CCD region the source has data for, in stamp coordinates (``_max_stamp`` shifted by the pixel offsets)."""
		m = self._max_stamp
		return (m[0] + self.pixel_offset_row, m[1] + self.pixel_offset_row, m[2] + self.pixel_offset_col, m[3] + self.pixel_offset_col)

	def default_stamp(self):
		"""(rows, columns) of the default stamp for this target's magnitude (BasePhotometry.py:521-564)."""
		return stamps.default_stamp_size(self.target['tmag'])

	def resize_stamp(self, down=None, up=None, left=None, right=None, width=None, height=None):
		"""Grow / re-centre the stamp (BasePhotometry.py:567-613); ``True`` if it changed."""
		wanted = stamps.moved(self._stamp, self._stamp_limits, self.target_pos_row, self.target_pos_column,
			down=down, up=up, left=left, right=right, width=width, height=height)
		changed = self._adopt_stamp(wanted)
		if changed:
			self._details['stamp_resizes'] = self._details.get('stamp_resizes', 0) + 1
		return changed

	def _set_stamp(self, compare_stamp=None):
		"""First call: the default stamp (FFI) or the whole target pixel file (BasePhotometry.py:616-693)."""
		if not self._stamp:
			if self.datasource == 'ffi':
				n_rows, n_columns = self.default_stamp()
				wanted = stamps.centred_stamp(self.target_pos_row, self.target_pos_column, n_rows, n_columns)
			else:
				wanted = self._max_stamp
		else:
			wanted = self._stamp
		limits = self._stamp_limits if self.datasource == 'ffi' else self._max_stamp
		return self._adopt_stamp(stamps.clip_stamp(wanted, limits), compare_stamp)

	def _adopt_stamp(self, stamp, previous=None):
		"""Make ``stamp`` current; everything derived from the old cut-out is dropped.  ``False`` if nothing changed."""
		previous = self._stamp if previous is None else previous
		self._stamp = tuple(int(v) for v in stamp)
		self._details['stamp'] = self._stamp
		if previous and self._stamp == tuple(previous):
			return False
		self.target_pos_row_stamp = self.target_pos_row - self._stamp[0]
		self.target_pos_column_stamp = self.target_pos_column - self._stamp[2]
		self._sumimage = self._catalog = self._cubes = self._aperture = self._psf = None
		return True

	def get_pixel_grid(self):
		"""BasePhotometry.py:696-706: 1-based (cols, rows) mesh grid."""
		return np.meshgrid(
			np.arange(self._stamp[2]+1, self._stamp[3]+1, 1, dtype='int32'),
			np.arange(self._stamp[0]+1, self._stamp[1]+1, 1, dtype='int32')
		)

	@property
	def stamp(self):
		return self._stamp

	# -- data cubes (BasePhotometry.py:720-985) ----------------------------------------------------
	def _load(self):
		if self._cubes is None:
			self._cubes = self.source.cutout(self._stamp)
		return self._cubes

	@property
	def images_cube(self):
		return self._load()['images']

	@property
	def images_err_cube(self):
		return self._load()['images_err']

	@property
	def backgrounds_cube(self):
		return self._load()['backgrounds']

	@property
	def pixelflags_cube(self):
		"""BasePhotometry.py:832-877: uint8 ``(rows, cols, times)``; zeros when the source carries no pixel flags (:868-870)."""
		c = self._load()
		if 'pixel_flags' not in c:
			c['pixel_flags'] = np.zeros(c['images'].shape, dtype='uint8')
		return c['pixel_flags']

	@property
	def pixelflags(self):
		for k in range(self.Ntimes):
			yield self.pixelflags_cube[:, :, k]

	@property
	def images(self):
		for k in range(self.Ntimes):
			yield self.images_cube[:, :, k]

	@property
	def images_err(self):
		for k in range(self.Ntimes):
			yield self.images_err_cube[:, :, k]

	@property
	def backgrounds(self):
		for k in range(self.Ntimes):
			yield self.backgrounds_cube[:, :, k]

	@property
	def sumimage(self):
		"""BasePhotometry.py:989-1029, computed on the device (``tp_sumimage``)."""
		if self._sumimage is None:
			from .device import DeviceCube
			ctx = self.ctx
			cube = DeviceCube.from_host(ctx, self.images_cube)
			q = ctx.array(np.asarray(self.lightcurve['quality'], dtype='int32'))
			self._sumimage = engine.sumimage(ctx, cube, q).to_host()[0]
			cube.free()
		return self._sumimage

	@property
	def aperture(self):
		"""BasePhotometry.py:1033-1074 (FFI branch)."""
		if self._aperture is None:
			cols, rows = self.get_pixel_grid()
			ap = np.asarray(np.isfinite(self.sumimage), dtype='int32')
			ap[(45 <= cols) & (cols <= 556)] |= 32
			ap[(557 <= cols) & (cols <= 1068)] |= 64
			ap[(1069 <= cols) & (cols <= 1580)] |= 128
			ap[(1581 <= cols) & (cols <= 2092)] |= 256
			bpu = getattr(self.source, 'backgrounds_pixels_used', None)
			if bpu is not None: # pixels used for the background calculation (:1052-1061)
				r0, c0 = self.source.max_stamp[0], self.source.max_stamp[2]
				ap[bpu[self._stamp[0] - r0:self._stamp[1] - r0, self._stamp[2] - c0:self._stamp[3] - c0]] |= 4
			self._aperture = ap
		return self._aperture

	@property
	def settings(self):
		if self._settings is None:
			self._settings = load_settings()
		return self._settings

	@property
	def catalog(self):
		"""BasePhotometry.py:1094-1181: stars in the stamp (+5 px buffer), float32 pixel columns."""
		if self._catalog is None:
			c = self.source.catalog_in_stamp(self._stamp, buffer_size=5)
			col = np.asarray(c['column'], dtype='float64')
			row = np.asarray(c['row'], dtype='float64')
			self._catalog = Table(starid=np.asarray(c['starid'], dtype='int64'), tmag=np.asarray(c['tmag'], dtype='float32'),
				column=col.astype('float32'), row=row.astype('float32'),
				column_stamp=(col - self._stamp[2]).astype('float32'), row_stamp=(row - self._stamp[0]).astype('float32'))
		return self._catalog

	def catalog_attime(self, time):
		"""BasePhotometry.py:1224-1258 with a translation kernel: reference catalogue + jitter at ``time``."""
		jit = getattr(self.source, 'jitter', None)
		if jit is None:
			return self.catalog
		tref = np.asarray(self.lightcurve['time']) - np.asarray(self.lightcurve['timecorr'])
		k = int(np.argmin(np.abs(tref - time)))
		cat = Table(**{key: np.array(v, copy=True) for key, v in self.catalog.cols.items()})
		cat['column'] = cat['column'] + np.float32(jit[k, 0])
		cat['row'] = cat['row'] + np.float32(jit[k, 1])
		cat['column_stamp'] = cat['column_stamp'] + np.float32(jit[k, 0])
		cat['row_stamp'] = cat['row_stamp'] + np.float32(jit[k, 1])
		return cat

	@property
	def psf(self):
		"""The PRF model of the source (``photometry_amd.psf.PRFModel``); the reference builds a PSF object here."""
		if self._psf is None:
			self._psf = getattr(self.source, 'prf', None)
			if self._psf is None:
				# like psf.py:66-72: the SPOC PRF file of (sector, camera, CCD) from the data directory (TESSPHOT_PSF_DIR, or
				# the source's psf_dir; upstream it is photometry/data/psf inside the package)
				from . import psf as hpsf
				psf_dir = getattr(self.source, 'psf_dir', None) or os.environ.get('TESSPHOT_PSF_DIR')
				if not psf_dir:
					raise FileNotFoundError("the stamp source provides no PRF model and no PRF directory is configured (TESSPHOT_PSF_DIR)")
				key = (os.path.abspath(psf_dir), int(self.sector), int(self.camera), int(self.ccd))
				if key not in _PRF_CACHE:
					_PRF_CACHE[key] = hpsf.PRFModel.for_ccd(psf_dir, self.sector, self.camera, self.ccd)
				self._psf = _PRF_CACHE[key]
		return self._psf

	def delete_plots(self):
		pass

	def report_details(self, error=None, skip_targets=None):
		"""BasePhotometry.py:1291-1306"""
		if skip_targets is not None:
			self._details['skip_targets'] = skip_targets
		if error is not None:
			if 'errors' not in self._details:
				self._details['errors'] = []
			self._details['errors'].append(error)

	def do_photometry(self):
		raise NotImplementedError("You have to implement the actual lightcurve extraction yourself... Sorry!")

	# -- wrapper with diagnostics (BasePhotometry.py:1323-1414) --------------------------------------
	def photometry(self, *args, **kwargs):
		"""
		Run ``do_photometry`` and derive the light-curve diagnostics the scheduler stores (BasePhotometry.py:1337-1414):
		the reductions run on the device (``tp_lightcurve_diagnostics``, the same kernel the batch path uses).
		"""
		logger = logging.getLogger(__name__)
		self._status = self.do_photometry(*args, **kwargs)
		if self._status == STATUS.UNKNOWN:
			raise ValueError("STATUS was not set by do_photometry")
		if self._status in (STATUS.OK, STATUS.WARNING):
			d = self._device_diagnostics()
			problems = int(d['flags'])
			if problems & 1:
				raise ValueError("Final lightcurve fluxes are all NaNs")
			if problems & 2:
				raise ValueError("Final lightcurve errors are all NaNs")
			if problems & 4:
				raise ValueError("Invalid time-vector specified")
			if problems & 8:
				logger.warning("Could not detrend lightcurve for variability calculation.")
			for key in ('mean_flux', 'variance', 'rms_hour', 'ptp', 'variability'):
				self._details[key] = float(d[key])
			self._details['pos_centroid'] = np.array([d['pos_centroid_col'], d['pos_centroid_row']])
			if self.final_phot_mask is not None:
				self._details['mask_size'] = int(d['mask_size'])
				self._details['edge_flux'] = float(d['edge_flux'])
			if 'AP_CONT' in self.additional_headers:
				self._details['contamination'] = self.additional_headers['AP_CONT'][0]
		if self.message_queue:
			self._details.setdefault('errors', []).extend(self.message_queue)
			del self.message_queue[:]

	def _device_diagnostics(self):
		"""The diagnostics block as a dict (columns of ``engine.DIAGNOSTICS_COLUMNS``) for this one light curve."""
		ctx = self.ctx
		lc = self.lightcurve
		T = self.Ntimes
		block = np.zeros((5, 1, T))
		block[0, 0], block[1, 0] = lc['flux'], lc['flux_err']
		block[3, 0], block[4, 0] = lc['pos_centroid'][:, 0], lc['pos_centroid'][:, 1]
		dlc = engine.LightCurves(ctx, 1, T, block=ctx.array(block))
		mask = sumimage = None
		if self.final_phot_mask is not None:
			mask = ctx.array(np.asarray(self.final_phot_mask, dtype='uint8')[None])
			sumimage = ctx.array(np.asarray(self.sumimage, dtype='float64')[None])
		out = engine.lightcurve_diagnostics(ctx, dlc, ctx.array(np.asarray(lc['time'], dtype='float64')),
			ctx.array(np.asarray(lc['quality'], dtype='int32')), status=ctx.array(np.array([self._status.value], dtype='int32')),
			sumimage=sumimage, mask=mask).to_host()[0]
		return dict(zip(engine.DIAGNOSTICS_COLUMNS, out))

	def save_lightcurve(self, output_folder=None, version=None):
		"""
		FITS light-curve writer, FILEVER 1.5 (BasePhotometry.py:1417-1730) through :mod:`photometry_amd.fitsio` (astropy is
		not available here): primary header (:1463-1505), the ``LIGHTCURVE`` binary table with the 14 columns and their
		units / display formats (:1507-1600), the time keywords of the table header (:1602-1641), the ``SUMIMAGE`` and
		``APERTURE`` image extensions with bits 2 (photometry) and 8 (position) OR-ed into the aperture image (:1643-1665),
		``CHECKSUM`` / ``DATASUM`` (:1720), the reference's file name (:1708-1717).  Every card, comment, column definition and
		array is pinned by ``tests/golden/golden_fitsfile.*`` (the reference's own function recorded card by card).
		The WCS keywords of the image extensions come from the source (``wcs_header(stamp)``; the reference slices an astropy
		``WCS``, :1659-1661); a source without one gets the stamp limits written instead (``STAMP_*``).  ``DATE-OBS`` /
		``DATE-END`` are converted from TDB by :func:`fitsio.tdb_to_utc_isot`.
		"""
		import datetime
		from . import fitsio
		from .fitsio import card
		if output_folder is None:
			output_folder = self.output_folder
		if version is None:
			if self.version is None:
				raise ValueError("VERSION has not been set")
			version = self.version
		os.makedirs(output_folder, exist_ok=True)
		cadence = int(self.cadence) if self.cadence else 1800
		data_rel = int(self.data_rel or 0)
		SumImage = np.asarray(self.sumimage, dtype='float64')
		lc = self.lightcurve
		# Background Shenanigans anywhere in the stamp at a timestamp -> CorrectorQualityFlags.BackgroundShenanigans (:1445-1449)
		quality = np.zeros(len(lc['time']), dtype='int32')
		quality[np.any(self.pixelflags_cube & PIXEL_BACKGROUND_SHENANIGANS != 0, axis=(0, 1))] |= CORRECTOR_BACKGROUND_SHENANIGANS
		# timestamps without a defined time are dropped (:1451-1452); upstream leaves QUALITY at its full length there, which no
		# table can hold -- it is cut to the same rows here
		indx = np.isfinite(lc['time'])
		quality = quality[indx]
		col = {k: np.asarray(lc[k])[indx] for k in lc.keys()}
		tgt = self.target
		undef = fitsio.Undefined()

		def opt(key): # "undefined" card for a missing or zero value, as upstream writes it (:1487-1490)
			v = tgt.get(key)
			return undef if not v else v
		if tgt.get('pm_ra') is None or tgt.get('pm_decl') is None:
			pmtotal = undef
		else:
			pmtotal = np.sqrt(tgt['pm_ra']**2 + tgt['pm_decl']**2)
		hdr = self.header
		prim = [
			card('NEXTEND', 3, 'number of standard extensions'), card('EXTNAME', 'PRIMARY', 'name of extension'),
			card('ORIGIN', 'TASOC/Aarhus', 'institution responsible for creating this file'),
			card('DATE', datetime.datetime.now().strftime("%Y-%m-%d"), 'date the file was created'),
			card('TELESCOP', 'TESS', 'telescope'), card('INSTRUME', 'TESS Photometer', 'detector type'),
			card('FILTER', 'TESS', 'Photometric bandpass filter'), card('OBJECT', f"TIC {self.starid:d}", 'string version of TICID'),
			card('TICID', int(self.starid), 'unique TESS target identifier'), card('CAMERA', int(self.camera), 'Camera number'),
			card('CCD', int(self.ccd), 'CCD number'), card('SECTOR', int(self.sector), 'Observing sector'),
			card('PROCVER', PROCVER, 'Version of photometry pipeline'), card('FILEVER', '1.5', 'File format version'),
			card('DATA_REL', self.data_rel, 'Data release number'), card('VERSION', int(version), 'Version of the processing'),
			card('PHOTMET', self.method, 'Photometric method used'),
			card('RADESYS', 'ICRS', 'reference frame of celestial coordinates'), card('EQUINOX', 2000.0, 'equinox of celestial coordinate system'),
			card('RA_OBJ', tgt.get('ra_J2000'), '[deg] Right ascension'), card('DEC_OBJ', tgt.get('decl_J2000'), '[deg] Declination'),
			card('PMRA', opt('pm_ra'), '[mas/yr] RA proper motion'), card('PMDEC', opt('pm_decl'), '[mas/yr] Dec proper motion'),
			card('PMTOTAL', pmtotal, '[mas/yr] total proper motion'),
			card('TESSMAG', float(tgt['tmag']), '[mag] TESS magnitude'), card('TEFF', opt('teff'), '[K] Effective temperature'),
			card('TICVER', self.ticver, 'TESS Input Catalog version'),
			card('CRMITEN', hdr.get('CRMITEN'), 'spacecraft cosmic ray mitigation enabled'),
			card('CRBLKSZ', hdr.get('CRBLKSZ'), '[exposures] s/c cosmic ray mitigation block siz'),
			card('CRSPOC', hdr.get('CRSPOC'), 'SPOC cosmic ray cleaning enabled'),
		]
		for key, value in self.additional_headers.items(): # K2P2 settings, AP_CONT, ... (:1497-1499)
			if isinstance(value, (tuple, list)):
				prim.append(card(key, value[0], value[1] if len(value) > 1 else None))
			else:
				prim.append(card(key, value))
		prim.append(card('DATAVAL', 0, 'Data validation flags'))

		n = len(col['time'])
		nan = np.full(n, np.nan)
		f64 = 'column format: 64-bit floating point'
		i32 = 'column format: signed 32-bit integer'
		disp = 'column display format'

		def column(name, fmt, arr, unit=None, dsp=None, title=None, unitc=None):
			return {'name': name, 'format': fmt, 'array': arr, 'unit': unit, 'disp': dsp,
				'comments': {'TTYPE': title, 'TFORM': f64 if fmt == 'D' else ('column format: 32-bit floating point' if fmt == 'E' else i32),
					'TUNIT': unitc, 'TDISP': disp}}
		columns = [
			column('TIME', 'D', col['time'], 'BJD - 2457000, days', 'D14.7', 'column title: data time stamps', 'column units: Barycenter corrected TESS Julian'),
			column('TIMECORR', 'E', col['timecorr'], 'd', 'E13.6', 'column title: barycenter - timeslice correction', 'column units: day'),
			column('CADENCENO', 'J', col['cadenceno'], None, 'I10', 'column title: unique cadence number'),
			column('FLUX_RAW', 'D', col['flux'], 'e-/s', 'E26.17', 'column title: photometric flux', 'column units: electrons per second'),
			column('FLUX_RAW_ERR', 'D', col['flux_err'], 'e-/s', 'E26.17', 'column title: photometric flux error', 'column units: electrons per second'),
			column('FLUX_BKG', 'D', col['flux_background'], 'e-/s', 'E26.17', 'column title: photometric background flux', 'column units: electrons per second'),
			column('FLUX_CORR', 'D', nan, 'ppm', 'E26.17', 'column title: corrected photometric flux', 'column units: rel. flux in parts-per-million'),
			column('FLUX_CORR_ERR', 'D', nan, 'ppm', 'E26.17', 'column title: corrected photometric flux error', 'column units: parts-per-million'),
			column('QUALITY', 'J', quality, None, 'B16.16', 'column title: photometry quality flags'),
			column('PIXEL_QUALITY', 'J', col['quality'], None, 'B16.16', 'column title: pixel quality flags'),
			column('MOM_CENTR1', 'D', col['pos_centroid'][:, 0], 'pixels', 'F10.5', 'column title: moment-derived column centroid', 'column units: pixels'),
			column('MOM_CENTR2', 'D', col['pos_centroid'][:, 1], 'pixels', 'F10.5', 'column title: moment-derived row centroid', 'column units: pixels'),
			column('POS_CORR1', 'D', col['pos_corr'][:, 0], 'pixels', 'F14.7', 'column title: column position correction', 'column units: pixels'),
			column('POS_CORR2', 'D', col['pos_corr'][:, 1], 'pixels', 'F14.7', 'column title: row position correction', 'column units: pixels'),
		]
		# time keywords (:1602-1641)
		tdel = cadence / 86400
		tstart = float(col['time'][0] - tdel/2) if n else float('nan')
		tstop = float(col['time'][-1] + tdel/2) if n else float('nan')
		telapse = tstop - tstart
		frametime, int_time, readtime = 2.0, 1.98, 0.02
		if hdr.get('CRMITEN'):
			crblocksize = hdr['CRBLKSZ']
			deadc = (int_time * (crblocksize-2)/crblocksize) / frametime
		else:
			deadc = int_time / frametime
		num_frm = self.num_frm
		bjdref = 2457000
		tcards = [
			card('INHERIT', True, 'inherit the primary header'),
			card('TIMEREF', 'SOLARSYSTEM', 'barycentric correction applied to times'),
			card('TIMESYS', 'TDB', 'time system is Barycentric Dynamical Time (TDB)'),
			card('BJDREFI', bjdref, 'integer part of BTJD reference date'), card('BJDREFF', 0.0, 'fraction of the day in BTJD reference date'),
			card('TIMEUNIT', 'd', 'time unit for TIME, TSTART and TSTOP'),
			card('TSTART', tstart, 'observation start time in BTJD'), card('TSTOP', tstop, 'observation stop time in BTJD'),
			card('DATE-OBS', fitsio.tdb_to_utc_isot(tstart, bjdref) if n else None, 'TSTART as UTC calendar date'),
			card('DATE-END', fitsio.tdb_to_utc_isot(tstop, bjdref) if n else None, 'TSTOP as UTC calendar date'),
			card('MJD-BEG', tstart + (bjdref - 2400000.5), 'observation start time in MJD'),
			card('MJD-END', tstop + (bjdref - 2400000.5), 'observation start time in MJD'),
			card('TELAPSE', telapse, '[d] TSTOP - TSTART'), card('LIVETIME', telapse*deadc, '[d] TELAPSE multiplied by DEADC'),
			card('DEADC', deadc, 'deadtime correction'), card('EXPOSURE', telapse*deadc, '[d] time on source'),
			card('XPOSURE', frametime*deadc*num_frm, '[s] Duration of exposure'),
			card('TIMEPIXR', 0.5, 'bin time beginning=0 middle=0.5 end=1'), card('TIMEDEL', tdel, '[d] time resolution of data'),
			card('INT_TIME', int_time, '[s] photon accumulation time per frame'), card('READTIME', readtime, '[s] readout time per frame'),
			card('FRAMETIM', frametime, '[s] frame time (INT_TIME + READTIME)'), card('NUM_FRM', num_frm, 'number of frames per time stamp'),
			card('NREADOUT', int(self.n_readout), 'number of read per cadence'),
		]
		mask = np.array(self.aperture, dtype='int32', copy=True) # :1643-1649
		if self.final_phot_mask is not None:
			mask[self.final_phot_mask] |= 2
		if self.final_position_mask is not None:
			mask[self.final_position_mask] |= 8
		# image extensions: the WCS of the stamp (:1651-1661) when the source has one, then INHERIT
		wcs_header = getattr(self.source, 'wcs_header', None)
		if wcs_header is not None:
			icards = [card(*c) for c in wcs_header(self._stamp)]
		else:
			icards = [card('STAMP_R1', int(self._stamp[0]), 'first CCD row of the stamp'), card('STAMP_R2', int(self._stamp[1]), 'last CCD row + 1'),
				card('STAMP_C1', int(self._stamp[2]), 'first CCD column of the stamp'), card('STAMP_C2', int(self._stamp[3]), 'last CCD column + 1')]
		icards.append(card('INHERIT', True, 'inherit the primary header'))
		fname = (f'tess{self.starid:011d}-s{int(self.sector):03d}-{int(self.camera):d}-{int(self.ccd):d}-c{cadence:04d}'
			f'-dr{data_rel:02d}-v{int(version):02d}-tasoc_lc.fits.gz')
		path = os.path.join(output_folder, fname)
		fitsio.write(path, [fitsio.primary_hdu(prim), fitsio.bintable_hdu('LIGHTCURVE', columns, tcards),
			fitsio.image_hdu('SUMIMAGE', SumImage, icards), fitsio.image_hdu('APERTURE', mask, icards)])
		# relative to the input folder when the output lies inside it, else to the base output folder (:1722-1726)
		base = self.output_folder_base
		inp = self.input_folder if isinstance(self.input_folder, (str, bytes, os.PathLike)) else None
		if inp is not None and os.path.realpath(output_folder).startswith(os.path.realpath(inp)):
			base = os.path.abspath(inp)
		self._details['filepath_lightcurve'] = os.path.relpath(path, base).replace('\\', '/')
		return path


#--------------------------------------------------------------------------------------------------
class _OneTargetScene(object):
	"""Adapter: a plugin object as a batch of one for ``pipeline.ApertureBatch``."""
	def __init__(self, pho):
		cubes = pho._load()
		self.n_targets = 1
		H, W, T = cubes['images'].shape
		self.n_cad, self.height, self.width = T, H, W
		self.images = cubes['images'][None]
		self.images_err = cubes['images_err'][None]
		self.backgrounds = cubes['backgrounds'][None]
		self.quality = np.asarray(pho.lightcurve['quality'], dtype='int32')
		self.time = np.asarray(pho.lightcurve['time'], dtype='float64')
		self.stamps = np.asarray([pho._stamp], dtype='int32')
		cat = pho.catalog
		self.cat_offsets = np.array([0, len(cat)], dtype='int64')
		self.catalog = {k: cat[k] for k in ('starid', 'tmag', 'row', 'column', 'row_stamp', 'column_stamp')}
		self.target_pos_row = np.array([pho.target_pos_row])
		self.target_pos_column = np.array([pho.target_pos_column])
		self.target_tmag = np.array([pho.target['tmag']])
		self.target_starid = np.array([pho.starid], dtype='int64')
		self.aperture = None
		self.cadence_s = pho.cadence if pho.cadence else 1800


#: the K2P2 settings of the aperture plugin (photometry.py:54-64) and the FITS keywords they are reported under (:203-211)
K2P2_SETTINGS = {'thresh': 0.8, 'min_no_pixels_in_mask': 4, 'min_for_cluster': 4,
	'cluster_radius': np.sqrt(2) + np.finfo(np.float64).eps, 'segmentation': True, 'ws_blur': 0.5,
	'ws_thres': 0, 'ws_footprint': 3, 'extend_overflow': True}
K2P2_HEADERS = (('KP_THRES', 'thresh', 'K2P2 sum-image threshold'), ('KP_MIPIX', 'min_no_pixels_in_mask', 'K2P2 min pixels in mask'),
	('KP_MICLS', 'min_for_cluster', 'K2P2 min pix. for cluster'), ('KP_CLSRA', 'cluster_radius', 'K2P2 cluster radius'),
	('KP_WS', 'segmentation', 'K2P2 watershed segmentation'), ('KP_WSBLR', 'ws_blur', 'K2P2 watershed blur'),
	('KP_WSTHR', 'ws_thres', 'K2P2 watershed threshold'), ('KP_WSFOT', 'ws_footprint', 'K2P2 watershed footprint'),
	('KP_EX', 'extend_overflow', 'K2P2 extend overflow'))

#: error kinds of the device flags (include/tessphot_hip.h) that are uncaught exceptions in the reference's plugin
_MASK_EXCEPTIONS = {1: "K2P2NoFlux: No measured flux in sum-image", 2: "Selected KDE bandwidth is 0. Cannot estimate density.",
	3: "attempt to get argmin of an empty sequence", 4: "index out of bounds for the target pixel"}


def mask_outcome(flags, logger):
	"""
	Log what the reference's plugin logs for the mask stage of one attempt and return ``'error'`` (the plugin returns
	STATUS.ERROR: too many masks, photometry.py:113-115), or ``None`` to go on; raises for the conditions that are uncaught
	exceptions upstream (they end as STATUS.ERROR through ``tessphot.run_plugin``).
	"""
	kind = flags >> 8
	if flags & 32:
		logger.error('No flux above threshold.')
	if flags & 1:
		logger.warning("No masks found. Using minimum aperture." if flags & (32 | 64) else 'No mask found for main target. Using minimum aperture.')
	if kind == 5:
		logger.error('Too many masks.')
		return 'error'
	if kind in _MASK_EXCEPTIONS:
		raise RuntimeError(_MASK_EXCEPTIONS[kind])
	return None


class AperturePhotometry(BasePhotometry):
	"""
	Simple aperture photometry with K2P2 masks (AperturePhotometry/photometry.py:17-257).  One attempt -- sum image, mask
	creation and selection, extraction, contamination -- is one pass of the fused device kernel over the current cut-out; the
	loop around it grows the stamp while the mask touches an edge (:75-170, decisions in :mod:`photometry_amd.stamps`).
	"""

	def _attempt(self):
		"""One pass over the current stamp: device results of this target as a dict of host arrays."""
		res = pipeline.run_aperture(self.ctx, _OneTargetScene(self), cubes='host', diagnostics=False)
		self._sumimage = res['sumimage'][0]
		return res

	def do_photometry(self):
		logger = logging.getLogger(__name__)
		logger.info("Running aperture photometry...")
		tmag = self.target['tmag']
		bright = tmag <= self.settings.getfloat('haloswitch', 'tmag_limit') and not self.datasource.startswith('tpf:')
		flux_budget = self.settings.getfloat('haloswitch', 'flux_limit') * mag2flux(tmag)

		attempts_left = stamps.retry_limit(tmag)
		while True:
			res = self._attempt()
			attempts_left -= 1
			flags = int(res['flags'][0])
			if mask_outcome(flags, logger) == 'error':
				return STATUS.ERROR
			mask_main = res['mask'][0].astype(bool)
			wanted = stamps.edge_requests(flags)
			if not wanted:
				break
			logger.info("Touching the edges! Retrying.")
			before, sumimage_before = self._stamp, self._sumimage
			if not self.resize_stamp(**wanted):
				self._sumimage = sumimage_before      # nothing changed: the attempt just made stands
				logger.warning("Could not resize stamp any further.")
				break
			if bright:
				stuck_flux = stamps.quick_break_flux(sumimage_before, mask_main, before, self._stamp, wanted)
				if stuck_flux is not None and stuck_flux > flux_budget:
					logger.error('Stamp resize hit limit. Haloswitch quick break.')
					self._details['edge_flux'] = stuck_flux
					return STATUS.ERROR
			if attempts_left == 0:
				logger.error('Too many stamp resizes.')
				return STATUS.ERROR

		lc = self.lightcurve
		for key in ('flux', 'flux_err', 'flux_background', 'pos_centroid'):
			lc[key] = res[key][0]
		self.final_phot_mask = self.final_position_mask = mask_main
		for key, name, comment in K2P2_HEADERS:
			value = K2P2_SETTINGS[name]
			self.additional_headers[key] = (bool(value) if isinstance(value, bool) else value, comment)

		status = STATUS.OK
		contamination = float(res['contamination'][0])
		if flags >> 8 == 6:      # no catalogue star inside the mask (photometry.py:243-246)
			logger.error("No targets in mask.")
			contamination, status = np.nan, STATUS.ERROR
		logger.info("Contamination: %f", contamination)
		if not np.isnan(contamination):
			self.additional_headers['AP_CONT'] = (contamination, 'AP contamination')
		inside = res['cat_in_mask'][:len(self.catalog)].astype(bool)
		others = [int(sid) for sid in self.catalog['starid'][inside] if sid != self.starid]
		if others:
			logger.info("These stars could be skipped: %s", others)
			self.report_details(skip_targets=others)
		return STATUS.WARNING if flags & 1 else status


class LinPSFPhotometry(BasePhotometry):
	"""Linear PSF photometry (linpsf_photometry.py:40-219) on the device."""

	def __init__(self, *args, **kwargs):
		super().__init__(*args, **kwargs)
		self.cutoff_radius = 5

	def do_photometry(self):
		from . import psf as hpsf
		from .device import DeviceCube
		logger = logging.getLogger(__name__)
		ctx = self.ctx
		model = self.psf
		cat = self.catalog
		catalog = {k: cat[k] for k in ('starid', 'tmag', 'row_stamp', 'column_stamp')}
		sel, star_offsets, target_index = hpsf.select_stars(catalog, np.array([0, len(cat)]), np.array([self.starid]))
		nfit = int(star_offsets[-1])
		T = self.Ntimes
		pos_row = np.empty((nfit, T))
		pos_col = np.empty((nfit, T))
		tref = np.asarray(self.lightcurve['time']) - np.asarray(self.lightcurve['timecorr'])
		for k in range(T):
			ck = self.catalog_attime(tref[k])
			pos_row[:, k] = ck['row_stamp'][sel]
			pos_col[:, k] = ck['column_stamp'][sel]
		coef = engine.linpsf_prf(ctx, ctx.array(model.base_coef), ctx.array(model.weights(np.asarray([self._stamp]))))
		cube = DeviceCube.from_host(ctx, self.images_cube)
		res = engine.linpsf_fit(ctx, cube, coef, ctx.array(model.tx), ctx.array(model.ty), ctx.array(star_offsets), ctx.array(target_index),
			ctx.array(pos_row), ctx.array(pos_col), max(nfit, 1), cutoff_radius=self.cutoff_radius).to_host()
		self.lightcurve['flux'] = res['flux'][0]
		self.lightcurve['flux_err'] = res['flux_err'][0]
		status = int(res['status'][0])
		if status == 2:
			self.report_details(error='All target flux values are NaN.')
			return STATUS.ERROR
		contamination = float(res['contamination'][0])
		logger.info("Contamination: %f", contamination)
		self.additional_headers['PSF_CONT'] = (contamination, 'PSF contamination')
		if contamination > 0.1:
			self.report_details(error='High contamination')
			return STATUS.WARNING
		return STATUS.OK


def psf_star_selection(row_stamp, column_stamp, tmag, target_row_stamp, target_column_stamp, target_tmag):
	"""
	The stars PSFPhotometry fits (psf_photometry.py:117-130): closer than 5 pixels to the target position and not more than
	5 magnitudes fainter, the five closest, nearest first.  Returns their indices into the catalogue arrays.
	"""
	dist = np.sqrt((target_row_stamp - np.asarray(row_stamp))**2 + (target_column_stamp - np.asarray(column_stamp))**2)
	near = np.flatnonzero((dist < 5) & (target_tmag - np.asarray(tmag) > -5))
	return near[np.argsort(dist[near], kind='stable')][:5]


class PSFPhotometry(BasePhotometry):
	"""
	Non-linear PSF photometry (psf_photometry.py:19-196): per cadence a Nelder-Mead fit of position and flux of the target and
	its up to four nearest neighbours against the pixel-integrated PRF, on the device (``tp_psf_fit``; the cadences of a target
	form a warm-start chain, a batch of targets runs side by side).
	"""
	available = True

	def __init__(self, *args, **kwargs):
		super().__init__(*args, **kwargs)
		self.cutoff_radius = 5
		# statistics of the read noise term of the likelihood (BasePhotometry.py:267-270: header values or these defaults)
		self.readnoise = getattr(self.source, 'readnoise', 10)
		self.gain = getattr(self.source, 'gain', 100)

	def _minimum_aperture(self):
		"""psf_photometry.py:29-41"""
		cols, rows = self.get_pixel_grid()
		near = (np.abs(cols - self.target_pos_column - 1) <= 1) & (np.abs(rows - self.target_pos_row - 1) <= 1)
		return near & (self.aperture & 1 != 0)

	def do_photometry(self):
		from .device import DeviceCube
		ctx = self.ctx
		model = self.psf
		cat = self.catalog
		sel = psf_star_selection(cat['row_stamp'], cat['column_stamp'], cat['tmag'], self.target_pos_row_stamp,
			self.target_pos_column_stamp, self.target['tmag'])
		params0 = np.column_stack((np.asarray(cat['row_stamp'][sel], dtype='float64'), np.asarray(cat['column_stamp'][sel], dtype='float64'),
			mag2flux(np.asarray(cat['tmag'][sel], dtype='float64'))))
		coef = engine.linpsf_prf(ctx, ctx.array(model.base_coef), ctx.array(model.weights(np.asarray([self._stamp]))))
		res = engine.psf_fit(ctx, DeviceCube.from_host(ctx, self.images_cube), DeviceCube.from_host(ctx, self.backgrounds_cube), coef,
			ctx.array(model.tx), ctx.array(model.ty), ctx.array(np.array([0, len(sel)], dtype='int64')), ctx.array(params0),
			ctx.array(self._minimum_aperture().astype('uint8')[None]), variance_floor=self.n_readout * self.readnoise**2 / self.gain**2,
			cutoff_radius=self.cutoff_radius)
		lc = self.lightcurve
		lc['flux'] = res['flux'].to_host()[0]
		lc['flux_err'] = res['flux_err'].to_host()[0]
		lc['pos_centroid'] = np.column_stack((res['centroid_row'].to_host()[0], res['centroid_col'].to_host()[0]))   # (row, column) as upstream (:176)
		if np.any(np.isnan(lc['flux'])):
			logging.getLogger(__name__).warning("We should flag that this has not gone well.")
		return STATUS.OK


class HaloPhotometry(BasePhotometry):
	"""Halo photometry (halo/halo_photometry.py, third-party halophot) -- not part of this engine."""
	available = False

	def do_photometry(self):
		raise NotImplementedError("HaloPhotometry is outside the hot path implemented by photometry_amd")
