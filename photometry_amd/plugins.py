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
``tessphot._try_photometry`` turns into ``STATUS.ERROR`` with the traceback in ``_details['errors']``
like the reference does for any exception (tessphot.py:37-49).

File I/O (HDF5 cut-outs, SQLite catalogues, SPICE, WCS, FITS light curves) is replaced by a
``StampSource`` (``photometry_amd.source``) and a ``.npz`` light-curve writer; see DESIGN.md.
"""

import os
import logging
import configparser
import numpy as np
from .status import STATUS
from . import engine, pipeline

#: photometry/data/settings.ini of the reference
DEFAULT_SETTINGS = {'todolist': {'faint_limit': '15.0'}, 'fixes': {'time_offset': 'True'},
	'haloswitch': {'tmag_limit': '6.0', 'flux_limit': '0.01'}}

TESS_DEFAULT_BITMASK = engine.TESS_DEFAULT_BITMASK
mad_to_sigma = 1.482602218505602 #: photometry/utilities.py:25


def load_settings():
	"""io.load_settings (photometry/io.py:96-107) with the reference's defaults."""
	s = configparser.ConfigParser()
	s.read_dict(DEFAULT_SETTINGS)
	return s


def mag2flux(mag, zp=20.451):
	"""photometry/utilities.py:134-149"""
	return np.clip(10**(-0.4*(mag - zp)), 0, None)


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


class ListHandler(logging.Handler):
	"""utilities.ListHandler (photometry/utilities.py:439-458): collects ``"LEVEL: msg"`` strings."""
	def __init__(self, message_queue, *args, **kwargs):
		super().__init__(*args, **kwargs)
		self.message_queue = message_queue

	def emit(self, record):
		self.message_queue.append(record.levelname + ': ' + record.getMessage())


def rms_timescale(time, flux, timescale=3600/86400):
	"""utilities.rms_timescale (photometry/utilities.py:227-264)."""
	from scipy.stats import binned_statistic
	time, flux = np.asarray(time), np.asarray(flux)
	if len(flux) == 0 or np.all(np.isnan(flux)):
		return np.nan
	if len(time) == 0 or np.all(np.isnan(time)):
		raise ValueError("Invalid time-vector specified. No valid timestamps.")
	time_min, time_max = np.nanmin(time), np.nanmax(time)
	if not np.isfinite(time_min) or not np.isfinite(time_max) or time_max - time_min <= 0:
		raise ValueError("Invalid time-vector specified")
	bins = np.append(np.arange(time_min, time_max, timescale), time_max)
	indx = np.isfinite(flux)
	flux_bin, _, _ = binned_statistic(time[indx], flux[indx], np.nanmean, bins=bins)
	med = np.nanmedian(flux_bin) if np.any(np.isfinite(flux_bin)) else np.nan
	return mad_to_sigma * np.nanmedian(np.abs(flux_bin - med))


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
		self._handler = ListHandler(message_queue=self.message_queue, level=logging.WARNING)
		logging.getLogger('photometry_amd').addHandler(self._handler)

		tgt = src.target(starid)
		self.target = {'tmag': tgt['tmag']}
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

	# -- stamp logic (BasePhotometry.py:521-706) ---------------------------------------------------
	def default_stamp(self):
		"""BasePhotometry.py:521-564: stamp size as a function of Tmag, at least 15x15."""
		tmag = np.array([0.0, 0.52631579, 1.05263158, 1.57894737, 2.10526316,
			2.63157895, 3.15789474, 3.68421053, 4.21052632, 4.73684211,
			5.26315789, 5.78947368, 6.31578947, 6.84210526, 7.36842105,
			7.89473684, 8.42105263, 8.94736842, 9.47368421, 10.0, 13.0])
		height = np.array([831.98319063, 533.58494422, 344.0840884, 223.73963332,
			147.31365728, 98.77856016, 67.95585074, 48.38157414,
			35.95072974, 28.05639497, 23.043017, 19.85922009,
			17.83731732, 16.5532873, 15.73785092, 15.21999971,
			14.89113301, 14.68228285, 14.54965042, 14.46542084, 14.0])
		width = np.array([157.71602062, 125.1238281, 99.99440209, 80.61896267,
			65.6799962, 54.16166547, 45.28073365, 38.4333048,
			33.15375951, 28.05639497, 23.043017, 19.85922009,
			17.83731732, 16.5532873, 15.73785092, 15.21999971,
			14.89113301, 14.68228285, 14.54965042, 14.46542084, 14.0])
		Ncolumns = np.interp(self.target['tmag'], tmag, width)
		Nrows = np.interp(self.target['tmag'], tmag, height)
		Nrows = np.maximum(np.ceil(Nrows), 15)
		Ncolumns = np.maximum(np.ceil(Ncolumns), 15)
		return Nrows, Ncolumns

	def resize_stamp(self, down=None, up=None, left=None, right=None, width=None, height=None):
		"""BasePhotometry.py:567-613"""
		old_stamp = self._stamp
		st = list(self._stamp)
		if up:
			st[1] += up
		if down:
			st[0] -= down
		if left:
			st[2] -= left
		if right:
			st[3] += right
		if height:
			st[0] = int(np.round(self.target_pos_row)) - height//2
			st[1] = int(np.round(self.target_pos_row)) + height//2 + 1
		if width:
			st[2] = int(np.round(self.target_pos_column)) - width//2
			st[3] = int(np.round(self.target_pos_column)) + width//2 + 1
		self._stamp = tuple(st)
		stamp_changed = self._set_stamp(compare_stamp=old_stamp)
		if stamp_changed:
			self._details['stamp_resizes'] = self._details.get('stamp_resizes', 0) + 1
		return stamp_changed

	def _set_stamp(self, compare_stamp=None):
		"""BasePhotometry.py:616-693"""
		if not self._stamp:
			if self.datasource == 'ffi':
				Nrows, Ncolumns = self.default_stamp()
				self._stamp = (
					int(np.round(self.target_pos_row)) - Nrows//2,
					int(np.round(self.target_pos_row)) + Nrows//2 + 1,
					int(np.round(self.target_pos_column)) - Ncolumns//2,
					int(np.round(self.target_pos_column)) + Ncolumns//2 + 1
				)
			else:
				self._stamp = self._max_stamp
		st = list(self._stamp)
		st[0] = int(np.maximum(st[0], self._max_stamp[0] + self.pixel_offset_row))
		st[1] = int(np.minimum(st[1], self._max_stamp[1] + self.pixel_offset_row))
		st[2] = int(np.maximum(st[2], self._max_stamp[2] + self.pixel_offset_col))
		st[3] = int(np.minimum(st[3], self._max_stamp[3] + self.pixel_offset_col))
		self._stamp = tuple(st)
		if self._stamp[0] > self._stamp[1] or self._stamp[2] > self._stamp[3]:
			raise ValueError("Invalid stamp selected")
		self._details['stamp'] = self._stamp
		if self._stamp == compare_stamp:
			return False
		self.target_pos_row_stamp = self.target_pos_row - self._stamp[0]
		self.target_pos_column_stamp = self.target_pos_column - self._stamp[2]
		self._sumimage = None
		self._catalog = None
		self._cubes = None
		self._aperture = None
		self._psf = None
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
				raise FileNotFoundError("the stamp source provides no PRF model")
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
		logger = logging.getLogger(__name__)
		self._status = self.do_photometry(*args, **kwargs)
		if self._status == STATUS.UNKNOWN:
			raise ValueError("STATUS was not set by do_photometry")
		if self._status in (STATUS.OK, STATUS.WARNING):
			lc = self.lightcurve
			if np.all(np.isnan(lc['flux'])):
				raise ValueError("Final lightcurve fluxes are all NaNs")
			if np.all(np.isnan(lc['flux_err'])):
				raise ValueError("Final lightcurve errors are all NaNs")
			indx_good = (np.asarray(lc['quality']) & TESS_DEFAULT_BITMASK) == 0
			gflux, gerr, gtime = lc['flux'][indx_good], lc['flux_err'][indx_good], lc['time'][indx_good]
			with np.errstate(invalid='ignore', divide='ignore'):
				self._details['mean_flux'] = np.nanmedian(gflux)
				flux = (gflux / self._details['mean_flux']) - 1
				flux_err = np.abs(1/self._details['mean_flux']) * gerr
				self._details['variance'] = np.nanvar(flux, ddof=1)
				self._details['rms_hour'] = rms_timescale(gtime, flux, timescale=3600/86400)
				self._details['ptp'] = np.nanmedian(np.abs(np.diff(flux)))
				self._details['pos_centroid'] = np.nanmedian(lc['pos_centroid'][indx_good], axis=0)
				indx = np.isfinite(gtime) & np.isfinite(flux) & np.isfinite(flux_err)
				detrend = 0
				if np.any(indx):
					mintime = np.nanmin(gtime[indx])
					try:
						p = np.polyfit(gtime[indx] - mintime, flux[indx], 3, w=1/flux_err[indx])
						detrend = np.polyval(p, gtime - mintime)
					except Exception: # noqa: B902  (np.RankWarning / LinAlgError -> no detrending, BasePhotometry.py:1385-1387)
						logger.warning("Could not detrend lightcurve for variability calculation.")
				else:
					logger.warning("Could not detrend lightcurve for variability calculation.")
				self._details['variability'] = np.nanstd(flux - detrend) / np.nanmedian(flux_err)
			if self.final_phot_mask is not None:
				self._details['mask_size'] = int(np.sum(self.final_phot_mask))
				edge = np.zeros_like(self.sumimage, dtype='bool')
				edge[:, (0, -1)] = True
				edge[(0, -1), 1:-1] = True
				self._details['edge_flux'] = np.nansum(self.sumimage[self.final_phot_mask & edge])
			if self.additional_headers and 'AP_CONT' in self.additional_headers:
				self._details['contamination'] = self.additional_headers['AP_CONT'][0]
		if self.message_queue:
			if not self._details.get('errors'):
				self._details['errors'] = []
			self._details['errors'] += self.message_queue
			self.message_queue.clear()

	def save_lightcurve(self, output_folder=None, version=None):
		"""
		FITS light-curve writer, FILEVER 1.5 (BasePhotometry.py:1417-1730) through :mod:`photometry_amd.fitsio` (astropy is
		not available here): primary header (:1463-1505), the ``LIGHTCURVE`` binary table with the 14 columns and their
		units / display formats (:1507-1600), the time keywords of the table header (:1602-1641), the ``SUMIMAGE`` and
		``APERTURE`` image extensions with bits 2 (photometry) and 8 (position) OR-ed into the aperture image (:1643-1665),
		``CHECKSUM`` / ``DATASUM`` (:1720), the reference's file name (:1708-1717).  Not written: the WCS keywords of the
		image extensions and ``DATE-OBS`` / ``DATE-END`` (astropy ``WCS`` / ``Time`` upstream).
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
		data_rel = int(getattr(self, 'data_rel', 0) or 0)
		SumImage = np.asarray(self.sumimage, dtype='float64')
		lc = self.lightcurve
		indx = np.isfinite(lc['time']) # :1451-1452
		col = {k: np.asarray(lc[k])[indx] for k in lc.keys()}
		tgt = self.target
		undef = fitsio.Undefined()

		def opt(key):
			v = tgt.get(key) if isinstance(tgt, dict) else None
			return undef if not v else v

		prim = [
			card('NEXTEND', 3, 'number of standard extensions'), card('EXTNAME', 'PRIMARY', 'name of extension'),
			card('ORIGIN', 'TASOC/Aarhus', 'institution responsible for creating this file'),
			card('DATE', datetime.datetime.now().strftime("%Y-%m-%d"), 'date the file was created'),
			card('TELESCOP', 'TESS', 'telescope'), card('INSTRUME', 'TESS Photometer', 'detector type'),
			card('FILTER', 'TESS', 'Photometric bandpass filter'), card('OBJECT', f"TIC {self.starid:d}", 'string version of TICID'),
			card('TICID', int(self.starid), 'unique TESS target identifier'), card('CAMERA', int(self.camera), 'Camera number'),
			card('CCD', int(self.ccd), 'CCD number'), card('SECTOR', int(self.sector), 'Observing sector'),
			card('PROCVER', 'photometry_amd-0.1', 'Version of photometry pipeline'), card('FILEVER', '1.5', 'File format version'),
			card('DATA_REL', data_rel, 'Data release number'), card('VERSION', int(version), 'Version of the processing'),
			card('PHOTMET', self.method, 'Photometric method used'),
			card('RADESYS', 'ICRS', 'reference frame of celestial coordinates'), card('EQUINOX', 2000.0, 'equinox of celestial coordinate system'),
			card('RA_OBJ', opt('ra_J2000'), '[deg] Right ascension'), card('DEC_OBJ', opt('decl_J2000'), '[deg] Declination'),
			card('PMRA', opt('pm_ra'), '[mas/yr] RA proper motion'), card('PMDEC', opt('pm_decl'), '[mas/yr] Dec proper motion'),
			card('TESSMAG', float(tgt['tmag']), '[mag] TESS magnitude'), card('TEFF', opt('teff'), '[K] Effective temperature'),
		]
		for key, value in self.additional_headers.items(): # K2P2 settings, AP_CONT, ... (:1497-1499)
			if isinstance(value, tuple):
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
			column('QUALITY', 'J', np.zeros(n, dtype='int32'), None, 'B16.16', 'column title: photometry quality flags'),
			column('PIXEL_QUALITY', 'J', col['quality'], None, 'B16.16', 'column title: pixel quality flags'),
			column('MOM_CENTR1', 'D', col['pos_centroid'][:, 0], 'pixels', 'F10.5', 'column title: moment-derived column centroid', 'column units: pixels'),
			column('MOM_CENTR2', 'D', col['pos_centroid'][:, 1], 'pixels', 'F10.5', 'column title: moment-derived row centroid', 'column units: pixels'),
			column('POS_CORR1', 'D', col['pos_corr'][:, 0], 'pixels', 'F14.7', 'column title: column position correction', 'column units: pixels'),
			column('POS_CORR2', 'D', col['pos_corr'][:, 1], 'pixels', 'F14.7', 'column title: row position correction', 'column units: pixels'),
		]
		# time keywords (:1602-1641); without cosmic-ray mitigation information deadc = int_time / frametime
		tdel = cadence / 86400
		tstart = float(col['time'][0] - tdel/2) if n else float('nan')
		tstop = float(col['time'][-1] + tdel/2) if n else float('nan')
		telapse = tstop - tstart
		frametime, int_time, readtime = 2.0, 1.98, 0.02
		deadc = int_time / frametime
		num_frm = int(round(cadence / frametime))
		tcards = [
			card('INHERIT', True, 'inherit the primary header'),
			card('TIMEREF', 'SOLARSYSTEM', 'barycentric correction applied to times'),
			card('TIMESYS', 'TDB', 'time system is Barycentric Dynamical Time (TDB)'),
			card('BJDREFI', 2457000, 'integer part of BTJD reference date'), card('BJDREFF', 0.0, 'fraction of the day in BTJD reference date'),
			card('TIMEUNIT', 'd', 'time unit for TIME, TSTART and TSTOP'),
			card('TSTART', tstart, 'observation start time in BTJD'), card('TSTOP', tstop, 'observation stop time in BTJD'),
			card('MJD-BEG', tstart + 2457000 - 2400000.5, 'observation start time in MJD'),
			card('MJD-END', tstop + 2457000 - 2400000.5, 'observation start time in MJD'),
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
		icards = [card('INHERIT', True, 'inherit the primary header'),
			card('STAMP_R1', int(self._stamp[0]), 'first CCD row of the stamp'), card('STAMP_R2', int(self._stamp[1]), 'last CCD row + 1'),
			card('STAMP_C1', int(self._stamp[2]), 'first CCD column of the stamp'), card('STAMP_C2', int(self._stamp[3]), 'last CCD column + 1')]
		fname = (f'tess{self.starid:011d}-s{int(self.sector):03d}-{int(self.camera):d}-{int(self.ccd):d}-c{cadence:04d}'
			f'-dr{data_rel:02d}-v{int(version):02d}-tasoc_lc.fits.gz')
		path = os.path.join(output_folder, fname)
		fitsio.write(path, [fitsio.primary_hdu(prim), fitsio.bintable_hdu('LIGHTCURVE', columns, tcards),
			fitsio.image_hdu('SUMIMAGE', SumImage, icards), fitsio.image_hdu('APERTURE', mask, icards)])
		self._details['filepath_lightcurve'] = os.path.relpath(path, os.path.abspath(self.output_folder_base)).replace('\\', '/')
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


class AperturePhotometry(BasePhotometry):
	"""
	Simple aperture photometry with K2P2 masks (AperturePhotometry/photometry.py:17-257): the whole
	``do_photometry`` body -- sum image, mask creation, selection, extraction, contamination -- runs on
	the device; the stamp-resize retry loop (:75-170) stays here because it needs new cut-outs.
	"""

	def do_photometry(self):
		logger = logging.getLogger(__name__)
		logger.info("Running aperture photometry...")
		k2p2_settings = {'thresh': 0.8, 'min_no_pixels_in_mask': 4, 'min_for_cluster': 4,
			'cluster_radius': np.sqrt(2) + np.finfo(np.float64).eps, 'segmentation': True, 'ws_blur': 0.5,
			'ws_thres': 0, 'ws_footprint': 3, 'extend_overflow': True}
		ExpectedFlux = mag2flux(self.target['tmag'])
		haloswitch_tmag_limit = self.settings.getfloat('haloswitch', 'tmag_limit')
		haloswitch_flux_limit = self.settings.getfloat('haloswitch', 'flux_limit')
		allow_retries = 10 if self.target['tmag'] < 6 else 5
		ctx = self.ctx

		resize_args = {}
		res = None
		for retries in range(allow_retries):
			scene = _OneTargetScene(self)
			res = pipeline.run_aperture(ctx, scene, cubes='host', diagnostics=False) # photometry() below computes them like the reference
			self._sumimage = res['sumimage'][0]
			# bit 0 of the aperture image needs the sum image of THIS stamp (BasePhotometry.py:1043)
			flags = int(res['flags'][0])
			status = int(res['status'][0])
			err = flags >> 8
			if flags & 32:
				logger.error('No flux above threshold.')
			if flags & 1:
				logger.warning("No masks found. Using minimum aperture." if (flags & (32 | 64)) else
					'No mask found for main target. Using minimum aperture.')
			if err == 5:
				logger.error('Too many masks.')
				return STATUS.ERROR
			if err in (1, 2, 3, 4):
				raise RuntimeError({1: "K2P2NoFlux: No measured flux in sum-image", 2: "Selected KDE bandwidth is 0. Cannot estimate density.",
					3: "attempt to get argmin of an empty sequence", 4: "index out of bounds for the target pixel"}[err])
			mask_main = res['mask'][0].astype(bool)

			resize_args = {}
			if flags & 2:
				resize_args['down'] = 10
			if flags & 4:
				resize_args['up'] = 10
			if flags & 8:
				resize_args['left'] = 10
			if flags & 16:
				resize_args['right'] = 10
			if resize_args:
				logger.info("Touching the edges! Retrying.")
				stamp_before = self._stamp
				sumimage_before = self._sumimage
				if not self.resize_stamp(**resize_args):
					resize_args = {}
					self._sumimage = sumimage_before
					logger.warning("Could not resize stamp any further.")
					break
				if self.target['tmag'] <= haloswitch_tmag_limit and not self.datasource.startswith('tpf:'):
					edge = np.zeros_like(mask_main, dtype='bool')
					if resize_args.get('down') and self._stamp[0] == stamp_before[0]:
						edge[0, :] = True
					if resize_args.get('up') and self._stamp[1] == stamp_before[1]:
						edge[-1, :] = True
					if resize_args.get('left') and self._stamp[2] == stamp_before[2]:
						edge[:, 0] = True
					if resize_args.get('right') and self._stamp[3] == stamp_before[3]:
						edge[:, -1] = True
					if np.any(edge):
						EdgeFlux = np.nansum(sumimage_before[mask_main & edge])
						if EdgeFlux/ExpectedFlux > haloswitch_flux_limit:
							logger.error('Stamp resize hit limit. Haloswitch quick break.')
							self._details['edge_flux'] = EdgeFlux
							return STATUS.ERROR
			else:
				break

		if resize_args:
			logger.error('Too many stamp resizes.')
			return STATUS.ERROR

		lc = self.lightcurve
		lc['flux'] = res['flux'][0]
		lc['flux_err'] = res['flux_err'][0]
		lc['flux_background'] = res['flux_background'][0]
		lc['pos_centroid'] = res['pos_centroid'][0]
		self.final_phot_mask = mask_main
		self.final_position_mask = mask_main

		self.additional_headers['KP_THRES'] = (k2p2_settings['thresh'], 'K2P2 sum-image threshold')
		self.additional_headers['KP_MIPIX'] = (k2p2_settings['min_no_pixels_in_mask'], 'K2P2 min pixels in mask')
		self.additional_headers['KP_MICLS'] = (k2p2_settings['min_for_cluster'], 'K2P2 min pix. for cluster')
		self.additional_headers['KP_CLSRA'] = (k2p2_settings['cluster_radius'], 'K2P2 cluster radius')
		self.additional_headers['KP_WS'] = (bool(k2p2_settings['segmentation']), 'K2P2 watershed segmentation')
		self.additional_headers['KP_WSBLR'] = (k2p2_settings['ws_blur'], 'K2P2 watershed blur')
		self.additional_headers['KP_WSTHR'] = (k2p2_settings['ws_thres'], 'K2P2 watershed threshold')
		self.additional_headers['KP_WSFOT'] = (k2p2_settings['ws_footprint'], 'K2P2 watershed footprint')
		self.additional_headers['KP_EX'] = (bool(k2p2_settings['extend_overflow']), 'K2P2 extend overflow')

		my_status = STATUS.OK
		contamination = float(res['contamination'][0])
		if err == 6:
			logger.error("No targets in mask.")
			contamination = np.nan
			my_status = STATUS.ERROR
		logger.info("Contamination: %f", contamination)
		if not np.isnan(contamination):
			self.additional_headers['AP_CONT'] = (contamination, 'AP contamination')
		in_mask = res['cat_in_mask'][:len(self.catalog)].astype(bool)
		skip_targets = [int(s) for s in self.catalog['starid'][in_mask] if s != self.starid]
		if skip_targets:
			logger.info("These stars could be skipped: %s", skip_targets)
			self.report_details(skip_targets=skip_targets)
		if flags & 1:
			my_status = STATUS.WARNING
		return my_status


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


class PSFPhotometry(BasePhotometry):
	"""Non-linear PSF photometry (psf_photometry.py) -- not part of this engine."""
	def do_photometry(self):
		raise NotImplementedError("PSFPhotometry is outside the hot path implemented by photometry_amd")


class HaloPhotometry(BasePhotometry):
	"""Halo photometry (halo/halo_photometry.py, third-party halophot) -- not part of this engine."""
	def do_photometry(self):
		raise NotImplementedError("HaloPhotometry is outside the hot path implemented by photometry_amd")
