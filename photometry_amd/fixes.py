# -*- coding: utf-8 -*-
"""
Timestamp offset of the early TESS data releases (photometry/fixes/time_offset.py:64-180): sectors 1-21 carried start / mid /
end times that were 2.0 s late, minus the 31 / 21 / 11 ms of the focal-plane electronics, and -- in FFIs up to data release 27 --
lacked the staggered read-out of the cameras (0.5 s steps, order 1, 3, 4, 2) and CCDs (20 ms steps).  ``BasePhotometry`` applies
it to ``lightcurve['time']`` (BasePhotometry.py:244, :384) and the prepare stage to the frame times (prepare.py:459-461); a
``StampSource`` that hands over uncorrected times calls it the same way.
"""
import logging

_FIRST_RELEASE_27 = ('spoc-4.0.14-20200108', 'spoc-4.0.15-20200114', 'spoc-4.0.17-20200130')
_FIRST_RELEASE_29 = ('spoc-4.0.17-20200130', 'spoc-4.0.20-20200220', 'spoc-4.0.21-20200227')
_CAMERA_STAGGER = {1: 0.000, 2: 1.500, 3: 0.500, 4: 1.000}
_CCD_STAGGER = {1: 0.000, 2: 0.020, 3: 0.040, 4: 0.060}
_SHIFT = {'mid': 0.021, 'start': 0.031, 'end': 0.011}


def time_offset(time, header, datatype='ffi', timepos='mid', return_flag=False, settings=None):
	"""
	Corrected timestamps (days).  ``header``: mapping with ``DATA_REL`` and, where they matter, ``PROCVER``, ``CAMERA``, ``CCD``,
	``TIME_OFFSET_CORRECTED``.  Raises ``ValueError`` for an invalid ``timepos`` and for data releases 27 / 29 without
	``PROCVER`` (the two deliveries of sectors 20 and 21 cannot be told apart then), ``KeyError`` for a missing card, as the
	reference does.  ``settings``: a ``configparser`` object whose ``[fixes] time_offset = False`` switches the fix off.
	"""
	logger = logging.getLogger(__name__)
	datarel = int(header['DATA_REL'])
	procver = header.get('PROCVER', None)
	already_corrected = bool(header.get('TIME_OFFSET_CORRECTED', False))
	if timepos not in _SHIFT:
		raise ValueError("Invalid TIMEPOS")
	first_27 = False
	if already_corrected or datarel > 29:
		apply_correction = False
	elif datarel <= 26:
		apply_correction = True
	elif datarel in (27, 29) and procver is None:
		raise ValueError("The timestamps of these data may need to be corrected, but the PROCVER header is not present. "
			"HDF5 files may need to be re-created.")
	elif datarel == 27 and procver in _FIRST_RELEASE_27:
		first_27 = True
		apply_correction = True
	elif datarel == 29 and procver in _FIRST_RELEASE_29:
		apply_correction = True
	else:
		apply_correction = False
	if apply_correction and settings is not None and not settings.getboolean('fixes', 'time_offset', fallback=True):
		logger.warning("SettingsWarning: Time offset fix has been turned off in settings.")
		apply_correction = False
	if apply_correction:
		stagger = 0
		if datatype == 'ffi' and (datarel <= 26 or first_27):
			stagger = _CAMERA_STAGGER[int(header['CAMERA'])]
			stagger += _CCD_STAGGER[int(header['CCD'])]
		time = time + (stagger - 2.000 + _SHIFT[timepos]) / 86400
	return (time, apply_correction) if return_flag else time
