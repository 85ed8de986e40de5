# -*- coding: utf-8 -*-
"""
ORACLE (test infrastructure only) -- quality bitmasks.

Follows ``photometry/quality.py``: ``QualityFlagsBase.filter`` (:39-53),
``TESSQualityFlags.DEFAULT_BITMASK`` (:123-124), ``PixelQualityFlags`` (:157-173).
"""

import numpy as np

# TESSQualityFlags (photometry/quality.py:106-124)
AttitudeTweak = 1
SafeMode = 2
CoarsePoint = 4
EarthPoint = 8
ZeroCrossing = 16
Desat = 32
ApertureCosmic = 64
ManualExclude = 128
SensitivityDropout = 256
ImpulsiveOutlier = 512
CollateralCosmic = 1024
EarthMoonPlanetInFOV = 2048
ScatteredLight = 4096

#: photometry/quality.py:123-124 -> 1|2|4|8|32|64|128|4096 = 4335
TESS_DEFAULT_BITMASK = (AttitudeTweak | SafeMode | CoarsePoint | EarthPoint
	| Desat | ApertureCosmic | ManualExclude | ScatteredLight)
assert TESS_DEFAULT_BITMASK == 4335

# PixelQualityFlags (photometry/quality.py:157-166)
PIXEL_NotUsedForBackground = 1
PIXEL_ManualExclude = 2
PIXEL_BackgroundShenanigans = 4
PIXEL_DEFAULT_BITMASK = PIXEL_ManualExclude


def tess_filter(quality, flags=TESS_DEFAULT_BITMASK):
	"""``True`` where quality does NOT contain any of ``flags`` (quality.py:39-53)."""
	return (np.asarray(quality) & flags) == 0


def pixel_filter(pixel_flags, flags=PIXEL_DEFAULT_BITMASK):
	"""``True`` where the pixel flag does NOT contain any of ``flags`` (quality.py:39-53,166)."""
	return (np.asarray(pixel_flags) & flags) == 0
