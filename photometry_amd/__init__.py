# -*- coding: utf-8 -*-
"""
photometry_amd -- MI355X-native per-target photometry engine (tasoc/photometry hot path).

Host side (Python, mirroring the reference's plugin API) over a C-ABI shared library of
hand-written HIP kernels for gfx950 (``photometry_amd/csrc`` -> ``libtessphot_hip.so``).
There is no CPU fallback: every compute entry point raises if the library is missing.
"""
from .status import STATUS  # noqa: F401

__version__ = '0.1.0'
from .plugins import BasePhotometry, AperturePhotometry, LinPSFPhotometry, PSFPhotometry, HaloPhotometry  # noqa: F401,E402
from .tessphot import tessphot, tessphot_batch, tessphot_frames, tessphot_frames_pipelined  # noqa: F401,E402
