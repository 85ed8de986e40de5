# -*- coding: utf-8 -*-
"""
ORACLE (test infrastructure only) -- the stamp cutter.

Follows ``BasePhotometry._load_cube``, FFI branch (photometry/BasePhotometry.py:720-742): the cube of a target is
``hdf[group/%04d][ir1:ir2, ic1:ic2]`` for every cadence, stacked time-last, with ``ir = stamp_row - pixel_offset_row``
and ``ic = stamp_col - pixel_offset_col``.  Pinned by ``tests/golden/golden_cutout.npz`` (the reference's own
``_load_cube`` executed on a small frame stack).
"""

import numpy as np


def load_cube(frames, stamp, pixel_offset_row=0, pixel_offset_col=0):
	"""
	``frames``: float32 ``(T, R, C)``; ``stamp`` = (row_min, row_max, col_min, col_max) in CCD coordinates.
	Returns float32 ``(rows, cols, T)``.  Pixels outside the frame are NaN (the reference clips its stamps to the
	frame, BasePhotometry.py:643-679, so it never asks for them; numpy slicing would silently shrink the cube).
	"""
	frames = np.asarray(frames, dtype='float32')
	T, R, C = frames.shape
	ir1, ir2 = stamp[0] - pixel_offset_row, stamp[1] - pixel_offset_row
	ic1, ic2 = stamp[2] - pixel_offset_col, stamp[3] - pixel_offset_col
	cube = np.full((ir2 - ir1, ic2 - ic1, T), np.nan, dtype='float32')
	r1, r2, c1, c2 = max(ir1, 0), min(ir2, R), max(ic1, 0), min(ic2, C)
	if r2 > r1 and c2 > c1:
		for k in range(T): # BasePhotometry.py:733-734
			cube[r1 - ir1:r2 - ir1, c1 - ic1:c2 - ic1, k] = frames[k][r1:r2, c1:c2]
	return cube
