# -*- coding: utf-8 -*-
"""
ctypes binding of ``libtessphot_hip.so`` (declared in ``include/tessphot_hip.h``).

There is deliberately NO fallback: if the shared library is missing or cannot be loaded
:func:`load` raises :class:`TessphotLibraryError`, and every compute entry point of the
package goes through :func:`load`.
"""

import os
import ctypes
from ctypes import (c_int, c_int32, c_int64, c_uint32, c_uint64, c_uint8, c_float, c_double, c_char_p, c_void_p, POINTER,
	Structure, byref)

LIB_NAME = 'libtessphot_hip.so'
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


class TessphotLibraryError(RuntimeError):
	"""The HIP library is missing / not loadable: the engine cannot run."""


class TessphotError(RuntimeError):
	"""A library call returned an error code."""
	def __init__(self, code, message):
		super().__init__(f"libtessphot_hip error {code}: {message}")
		self.code = code


class tp_cube_desc(Structure):
	_fields_ = [('n_targets', c_int32), ('n_cad', c_int32), ('height', c_int32), ('width', c_int32), ('t_pitch', c_int64)]


class tp_zoom_image(Structure):
	_fields_ = [('d_coef', c_void_p), ('d_vmin', c_void_p), ('d_vmax', c_void_p),
		('mesh_rows', c_int32), ('mesh_cols', c_int32), ('box_size', c_int32), ('frame_cols', c_int32)]


class tp_radial_image(Structure):
	_fields_ = [('col_offset', c_double), ('xcen', c_double), ('ycen', c_double), ('d_knots', c_void_p), ('d_coefs', c_void_p), ('d_n_knots', c_void_p),
		('d_zeropoint', c_void_p), ('max_knots', c_int32), ('reserved', c_int32)]


class tp_frames_stack(Structure):
	_fields_ = [('d_images', c_void_p), ('d_images_err', c_void_p), ('d_backgrounds', c_void_p),
		('n_frames', c_int32), ('n_rows', c_int32), ('n_cols', c_int32), ('row0', c_int32), ('col0', c_int32),
		('d_sumimage', c_void_p), ('d_images_t', c_void_p), ('d_images_err_t', c_void_p), ('d_backgrounds_t', c_void_p), ('t_pitch', c_int64)]


class tp_k2p2_params(Structure):
	_fields_ = [('thresh', c_double), ('min_no_pixels_in_mask', c_int32), ('min_for_cluster', c_int32),
		('extend_overflow', c_int32), ('reserved', c_int32), ('ws_thres', c_double), ('saturation_limit', c_double)]


_p = c_void_p # device / host data pointers are passed as integers
_desc_p = POINTER(tp_cube_desc)

#: name -> (restype, argtypes); mirrors include/tessphot_hip.h exactly.
SIGNATURES = {
	'tp_version': (c_int, []),
	'tp_device_count': (c_int, [POINTER(c_int)]),
	'tp_ctx_create': (c_int, [c_int, POINTER(c_void_p)]),
	'tp_ctx_create_stream': (c_int, [c_int, c_int, POINTER(c_void_p)]),
	'tp_ctx_destroy': (c_int, [c_void_p]),
	'tp_last_error': (c_char_p, [c_void_p]),
	'tp_device_info': (c_int, [c_void_p, c_char_p, c_int, POINTER(c_int32), POINTER(c_uint64)]),
	'tp_device_numa_node': (c_int, [c_int, POINTER(c_int)]),
	'tp_malloc': (c_int, [c_void_p, c_uint64, POINTER(c_void_p)]),
	'tp_free': (c_int, [c_void_p, _p]),
	'tp_cache_trim': (c_int, [c_void_p]),
	'tp_memset': (c_int, [c_void_p, _p, c_int, c_uint64]),
	'tp_memcpy_h2d': (c_int, [c_void_p, _p, _p, c_uint64]),
	'tp_memcpy_d2h': (c_int, [c_void_p, _p, _p, c_uint64]),
	'tp_memcpy_d2d': (c_int, [c_void_p, _p, _p, c_uint64]),
	'tp_upload_cube': (c_int, [c_void_p, _p, c_int64, _p, c_int64, c_int64, c_int64]),
	'tp_host_alloc': (c_int, [c_void_p, c_uint64, POINTER(c_void_p)]),
	'tp_host_free': (c_int, [c_void_p, _p]),
	'tp_upload_cube_async': (c_int, [c_void_p, _p, c_int64, _p, c_int64, c_int64, c_int64]),
	'tp_memcpy_d2h_async': (c_int, [c_void_p, _p, _p, c_uint64]),
	'tp_sync': (c_int, [c_void_p]),
	'tp_event_create': (c_int, [c_void_p, POINTER(c_void_p)]),
	'tp_event_destroy': (c_int, [c_void_p, c_void_p]),
	'tp_event_record': (c_int, [c_void_p, c_void_p]),
	'tp_stream_wait_event': (c_int, [c_void_p, c_void_p]),
	'tp_event_sync': (c_int, [c_void_p, c_void_p]),
	'tp_timer_start': (c_int, [c_void_p, c_int]),
	'tp_timer_stop': (c_int, [c_void_p, c_int]),
	'tp_timer_elapsed_ms': (c_int, [c_void_p, c_int, POINTER(c_float)]),
	'tp_profile_enable': (c_int, [c_void_p, c_int]),
	'tp_profile_reset': (c_int, [c_void_p]),
	'tp_kernel_count': (c_int, []),
	'tp_kernel_name': (c_char_p, [c_int]),
	'tp_profile_get': (c_int, [c_void_p, c_int, POINTER(c_int64), POINTER(c_double)]),
	'tp_sumimage': (c_int, [c_void_p, _desc_p, _p, _p, c_int64, c_uint32, _p, c_int64, _p]),
	'tp_k2p2_masks': (c_int, [c_void_p, c_int32, c_int32, c_int32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
		POINTER(tp_k2p2_params), _p, _p, _p, _p, _p, _p]),
	'tp_aperture_extract': (c_int, [c_void_p, _desc_p, _p, _p, _p, c_int32, c_int64, _p, c_int64, _p, _p, _p,
		_p, _p, _p, _p, _p, c_int64]),
	'tp_aperture_photometry': (c_int, [c_void_p, _desc_p, _p, _p, _p, c_int32, c_int64, _p, c_int64,
		_p, c_int64, c_uint32,
		_p, _p, _p, _p, _p, _p, _p,
		_p, _p, _p, _p,
		_p, _p, POINTER(tp_k2p2_params),
		_p, _p, _p, _p, _p, _p, _p,
		_p, _p, _p, _p, _p, c_int64]),
	'tp_aperture_photometry_from_sumimage': (c_int, [c_void_p, _desc_p, _p, _p, _p, c_int32, c_int64, _p, c_int64,
		_p, c_int64, c_uint32,
		_p, _p, _p, _p, _p, _p, _p,
		_p, _p, _p, _p,
		_p, _p, POINTER(tp_k2p2_params),
		_p, _p, _p, _p, _p, _p, _p,
		_p, _p, _p, _p, _p, c_int64]),
	'tp_background_stamp': (c_int, [c_void_p, _desc_p, _p, c_double, c_double, _p, c_int64]),
	'tp_background_sumimage': (c_int, [c_void_p, _desc_p, _p, c_double, c_double, c_int32, _p, c_int64, c_uint32, _p, _p, c_int64, _p]),
	'tp_smooth_time': (c_int, [c_void_p, c_int32, c_int32, c_int64, c_int32, _p, _p]),
	'tp_subtract_background': (c_int, [c_void_p, _desc_p, _p, _p, _p, c_int64, _p, c_uint32, _p, _p]),
	'tp_background_mesh': (c_int, [c_void_p, _p, c_int32, c_int32, c_int32, c_int64, c_int64, _p, c_int64, _p, c_int64, c_double, c_int32, _p, _p]),
	'tp_background_mesh_finish': (c_int, [c_void_p, _p, _p, c_int32, c_int32, c_int32, c_int32, c_double, c_int32, _p, _p, _p, _p]),
	'tp_background_zoom': (c_int, [c_void_p, _p, _p, _p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int64, c_int64, _p]),
	'tp_frames_smooth_time': (c_int, [c_void_p, c_int32, c_int64, c_int64, c_int32, _p, _p]),
	'tp_frames_subtract': (c_int, [c_void_p, c_int64, _p, _p, _p, _p, c_uint32, _p, _p]),
	'tp_frames_transpose': (c_int, [c_void_p, _p, c_int32, c_int64, c_int64, _p, c_int64]),
	'tp_aperture_extract_stack': (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, _p, _p, _p, c_int64, c_int32, c_int32, c_int32, c_int32,
		_p, _p, _p, _p, _p, _p, _p, _p, c_int64]),
	'tp_frames_sumimage': (c_int, [c_void_p, c_int32, c_int64, c_int64, _p, _p, c_uint32, _p]),
	'tp_radial_zeropoint': (c_int, [c_void_p, _p, c_int32, c_int64, c_int64, _p, c_int64, _p, c_int64, c_double, _p, c_int32, _p]),
	'tp_radial_ring_modes': (c_int, [c_void_p, _p, c_int32, c_int64, c_int64, _p, c_int64, _p, c_int64, c_double, _p, _p, _p, c_int32, c_int32,
		c_double, _p, _p, _p]),
	'tp_radial_profiles': (c_int, [c_void_p, c_int32, c_int32, _p, _p, c_int32, c_int32, _p, _p, _p]),
	'tp_radial_evaluate': (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int64, c_double, c_double, c_double, _p, _p, _p, c_int32, _p, _p, c_int64, _p]),
	'tp_frames_pixel_flags': (c_int, [c_void_p, _p, c_int32, c_int32, c_int32, c_int64, c_int64, _p, c_int32, c_double, c_uint32, c_uint32, _p, _p]),
	'tp_frames_used_in_background': (c_int, [c_void_p, _p, c_int32, c_int64, c_uint32, c_double, _p]),
	'tp_frames_median_filter': (c_int, [c_void_p, _p, c_int32, c_int32, c_int32, c_int64, c_int64, _p, c_int32, _p]),
	'tp_frames_block_median_accumulate': (c_int, [c_void_p, _p, c_int64, c_int64, _p, c_int32, _p]),
	'tp_frames_threshold_flags': (c_int, [c_void_p, _p, _p, c_double, c_uint32, c_int64, c_int32, _p]),
	'tp_linpsf_prf': (c_int, [c_void_p, c_int32, c_int32, c_int32, _p, _p, _p]),
	'tp_linpsf_set_path': (c_int, [c_void_p, c_int32]),
	'tp_linpsf_last_counts': (c_int, [c_void_p, POINTER(ctypes.c_int64), c_int32]),
	'tp_star_positions': (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int64]),
	'tp_linpsf_fit': (c_int, [c_void_p, _desc_p, _p, _p, c_int64, _p, _p, _p, c_int32, c_int32, _p, _p, _p, _p, c_int64, c_double,
		_p, _p, _p, c_int64, _p, _p, _p]),
	'tp_psf_fit': (c_int, [c_void_p, _desc_p, _p, _p, _p, _p, _p, c_int32, _p, _p, _p, c_double, c_double, c_int32, c_int32,
		_p, _p, _p, _p, c_int64, _p, _p, _p]),
	'tp_linpsf_fit_xy': (c_int, [c_void_p, _desc_p, _p, _p, c_int64, _p, _p, _p, c_int32, c_int32, c_int32, _p, _p, _p, _p, c_int64, c_double,
		_p, _p, _p, c_int64, _p, _p, _p]),
	'tp_psf_fit_xy': (c_int, [c_void_p, _desc_p, _p, _p, _p, _p, _p, c_int32, c_int32, _p, _p, _p, c_double, c_double, c_int32, c_int32,
		_p, _p, _p, _p, c_int64, _p, _p, _p]),
	'tp_lightcurve_diagnostics': (c_int, [c_void_p, c_int32, c_int32, _p, _p, _p, _p, c_int64, _p, _p, c_int64, c_uint32,
		_p, _p, _p, c_int32, c_int32, c_double, _p]),
	'tp_cut_stamps': (c_int, [c_void_p, _p, c_int32, c_int32, c_int32, c_int64, c_int64, c_int32, c_int32, _p, _desc_p, _p]),
	'tp_comm_unique_id': (c_int, [c_char_p, c_int]),
	'tp_comm_init': (c_int, [c_void_p, c_char_p, c_int, c_int, c_int]),
	'tp_comm_destroy': (c_int, [c_void_p]),
	'tp_comm_info': (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
	'tp_comm_gather': (c_int, [c_void_p, _p, _p, c_uint64, c_int]),
	'tp_comm_allgather': (c_int, [c_void_p, _p, _p, c_uint64]),
	'tp_block_compact': (c_int, [c_void_p, _p, _p, _p, c_int32]),
	'tp_cut_stamps_multi': (c_int, [c_void_p, c_int32, _p, c_int32, c_int32, c_int32, c_int64, c_int64, c_int32, c_int32, _p, _desc_p, _p]),
	'tp_cut_stamps_masked': (c_int, [c_void_p, c_int32, _p, c_int32, c_int32, c_int32, c_int64, c_int64, c_int32, c_int32, _p, _desc_p, _p, _p]),
	'tp_radial_zeropoint_zoom': (c_int, [c_void_p, _p, c_int32, c_int64, c_int64, POINTER(tp_zoom_image), _p, c_int64, c_double, _p, c_int32, _p]),
	'tp_radial_ring_modes_zoom': (c_int, [c_void_p, _p, c_int32, c_int64, c_int64, POINTER(tp_zoom_image), _p, c_int64, c_double, _p, _p, _p, c_int32, c_int32, c_double, _p, _p, _p]),
	'tp_radial_evaluate_zoom': (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int64, POINTER(tp_radial_image), POINTER(tp_zoom_image), _p]),
	'tp_background_mesh_radial': (c_int, [c_void_p, _p, c_int32, c_int32, c_int32, c_int64, c_int64, _p, c_int64, POINTER(tp_radial_image), c_double, c_int32, _p, _p]),
	'tp_frames_engine_create': (c_int, [c_int, c_int32, POINTER(c_void_p)]),
	'tp_frames_engine_destroy': (c_int, [c_void_p]),
	'tp_frames_engine_info': (c_int, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_uint64)]),
	'tp_frames_catalog_create': (c_int, [c_int64, _p, _p, _p, _p, POINTER(c_void_p)]),
	'tp_frames_catalog_destroy': (c_int, [c_void_p]),
	'tp_crop_sumimage': (c_int, [c_void_p, _p, c_int32, c_int32, c_int64, c_int32, c_int32, _p, c_int32, c_int32, c_int32, _p]),
	'tp_frames_submit': (c_int, [c_void_p, POINTER(tp_frames_stack), c_void_p, c_int32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, c_double, POINTER(c_void_p)]),
	'tp_frames_wait': (c_int, [c_void_p]),
	'tp_frames_poll': (c_int, [c_void_p, POINTER(c_int32)]),
	'tp_frames_counts': (c_int, [c_void_p, POINTER(c_int32), POINTER(c_int64)]),
	'tp_frames_targets': (c_int, [c_void_p, _p, _p, _p, _p, _p, _p]),
	'tp_frames_group': (c_int, [c_void_p, c_int32, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int64), POINTER(c_int64), POINTER(c_void_p), POINTER(c_uint64)]),
	'tp_frames_group_lists': (c_int, [c_void_p, c_int32, _p, _p, _p]),
	'tp_frames_events': (c_int, [c_void_p, _p, _p, _p, _p, _p, _p]),
	'tp_frames_text': (c_char_p, [c_void_p, c_int32]),
	'tp_frames_release': (c_int, [c_void_p]),
	'tp_synth_fill': (c_int, [c_void_p, _desc_p, c_int32, _p, _p, _p, _p, _p, c_double, c_double, c_uint64, _p, _p, _p, _p]),
}

_lib = None


def load():
	"""Load (once) and return the ctypes library; raises TessphotLibraryError if impossible."""
	global _lib
	if _lib is not None:
		return _lib
	if not os.path.exists(LIB_PATH):
		raise TessphotLibraryError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
			"(or `make -C photometry_amd/csrc`).  photometry_amd has no CPU fallback.")
	# Hardware queues: HIP maps the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and streams that
	# share one run in turn.  The batched frames entry keeps several jobs of three streams each in flight, whose latency-bound
	# passes over a few resized stamps are meant to run under the large passes of other jobs: with 4 queues they queue up behind
	# them instead (measured: 2.3e5 -> 3.4e5 targets/s with 16).  Read when the HIP runtime initialises: set before the library loads.
	# This changes the environment of the embedding application (and of its child processes): a value the application set itself
	# is never overwritten, and TESSPHOT_KEEP_HW_QUEUES=1 leaves the variable alone altogether.
	if os.environ.get('TESSPHOT_KEEP_HW_QUEUES', '0') != '1':
		os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
	try:
		lib = ctypes.CDLL(LIB_PATH)
	except OSError as e:
		raise TessphotLibraryError(f"could not load {LIB_PATH}: {e}") from e
	for name, (restype, argtypes) in SIGNATURES.items():
		try:
			fn = getattr(lib, name)
		except AttributeError as e:
			raise TessphotLibraryError(f"{LIB_PATH} does not export {name}") from e
		fn.restype = restype
		fn.argtypes = argtypes
	_lib = lib
	return lib


def exported_symbols():
	"""Names declared in the header table (used by the CPU symbol test)."""
	return sorted(SIGNATURES.keys())


__all__ = ['load', 'TessphotError', 'TessphotLibraryError', 'tp_cube_desc', 'tp_k2p2_params', 'tp_frames_stack', 'tp_zoom_image', 'tp_radial_image', 'SIGNATURES', 'LIB_PATH',
	'byref', 'c_void_p', 'c_int', 'c_int32', 'c_int64', 'c_uint64', 'c_float', 'c_double', 'c_uint8']
