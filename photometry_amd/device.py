# -*- coding: utf-8 -*-
"""
Thin object layer over the C ABI: :class:`Context` (one ``tp_ctx`` = one GPU + one HIP
stream) and :class:`DeviceArray` (a typed HBM buffer owned by the caller).
No torch, no numpy-on-GPU: numpy is only used for host staging.
"""

import ctypes
import numpy as np
from . import _lib
from ._lib import TessphotError, tp_cube_desc


def round_up(n, m):
	return ((int(n) + m - 1) // m) * m


class DeviceArray(object):
	"""
	A typed buffer in HBM.  ``shape`` / ``dtype`` describe the logical contents.

	Lifetime: when the object goes, its block returns to the context's allocation cache (``tp_free``) and the NEXT allocation of its
	size class (powers of two from 4 KiB, eighth-steps above 64 KiB) may get the same block -- ordered on the context's stream, so
	work already queued is safe, but a ``ptr`` handed to a later call is not: keep the array named for as long as a raw pointer to
	it is in use (``d = ctx.array(x); lib.tp_...(d.ptr)``, never ``lib.tp_...(ctx.array(x).ptr)`` followed by another allocation).
	"""

	def __init__(self, ctx, shape, dtype, zero=False):
		self.ctx = ctx
		self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
		self.dtype = np.dtype(dtype)
		self.nbytes = int(np.prod(self.shape, dtype='int64')) * self.dtype.itemsize
		p = ctypes.c_void_p()
		ctx._check(ctx.lib.tp_malloc(ctx.handle, max(self.nbytes, 16), ctypes.byref(p)))
		self.ptr = p.value
		if zero:
			self.fill_bytes(0)

	@classmethod
	def from_host(cls, ctx, array, dtype=None):
		a = np.ascontiguousarray(array, dtype=dtype)
		d = cls(ctx, a.shape, a.dtype)
		if a.nbytes:
			ctx._check(ctx.lib.tp_memcpy_h2d(ctx.handle, d.ptr, a.ctypes.data, a.nbytes))
		return d

	def to_host(self, out=None):
		"""The array on the host; ``out``: a C-contiguous numpy array of this shape and dtype to copy into (no temporary)."""
		if out is None:
			out = np.empty(self.shape, dtype=self.dtype)
		else:
			assert out.shape == tuple(self.shape) and out.dtype == np.dtype(self.dtype) and out.flags['C_CONTIGUOUS'], (out.shape, out.dtype, self.shape, self.dtype)
		if self.nbytes:
			self.ctx._check(self.ctx.lib.tp_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, self.nbytes))
		return out

	def fill_bytes(self, value):
		if self.nbytes:
			self.ctx._check(self.ctx.lib.tp_memset(self.ctx.handle, self.ptr, int(value), self.nbytes))

	def free(self):
		if getattr(self, '_view', False):
			self.ptr = None
			return
		if self.ptr is not None and self.ctx is not None and self.ctx.handle is not None:
			self.ctx.lib.tp_free(self.ctx.handle, self.ptr)
		self.ptr = None

	def __del__(self):
		try:
			self.free()
		except Exception: # noqa: B902
			pass

	def slice0(self, start, count):
		"""Non-owning view of ``count`` entries along the first axis starting at ``start`` (same memory)."""
		v = DeviceArray.__new__(DeviceArray)
		v.ctx = self.ctx
		v.dtype = self.dtype
		v.shape = (int(count),) + self.shape[1:]
		row = int(np.prod(self.shape[1:], dtype='int64')) * self.dtype.itemsize
		v.nbytes = int(count) * row
		v.ptr = self.ptr + int(start) * row
		v._view = True
		v._base = self
		return v


class PinnedArray(object):
	"""
	Page-locked host memory (``tp_host_alloc``) seen as a numpy array: the source / destination of the asynchronous copies
	(:meth:`DeviceCube.upload_async`, :meth:`Context.download_async`) that overlap transfers with kernels.
	"""

	def __init__(self, ctx, shape, dtype):
		self.ctx = ctx
		self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
		self.dtype = np.dtype(dtype)
		self.nbytes = int(np.prod(self.shape, dtype='int64')) * self.dtype.itemsize
		p = ctypes.c_void_p()
		ctx._check(ctx.lib.tp_host_alloc(ctx.handle, max(self.nbytes, 16), ctypes.byref(p)))
		self.ptr = p.value
		buf = (ctypes.c_char * max(self.nbytes, 1)).from_address(self.ptr)
		self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape, dtype='int64'))).reshape(self.shape)

	def free(self):
		if self.ptr is not None and self.ctx.handle is not None:
			self.array = None
			self.ctx.lib.tp_host_free(self.ctx.handle, self.ptr)
		self.ptr = None

	def __del__(self):
		try:
			self.free()   # page-locked memory is a scarce resource: never left to the end of the process
		except Exception: # noqa: B902
			pass


def device_view(ctx, ptr, shape, dtype, base=None):
	"""A non-owning :class:`DeviceArray` over ``ptr`` (a piece of a larger allocation kept alive by ``base``)."""
	v = DeviceArray.__new__(DeviceArray)
	v.ctx = ctx
	v.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
	v.dtype = np.dtype(dtype)
	v.nbytes = int(np.prod(v.shape, dtype='int64')) * v.dtype.itemsize
	v.ptr = int(ptr)
	v._view = True
	v._base = base
	return v


class DeviceCube(object):
	"""
	A float32 stamp cube ``[n_targets][H][W][t_pitch]`` in HBM (time fastest, the reference's
	``(rows, cols, times)`` cube per target, BasePhotometry.py:732).  ``t_pitch`` is rounded up
	to a multiple of 32 so that every pixel's series starts on a 128-byte line: the 512 B / 1 KiB pieces a
	wavefront reads from a series then never straddle an extra line (measured: 8 % less HBM traffic in the
	extraction phase than with a 16-byte aligned pitch).
	"""

	def __init__(self, ctx, n_targets, n_cad, height, width, t_pitch=None):
		self.ctx = ctx
		self.n_targets, self.n_cad, self.height, self.width = int(n_targets), int(n_cad), int(height), int(width)
		self.t_pitch = round_up(n_cad, 32) if t_pitch is None else int(t_pitch)
		self.data = DeviceArray(ctx, (self.n_targets, self.height, self.width, self.t_pitch), 'float32')

	@property
	def ptr(self):
		return self.data.ptr

	@property
	def desc(self):
		return tp_cube_desc(self.n_targets, self.n_cad, self.height, self.width, self.t_pitch)

	@classmethod
	def from_host(cls, ctx, cube):
		"""``cube``: float32 ``(Nt, H, W, T)`` (or ``(H, W, T)`` for a single target)."""
		cube = np.asarray(cube)
		if cube.ndim == 3:
			cube = cube[None]
		cube = np.ascontiguousarray(cube, dtype='float32')
		Nt, H, W, T = cube.shape
		d = cls(ctx, Nt, T, H, W)
		if d.t_pitch != T:
			d.data.fill_bytes(0)
		ctx._check(ctx.lib.tp_upload_cube(ctx.handle, d.ptr, d.t_pitch, cube.ctypes.data, T, Nt*H*W, T))
		return d

	def upload_async(self, ctx, pinned, first_target=0, n_targets=None):
		"""
		Enqueue (on ``ctx``'s stream, without waiting) the upload of the targets ``[first_target, first_target + n_targets)``
		of ``pinned`` (:class:`PinnedArray` ``(Nt_host, H, W, T)`` float32) into this cube's targets ``[0, n_targets)``.
		"""
		n = self.n_targets if n_targets is None else int(n_targets)
		T = pinned.shape[3]
		assert pinned.shape[1:3] == (self.height, self.width) and T == self.n_cad and n <= self.n_targets
		rows = n * self.height * self.width
		src = pinned.ptr + int(first_target) * self.height * self.width * T * 4
		ctx._check(ctx.lib.tp_upload_cube_async(ctx.handle, self.ptr, self.t_pitch, src, T, rows, T))

	def to_host(self):
		full = self.data.to_host()
		return np.ascontiguousarray(full[..., :self.n_cad])

	def free(self):
		self.data.free()

	def slice0(self, start, count):
		"""Non-owning view of the targets ``[start, start+count)``."""
		v = DeviceCube.__new__(DeviceCube)
		v.ctx = self.ctx
		v.n_targets, v.n_cad, v.height, v.width, v.t_pitch = int(count), self.n_cad, self.height, self.width, self.t_pitch
		v.data = self.data.slice0(start, count)
		return v


def bind_host_to_device(device=0):
	"""
	Restrict this process (and the threads it starts from now on: the worker threads of the batched frames engine inherit it) to the
	CPUs of the NUMA node the GPU hangs on -- what ``mpiexec --bind-to`` / ``numactl`` do for the reference's MPI workers.  On a
	two-socket host a process whose threads float over both sockets runs the host-driven entries erratically (the batched frames
	entry: 13 ms or 20 ms per call from one start of the process to the next; bound to either node: 13 ms every time).  Returns the
	node, or None when it is unknown or the binding is not possible (the affinity is then left alone).
	"""
	import os
	lib = _lib.load()
	node = ctypes.c_int(-1)
	if lib.tp_device_numa_node(int(device), ctypes.byref(node)) != 0 or node.value < 0:
		return None
	try:
		cpus = set()
		for part in open(f'/sys/devices/system/node/node{node.value}/cpulist').read().strip().split(','):
			a, _, b = part.partition('-')
			cpus.update(range(int(a), int(b or a) + 1))
		cpus &= os.sched_getaffinity(0)
		if not cpus:
			return None
		os.sched_setaffinity(0, cpus)
	except (OSError, ValueError, AttributeError):
		return None
	return node.value


class Context(object):
	"""One GPU, one stream.  Not thread-safe (one Context per host thread)."""

	def __init__(self, device=0, high_priority=None):
		self.lib = _lib.load()
		h = ctypes.c_void_p()
		if high_priority is None:
			rc = self.lib.tp_ctx_create(int(device), ctypes.byref(h))
		else:
			rc = self.lib.tp_ctx_create_stream(int(device), 1 if high_priority else 0, ctypes.byref(h))
		if rc != 0:
			raise TessphotError(rc, (self.lib.tp_last_error(None) or b'').decode())
		self.handle = h.value
		self.device = int(device)

	def _check(self, rc):
		if rc != 0:
			raise TessphotError(rc, (self.lib.tp_last_error(self.handle) or b'').decode())

	def close(self):
		if getattr(self, 'handle', None) is not None:
			eng = self.__dict__.pop('_frames_engine', None)   # pipeline.FramesEngine: its streams and page-locked pool go first
			if eng is not None:
				eng.close()
			self.pinned_trim()
			for c in self.__dict__.get('_side', []):
				c.close()
			self.__dict__['_side'] = []
			self.lib.tp_ctx_destroy(self.handle)
			self.handle = None

	def __enter__(self):
		return self

	def __exit__(self, *args):
		self.close()

	def __del__(self):
		# DeviceArrays may outlive us at interpreter shutdown; the driver reclaims them.
		pass

	def sync(self):
		self._check(self.lib.tp_sync(self.handle))

	def info(self):
		name = ctypes.create_string_buffer(256)
		ncu = ctypes.c_int32()
		hbm = ctypes.c_uint64()
		self._check(self.lib.tp_device_info(self.handle, name, 256, ctypes.byref(ncu), ctypes.byref(hbm)))
		return {'name': name.value.decode(), 'n_cu': ncu.value, 'hbm_bytes': hbm.value}

	# -- allocation helpers ------------------------------------------------------------------
	def empty(self, shape, dtype):
		return DeviceArray(self, shape, dtype)

	def zeros(self, shape, dtype):
		return DeviceArray(self, shape, dtype, zero=True)

	def array(self, host, dtype=None):
		return DeviceArray.from_host(self, host, dtype=dtype)

	def cube(self, host):
		return DeviceCube.from_host(self, host)

	def pinned(self, shape, dtype):
		return PinnedArray(self, shape, dtype)

	# Page-locked buffers for results that are handed to the caller where the DMA left them (no staging copy): locking pages costs
	# milliseconds per 100 MB, so the buffers a result has released are kept for the next one (size classes of 1 MiB and
	# powers of two above; `pinned_trim` gives them back).
	def pinned_block(self, nbytes):
		"""A :class:`PinnedArray` of at least ``nbytes`` bytes (uint8) from the context's pool."""
		cap = 1 << 20
		while cap < nbytes:
			cap <<= 1
		pool = self.__dict__.setdefault('_pinned_pool', {})
		free = pool.get(cap)
		if free:
			return free.pop()
		return PinnedArray(self, (cap,), 'uint8')

	def pinned_release(self, block):
		if block is None or block.ptr is None or self.handle is None:
			return
		self.__dict__.setdefault('_pinned_pool', {}).setdefault(block.nbytes, []).append(block)

	def pinned_trim(self):
		for free in self.__dict__.get('_pinned_pool', {}).values():
			while free:
				free.pop().free()

	def side_contexts(self, n):
		"""``n`` further contexts (= HIP streams) on this device, created once: independent passes of a batched entry (the groups
		of stamp sizes of a round) run on them side by side."""
		side = self.__dict__.setdefault('_side', [])
		while len(side) < n:
			side.append(Context(self.device, high_priority=False))
		return side[:n]

	def download_async(self, pinned, device_array, nbytes=None, host_offset=0):
		"""Enqueue a device -> pinned-host copy on this context's stream (returns at once)."""
		n = device_array.nbytes if nbytes is None else int(nbytes)
		self._check(self.lib.tp_memcpy_d2h_async(self.handle, pinned.ptr + int(host_offset), device_array.ptr, n))

	# -- timing ------------------------------------------------------------------------------
	# -- cross-stream events -------------------------------------------------------------------
	def event(self):
		e = ctypes.c_void_p()
		self._check(self.lib.tp_event_create(self.handle, ctypes.byref(e)))
		return e.value

	def record(self, event):
		self._check(self.lib.tp_event_record(self.handle, event))

	def wait_event(self, event):
		self._check(self.lib.tp_stream_wait_event(self.handle, event))

	def event_sync(self, event):
		"""Block the host until ``event`` has happened."""
		self._check(self.lib.tp_event_sync(self.handle, event))

	def timer_start(self, slot=0):
		self._check(self.lib.tp_timer_start(self.handle, slot))

	def timer_stop(self, slot=0):
		self._check(self.lib.tp_timer_stop(self.handle, slot))

	def timer_ms(self, slot=0):
		ms = ctypes.c_float()
		self._check(self.lib.tp_timer_elapsed_ms(self.handle, slot, ctypes.byref(ms)))
		return float(ms.value)

	def profile(self, on=True):
		self._check(self.lib.tp_profile_enable(self.handle, 1 if on else 0))

	def profile_reset(self):
		self._check(self.lib.tp_profile_reset(self.handle))

	def profile_report(self):
		"""dict kernel name -> (launches, total_ms)."""
		out = {}
		for k in range(self.lib.tp_kernel_count()):
			n = ctypes.c_int64()
			ms = ctypes.c_double()
			self._check(self.lib.tp_profile_get(self.handle, k, ctypes.byref(n), ctypes.byref(ms)))
			if n.value:
				out[self.lib.tp_kernel_name(k).decode()] = (int(n.value), float(ms.value))
		return out
