# -*- coding: utf-8 -*-
"""
On-disk frame stacks and their streaming into HBM.

The reference keeps the prepared images of one CCD in an HDF5 file -- groups ``images``, ``images_err``, ``backgrounds``,
``pixel_flags`` with one 2-D dataset per cadence (``%04d``), 64 x 64 lzf-compressed, shuffled, checksummed chunks
(photometry/prepare.py:136-141, 251-257, 428-429, 496-502) -- and every target later reads its cut-outs back chunk by chunk,
``3 x T`` reads per target and per stamp resize (``BasePhotometry._load_cube``, BasePhotometry.py:720-751): the reference's
real wall-clock bottleneck.  Neither h5py nor libhdf5 exists in this image, so this module defines the build's own container
for the same content, laid out for what happens to it here -- being streamed ONCE into HBM, where the stamp cutter
(``tp_cut_stamps``) serves every target and every resize:

``<name>.tpstack``: a 4 KiB header (magic ``TPSTACK1``, little-endian JSON: group names, dtype, T, R, C, PIXEL_OFFSET_ROW /
COLUMN, time / cadenceno / quality vectors, free-form attributes), then for every group the raw frames ``[T][R][C]``, each
group starting on a 4 KiB boundary.  Frames are uncompressed: at PCIe / NVMe rates decompression on the host would be the
slower link.  A converter from the reference's HDF5 files is a dozen lines of h5py on a machine that has it
(INTEGRATION.md).

``load_stack`` memory-maps the file and uploads it in chunks of frames through two pinned staging buffers, the copy of chunk
``i + 1`` into pinned memory overlapping the H2D transfer of chunk ``i``.
"""

import json
import os
import numpy as np

MAGIC = b'TPSTACK1'
HEADER_BYTES = 4096
ALIGN = 4096


def _aligned(n):
	return (int(n) + ALIGN - 1) // ALIGN * ALIGN


def write_stack(path, groups, row_offset=0, col_offset=44, time=None, cadenceno=None, quality=None, attrs=None):
	"""
	Write ``groups`` (dict name -> array ``(T, R, C)``, all the same shape; float32 images or uint8 flags) to ``path``.
	``row_offset / col_offset``: CCD coordinates of frame pixel (0, 0) (PIXEL_OFFSET_ROW / _COLUMN, BasePhotometry.py:724-727).
	"""
	names = list(groups)
	first = np.asarray(groups[names[0]])
	T, R, C = first.shape
	meta = {'shape': [int(T), int(R), int(C)], 'row_offset': int(row_offset), 'col_offset': int(col_offset), 'groups': [], 'attrs': attrs or {}}
	for key, vec in (('time', time), ('cadenceno', cadenceno), ('quality', quality)):
		if vec is not None:
			meta[key] = [float(v) if key == 'time' else int(v) for v in np.asarray(vec)]
	offset = HEADER_BYTES
	arrays = []
	for name in names:
		a = np.ascontiguousarray(groups[name])
		if a.shape != (T, R, C):
			raise ValueError(f"group {name}: shape {a.shape} != {(T, R, C)}")
		if a.dtype not in (np.dtype('float32'), np.dtype('uint8')):
			raise ValueError(f"group {name}: float32 or uint8 expected")
		meta['groups'].append({'name': name, 'dtype': a.dtype.str, 'offset': offset, 'nbytes': int(a.nbytes)})
		arrays.append((offset, a))
		offset = _aligned(offset + a.nbytes)
	head = json.dumps(meta).encode('utf-8')
	if len(head) + len(MAGIC) + 8 > HEADER_BYTES:
		# long time vectors: the JSON goes behind the data, the header only says where
		meta_tail = head
		head = json.dumps({'meta_offset': offset, 'meta_nbytes': len(meta_tail)}).encode('utf-8')
	else:
		meta_tail = None
	with open(path, 'wb') as fh:
		fh.write(MAGIC + len(head).to_bytes(8, 'little') + head)
		for off, a in arrays:
			fh.seek(off)
			a.tofile(fh)
		if meta_tail is not None:
			fh.seek(offset)
			fh.write(meta_tail)
		else:
			fh.truncate(max(offset, fh.tell()))
	return path


def read_header(path):
	with open(path, 'rb') as fh:
		if fh.read(len(MAGIC)) != MAGIC:
			raise ValueError(f"{path}: not a TPSTACK1 file")
		n = int.from_bytes(fh.read(8), 'little')
		meta = json.loads(fh.read(n).decode('utf-8'))
		if 'meta_offset' in meta:
			fh.seek(meta['meta_offset'])
			meta = json.loads(fh.read(meta['meta_nbytes']).decode('utf-8'))
	return meta


def open_group(path, name, meta=None):
	"""Memory map of one group ``(T, R, C)`` (no copy)."""
	meta = read_header(path) if meta is None else meta
	g = next((x for x in meta['groups'] if x['name'] == name), None)
	if g is None:
		raise KeyError(name)
	return np.memmap(path, mode='r', dtype=np.dtype(g['dtype']), offset=g['offset'], shape=tuple(meta['shape']))


def upload_group(ctx, mm, frames_per_chunk=None, up=None):
	"""
	Stream a memory-mapped group into a new DeviceArray ``(T, R, C)``: two pinned staging buffers, the host copy of the next
	chunk overlapping the transfer of the current one (``tp_upload_cube_async`` on a second stream).
	"""
	from .device import Context
	T, R, C = mm.shape
	item = mm.dtype.itemsize
	if frames_per_chunk is None:
		frames_per_chunk = max(1, min(T, (256 << 20) // max(R * C * item, 1)))   # about 256 MiB per chunk
	if (R * C * item) % 4:
		raise ValueError('frame size must be a multiple of 4 bytes')   # before anything is allocated
	own = up is None
	dst = ctx.empty((T, R, C), mm.dtype)
	stage, done = [], []
	try:
		if own:
			up = Context(ctx.device, high_priority=False)
		stage = [ctx.pinned((frames_per_chunk, R, C), mm.dtype) for _ in range(2)]
		done = [up.event(), up.event()]
		for i, k0 in enumerate(range(0, T, frames_per_chunk)):
			s = i % 2
			n = min(frames_per_chunk, T - k0)
			if i >= 2:
				up.event_sync(done[s])                                 # the transfer that last read this staging buffer is over
			stage[s].array[:n] = mm[k0:k0 + n]                         # page cache / disk -> pinned memory, while chunk i-1 is in flight
			nbytes = n * R * C * item
			up._check(up.lib.tp_upload_cube_async(up.handle, dst.ptr + k0 * R * C * item, nbytes // 4, stage[s].ptr, nbytes // 4, 1, nbytes // 4))
			up.record(done[s])
		up.sync()
	except BaseException:
		dst.free()
		raise
	finally:
		# staging buffers, events and the transfer context go whatever happened
		if up is not None and getattr(up, 'handle', None) is not None:
			try:
				up.sync()
			except Exception: # noqa: B902
				pass
			for e in done:
				up.lib.tp_event_destroy(up.handle, e)
		for p in stage:
			p.free()
		if own and up is not None:
			up.close()
	return dst


def load_stack(ctx, path, groups=('images', 'images_err', 'backgrounds')):
	"""
	Load ``groups`` of a ``.tpstack`` file into HBM.  Returns ``(dict name -> DeviceArray (T, R, C), meta)``; feed the three
	image groups to :class:`photometry_amd.pipeline.FrameStack` (``from_device``) for the stamp cutter.
	"""
	meta = read_header(path)
	out = {}
	for name in groups:
		out[name] = upload_group(ctx, open_group(path, name, meta))
	return out, meta
