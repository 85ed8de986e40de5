# -*- coding: utf-8 -*-
"""
GPU parity of the stamp cutter (BasePhotometry._load_cube, BasePhotometry.py:720-742) through the
C ABI: bit-exact against the golden cubes cut by the reference itself and against the oracle for stamp widths on both
sides of the 16-lane segment, cadence counts off the 64-block, pixel offsets and stamps that stick out of the frame.
"""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
	from photometry_amd.device import Context
	c = Context(0)
	yield c
	c.close()


def test_golden_reference_load_cube(ctx, golden_dir):
	from photometry_amd import engine
	g = np.load(os.path.join(golden_dir, 'golden_cutout.npz'))
	H, W = g['cubes'].shape[1:3]
	cube = engine.cut_stamps(ctx, ctx.array(g['frames']), ctx.array(g['stamps']), H, W, *[int(v) for v in g['offsets']])
	ctx.sync()
	np.testing.assert_array_equal(cube.to_host(), g['cubes'])


# the library cuts dense batches (stamps covering at least an eighth of the frame) tile by tile from LDS and sparse ones stamp by
# stamp: the first four and the sixth case take the tile-major path (tile edges at column 64 / 128 and every second row, a
# last tile cut by the frame), the fifth and the last the per-stamp gather
@pytest.mark.parametrize('T,R,C,H,W,n', [(64, 30, 30, 15, 15, 25), (129, 50, 70, 11, 17, 25), (200, 64, 64, 21, 16, 25), (7, 20, 20, 5, 3, 25),
	(70, 200, 300, 15, 15, 5), (131, 97, 203, 15, 15, 400), (65, 31, 400, 9, 70, 2)])
def test_against_oracle(ctx, T, R, C, H, W, n):
	from photometry_amd import engine
	from oracle import cutout
	rng = np.random.default_rng(T + W)
	frames = rng.normal(0, 1, (T, R, C)).astype('float32')
	frames[rng.random((T, R, C)) < 0.02] = np.nan
	r0 = rng.integers(-3, R - H + 4, n)
	c0 = rng.integers(-3 + 44, C - W + 4 + 44, n)
	stamps = np.stack((r0, r0 + H, c0, c0 + W), axis=1).astype('int32')
	stamps[0] = (0, H, 44, 44 + W)
	stamps[1] = (R - H, R, 44 + C - W, 44 + C)
	# the cube is handed over full of 0xFF bytes (NaN): the cutter must write every element, the padding of the time axis (cadences
	# T .. t_pitch of every pixel) as zeros -- callers do not clear the cube
	from photometry_amd.device import DeviceCube
	out = DeviceCube(ctx, n, T, H, W)
	out.data.fill_bytes(255)
	d_frames, d_stamps = ctx.array(frames), ctx.array(stamps)
	cube = engine.cut_stamps(ctx, d_frames, d_stamps, H, W, 0, 44, out=out)
	ctx.sync()
	full = cube.data.to_host()
	assert full.shape == (n, H, W, cube.t_pitch)
	assert np.all(full[..., T:] == 0)
	got = full[..., :T]
	for i in range(n):
		np.testing.assert_array_equal(got[i], cutout.load_cube(frames, tuple(stamps[i]), 0, 44), err_msg=str(stamps[i]))


def test_cut_then_photometry_equals_uploaded_cubes(ctx):
	"""Cubes cut on the device feed the hot path exactly like cubes uploaded from the host."""
	from photometry_amd import simulate, engine, pipeline
	from photometry_amd.device import DeviceCube
	rng = np.random.default_rng(3)
	s = simulate.make_scene(6, 40, 15, 15, seed=8)
	simulate.fill_cubes(s)
	s.aperture = None
	# paste the stamps into one frame stack, non-overlapping
	R, C, T = 20, 6 * 16 + 4, 40
	names = ('images', 'images_err', 'backgrounds')
	frames = {k: rng.normal(0, 1, (T, R, C)).astype('float32') for k in names}
	stamps = np.array(s.stamps)
	for i in range(6):
		stamps[i] = (2, 17, 44 + 2 + 16 * i, 44 + 2 + 16 * i + 15)
		for k in names:
			frames[k][:, 2:17, 2 + 16 * i:2 + 16 * i + 15] = np.moveaxis(getattr(s, k)[i], 2, 0)
	ref = pipeline.run_aperture(ctx, s)
	dstamps = ctx.array(stamps.astype('int32'))
	cubes = {k: engine.cut_stamps(ctx, ctx.array(frames[k]), dstamps, 15, 15, 0, 44) for k in names}
	got = pipeline.run_aperture(ctx, s, cubes=cubes)
	for k in ('sumimage', 'mask', 'status', 'flux', 'flux_err', 'flux_background', 'diagnostics'):
		np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


def test_cutout_from_a_stack_file(tmp_path):
	"""File -> HBM -> stamp cutter: the golden frames written to a .tpstack file, streamed in through the pinned double buffer
	(several chunks), cut on the device == the reference's _load_cube on the same frames (golden_cutout.npz)."""
	import os
	from photometry_amd import engine, frameio
	from photometry_amd.device import Context
	g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_cutout.npz'))
	frames = np.ascontiguousarray(g['frames'])           # (T, R, C)
	path = str(tmp_path / 'golden.tpstack')
	frameio.write_stack(path, {'images': frames.astype('float32')}, row_offset=int(g['offsets'][0]), col_offset=int(g['offsets'][1]))
	ctx = Context(0)
	meta = frameio.read_header(path)
	dev = frameio.upload_group(ctx, frameio.open_group(path, 'images', meta), frames_per_chunk=9)
	np.testing.assert_array_equal(dev.to_host(), frames.astype('float32'))
	stamps = np.asarray(g['stamps'], dtype='int32')
	shapes = {(s[1] - s[0], s[3] - s[2]) for s in stamps}
	for (H, W) in shapes:
		idx = [i for i, s in enumerate(stamps) if (s[1] - s[0], s[3] - s[2]) == (H, W)]
		cube = engine.cut_stamps(ctx, dev, ctx.array(stamps[idx]), H, W, meta['row_offset'], meta['col_offset']).to_host()
		for j, i in enumerate(idx):
			np.testing.assert_array_equal(cube[j], g['cubes'][i])
	ctx.close()


def test_multi_stack_cut_equals_one_stack_at_a_time(ctx):
	"""tp_cut_stamps_multi (one binning of the stamps, one launch for the three image groups of a CCD) against tp_cut_stamps stack by
	stack: tile-major path and per-stamp gather, stamps reaching over the frame edge."""
	from photometry_amd import engine
	rng = np.random.default_rng(17)
	for (T, R, C, H, W, n) in ((70, 60, 130, 15, 15, 300), (33, 200, 300, 21, 11, 4)):
		stacks = [ctx.array(rng.normal(0, 1, (T, R, C)).astype('float32')) for _ in range(3)]
		r0 = rng.integers(-3, R - H + 4, n)
		c0 = rng.integers(-3 + 44, C - W + 4 + 44, n)
		d_stamps = ctx.array(np.stack((r0, r0 + H, c0, c0 + W), axis=1).astype('int32'))
		together = engine.cut_stamps_multi(ctx, stacks, d_stamps, H, W, 0, 44)
		ctx.sync()
		for k in range(3):
			alone = engine.cut_stamps(ctx, stacks[k], d_stamps, H, W, 0, 44)
			ctx.sync()
			np.testing.assert_array_equal(together[k].data.to_host(), alone.data.to_host())


def test_masked_cut_writes_the_in_mask_rows_only(ctx):
	"""tp_cut_stamps_masked (the tile lists hold in-mask pixels instead of stamps): the rows of in-mask pixels equal the full cut's,
	every other row of the cube is left as it was; stamps that reach beyond the frame (in-mask pixels there are NaN), an empty mask,
	a full mask."""
	import ctypes
	from photometry_amd import engine
	from photometry_amd.device import DeviceCube
	rng = np.random.default_rng(23)
	for (T, R, C, H, W, n) in ((70, 60, 130, 15, 15, 300), (33, 64, 64, 11, 7, 40)):
		stacks = [ctx.array(rng.normal(0, 1, (T, R, C)).astype('float32')) for _ in range(2)]
		r0 = rng.integers(-3, R - H + 4, n)
		c0 = rng.integers(-3 + 44, C - W + 4 + 44, n)
		d_stamps = ctx.array(np.stack((r0, r0 + H, c0, c0 + W), axis=1).astype('int32'))
		mask = (rng.random((n, H, W)) < 0.16).astype('uint8')
		mask[0] = 0
		mask[1] = 1
		d_mask = ctx.array(mask)
		full = engine.cut_stamps_multi(ctx, stacks, d_stamps, H, W, 0, 44)
		outs = [DeviceCube(ctx, n, T, H, W) for _ in stacks]
		for o in outs:
			o.data.fill_bytes(0x7b)                  # a pattern no cut produces
		desc = outs[0].desc
		fp = (ctypes.c_void_p * 2)(*[f.ptr for f in stacks])
		cp = (ctypes.c_void_p * 2)(*[o.ptr for o in outs])
		ctx._check(ctx.lib.tp_cut_stamps_masked(ctx.handle, 2, fp, T, R, C, C, R * C, 0, 44, d_stamps.ptr, ctypes.byref(desc), d_mask.ptr, cp))
		ctx.sync()
		pattern = np.array([0x7b7b7b7b], dtype='uint32').view('float32')[0]
		for k in range(2):
			got, want = outs[k].data.to_host(), full[k].data.to_host()
			m = mask.astype(bool)
			np.testing.assert_array_equal(got[m], want[m])
			assert np.all(got[~m] == pattern)
