#!/usr/bin/env python3
"""lab: the tile-major stamp cutter with other tile shapes (lib_c_<rows>_<cols>_<frames>.so built by build_variants.sh with
-DTP_CUT_TILE_ROWS/_COLS/_CAD); each variant in its own process, checked against the product library's output first."""
import os, sys, subprocess, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
	sys.path.insert(0, ROOT)
	from photometry_amd import _lib
	if sys.argv[1] != 'product':
		_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'lab', f'lib_{sys.argv[1]}.so')
	from photometry_amd import engine
	from photometry_amd.device import Context, DeviceCube
	ctx = Context(0)
	Nt, T, H, W = 10000, 1300, 15, 15
	out = DeviceCube(ctx, Nt, T, H, W)
	res = []
	for FR in (512, 1024, 2048):
		rng = np.random.default_rng(1)
		frames = ctx.array(rng.normal(size=(8, FR, FR)).astype('float32')) if FR == 512 else None
		r0 = rng.integers(0, FR - H, Nt); c0 = rng.integers(0, FR - W, Nt)
		st = np.stack((r0, r0 + H, c0 + 44, c0 + 44 + W), axis=1).astype('int32')
		d = ctx.array(st)
		if frames is not None: # a small correctness check on real values
			small = engine.cut_stamps(ctx, frames, d, H, W, 0, 44)
			ctx.sync()
			got = small.to_host()[::997]
			fh = frames.to_host()
			for n, i in enumerate(range(0, Nt, 997)):
				assert np.array_equal(got[n], np.moveaxis(fh[:, r0[i]:r0[i] + H, c0[i]:c0[i] + W], 0, 2)), (sys.argv[1], i)
			small.free(); frames.free()
		frames = ctx.zeros((T, FR, FR), 'float32')
		for _ in range(2):
			engine.cut_stamps(ctx, frames, d, H, W, 0, 44, out=out)
		ctx.sync()
		t0 = time.perf_counter()
		for _ in range(5):
			engine.cut_stamps(ctx, frames, d, H, W, 0, 44, out=out)
		ctx.sync()
		res.append(round((time.perf_counter() - t0) / 5 * 1e3, 3))
		frames.free()
	print(sys.argv[1], 'ms for 512 / 1024 / 2048 stacks:', res, flush=True)
else:
	import glob
	for v in ['product'] + sorted(os.path.basename(p)[4:-3] for p in glob.glob(os.path.join(ROOT, 'tools', 'lab', 'lib_c_*.so'))):
		subprocess.run([sys.executable, __file__, v])
