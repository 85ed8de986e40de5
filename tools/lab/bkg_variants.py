#!/usr/bin/env python3
"""lab: time B* variants (each in its own process: one library per process)."""
import os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
	sys.path.insert(0, ROOT)
	from photometry_amd import _lib
	_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'lab', f'lib_{sys.argv[1]}.so')
	from photometry_amd import simulate, engine
	from photometry_amd.device import Context
	ctx = Context(0)
	Nt = 10000
	scene = simulate.make_scene(Nt, 1300, 15, 15, seed=1000)
	raw = engine.synth_fill(ctx, scene, images=False, images_err=False, backgrounds=False, raw=True)['raw']
	out = ctx.zeros((Nt, raw.t_pitch), 'float32')
	for _ in range(2):
		engine.background_stamp(ctx, raw, out=out)
	ctx.sync()
	t0 = time.perf_counter()
	for _ in range(5):
		engine.background_stamp(ctx, raw, out=out)
	ctx.sync()
	print(sys.argv[1], 'B* ms', round((time.perf_counter() - t0) / 5 * 1e3, 3), flush=True)
else:
	import glob
	for v in sorted(os.path.basename(p)[4:-3] for p in glob.glob(os.path.join(ROOT, 'tools', 'lab', 'lib_*.so'))):
		subprocess.run([sys.executable, __file__, v])
