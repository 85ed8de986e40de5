#!/usr/bin/env python3
"""lab: time tp_background_sumimage variants (each in its own process: one library per process)."""
import os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
	sys.path.insert(0, ROOT)
	from photometry_amd import _lib
	_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'lab', f'lib_{sys.argv[1]}.so')
	from photometry_amd import simulate, engine
	from photometry_amd.device import Context
	ctx = Context(0)
	Nt = int(os.environ.get("NT", 10000))
	scene = simulate.make_scene(Nt, 1300, 15, 15, seed=1000)
	raw = engine.synth_fill(ctx, scene, images=False, images_err=False, backgrounds=False, raw=True)['raw']
	q = ctx.array(scene.quality.astype('int32'))
	outs = engine.background_sumimage(ctx, raw, q, 3)
	for _ in range(2):
		engine.background_sumimage(ctx, raw, q, 3, bkg_raw=outs[0], bkg=outs[1], sumimage=outs[2])
	ctx.sync()
	ctx.profile(True)
	ctx.profile_reset()
	t0 = time.perf_counter()
	for _ in range(5):
		engine.background_sumimage(ctx, raw, q, 3, bkg_raw=outs[0], bkg=outs[1], sumimage=outs[2])
	ctx.sync()
	wall = (time.perf_counter() - t0) / 5 * 1e3
	ctx.profile(False)
	print({k: round(v[1] / max(v[0], 1), 3) for k, v in ctx.profile_report().items() if v[0]}, 'wall', round(wall, 3))
	import hashlib
	h = hashlib.sha256(outs[0].to_host().tobytes() + outs[1].to_host().tobytes() + outs[2].to_host().tobytes()).hexdigest()[:12]
	print(sys.argv[1], 'outputs', h, flush=True)
else:
	import glob
	for v in sorted(os.path.basename(p)[4:-3] for p in glob.glob(os.path.join(ROOT, 'tools', 'lab', 'lib_*.so'))):
		subprocess.run([sys.executable, __file__, v])
