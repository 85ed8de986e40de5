#!/usr/bin/env python3
"""Diagnostic: the configs[2] step on the C3 raw + error cubes: per-kernel HIP-event times and the step's wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('TP_LAB_LIB'):
	from photometry_amd import _lib
	_lib.LIB_PATH = os.environ['TP_LAB_LIB']
from photometry_amd import simulate, engine, pipeline
from photometry_amd.device import Context

ctx = Context(0)
Nt = int(os.environ.get('NT', 10000))
T = int(os.environ.get('T', 1300))
scene = simulate.make_scene(Nt, T, 15, 15, seed=1000)
if os.environ.get('SORT'):   # experiment: the targets in order of brightness (a proxy for the work of the mask + extraction launch)
	import numpy as np
	o = np.argsort(scene.target_tmag, kind='stable')
	scene = scene.subset(o if os.environ['SORT'] == 'asc' else o[::-1])
scene.aperture = None
if 'CADENCE' in os.environ:
	scene.cadence_s = int(os.environ['CADENCE'])
cubes = engine.synth_fill(ctx, scene, images=False, images_err=True, backgrounds=False, raw=True)
batch = pipeline.ApertureBatch(ctx, scene, cubes={'raw': cubes['raw'], 'raw_err': cubes['images_err']})
work = pipeline.ApertureWork(ctx, batch)
for _ in range(2):
	pipeline.aperture_step(ctx, batch, work)
ctx.sync()
ctx.profile(True)
ctx.profile_reset()
n = int(os.environ.get('STEPS', 8))
t0 = time.perf_counter()
for _ in range(n):
	pipeline.aperture_step(ctx, batch, work)
ctx.sync()
wall = (time.perf_counter() - t0) / n * 1e3
ctx.profile(False)
rep = ctx.profile_report()
print('step ms', round(wall, 3), {k: round(v[1] / max(v[0], 1), 3) for k, v in rep.items() if v[0]})
# FCLK=1: phases of the fused launch from in-kernel clocks -- a scratch build (tools/lab/clock_hooks.patch: made against the round-4 sources, to be re-made before use):
# SRC=fused.hip bash tools/lab/build_variants.sh "fclk:-DTP_LAB_FUSED_CLOCK"; git apply -R tools/lab/clock_hooks.patch
if os.environ.get('TP_LAB_LIB') and os.environ.get('FCLK'):
	import ctypes
	buf = (ctypes.c_ulonglong * 8)()
	ctx.lib.tp_lab_fused_clocks(buf, 1)
	pipeline.aperture_step(ctx, batch, work)
	ctx.sync()
	ctx.lib.tp_lab_fused_clocks(buf, 0)
	c = list(buf)
	print('fused kernel, per target (cycles): sum image to LDS %.0f, mask builder %.0f, mask list + series staging %.0f, extraction %.0f; wavefronts %d' % (c[0] / Nt, c[1] / Nt, c[2] / Nt, c[3] / Nt, c[4]))
