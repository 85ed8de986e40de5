#!/usr/bin/env python3
"""Diagnostic: the LinPSF step on the C3 workload with the vector-ALU kernels (path 0) and with the matrix-core fit (path 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('TP_LAB_LIB'):
	from photometry_amd import _lib
	_lib.LIB_PATH = os.environ['TP_LAB_LIB']
import bench
from photometry_amd import simulate, engine, pipeline
from photometry_amd.device import Context
import numpy as np

class A: pass
args = A(); args.steps = 3; args.warmup = 1; args.seed = 1; args.cpu_sample = 0
Nt, T, H, W = int(os.environ.get('NT', 10000)), 1300, 15, 15
ctx = Context(0)
scene = simulate.make_scene(Nt, T, H, W, seed=1000)
scene.aperture = None
cubes = engine.synth_fill(ctx, scene, images=False, images_err=False, backgrounds=False, raw=True)
batch = pipeline.ApertureBatch(ctx, scene, cubes={'raw': cubes['raw'], 'raw_err': cubes['raw']})
work = pipeline.ApertureWork(ctx, batch, packed=True)
engine.background_stamp(ctx, batch.images, out=work.bkg_raw)
engine.smooth_time(ctx, work.bkg_raw, batch.n_cad, batch.time_smooth, out=work.bkg)
ctx.sync()
outs = {}
if os.environ.get('MAXS'):
	_step = pipeline.linpsf_step
	def _limited(ctx, batch, *a, **k):
		batch.max_stars = int(os.environ['MAXS'])
		return _step(ctx, batch, *a, **k)
	pipeline.linpsf_step = _limited
for path in [int(x) for x in os.environ.get('PATHS', '0,1,0,1').split(',')]:
	engine.linpsf_set_path(ctx, path)
	res = bench.leg_linpsf(ctx, scene, cubes, work, args, Nt, T, H, W, np, engine, pipeline)
	print('path', path, res['value'], 'targets/s', res['ms_per_step'], 'ms/step', flush=True)
	for k, v in res['kernels'].items():
		print('    ', k, v)
	outs[path] = res
