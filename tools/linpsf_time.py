#!/usr/bin/env python3
"""Diagnostic: per-kernel time of the LinPSF fit on the C3 workload (NT targets)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('TP_LAB_LIB'):
	from photometry_amd import _lib
	_lib.LIB_PATH = os.environ['TP_LAB_LIB']
from benchlib.legs.linpsf import leg_linpsf
from photometry_amd import simulate, engine, pipeline
from photometry_amd.device import Context
import numpy as np

class A: pass
args = A(); args.steps = 3; args.warmup = 1; args.seed = 1; args.cpu_sample = 0; args.linpsf_drift = int(os.environ.get('DRIFT', 0))
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
if os.environ.get('LINPSF_PATH'):
	engine.linpsf_set_path(ctx, int(os.environ['LINPSF_PATH']))
res = leg_linpsf(ctx, scene, cubes, work, args, Nt, T, H, W, np, engine, pipeline)
print(res['value'], 'targets/s', res['ms_per_step'], 'ms/step')
for k, v in res['kernels'].items():
	print('  ', k, v)
