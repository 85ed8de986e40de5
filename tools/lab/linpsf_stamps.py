#!/usr/bin/env python3
"""lab: in-kernel s_memtime stamps of the matrix-core LinPSF fit (library built with -DTP_LAB_STAMP after
``git apply tools/lab/linpsf_lab_hooks.patch``: the hooks are not in the product source), summed per star count."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from photometry_amd import _lib
_lib.LIB_PATH = os.environ['TP_LAB_LIB']
import numpy as np
from photometry_amd import simulate, engine, pipeline, psf as hpsf
from photometry_amd.device import Context
Nt, T, H, W = 10000, 1300, 15, 15
ctx = Context(0)
scene = simulate.make_scene(Nt, T, H, W, seed=1000)
scene.aperture = None
cubes = engine.synth_fill(ctx, scene, images=False, images_err=False, backgrounds=False, raw=True)
prf = simulate.synthetic_prf(seed=1)
model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
batch = pipeline.LinPSFBatch(ctx, scene, model, images=cubes['raw'])
engine.linpsf_set_path(ctx, 1)
pipeline.linpsf_step(ctx, batch)
ctx.sync()
batch.out.flux_err = ctx.zeros((Nt, T), 'float64')
pipeline.linpsf_step(ctx, batch)
ctx.sync()
st = batch.out.flux_err.to_host()[:, :6]
ns = np.diff(batch.star_offsets_h)
for s in sorted(set(ns)):
	m = ns == s
	tot = st[m].sum(axis=0)
	print('stars', s, 'targets', int(m.sum()), 'waves', int(tot[5]), 'cycles per wave: bgen %.0f tiles %.0f reduce %.0f solve %.0f total %.0f' % tuple(tot[:5] / tot[5]))
