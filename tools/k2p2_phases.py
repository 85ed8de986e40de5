#!/usr/bin/env python3
"""Diagnostic: per-phase cycle counts of the K2P2 mask kernel (needs a library built with -DTP_LAB_K2P2_TIMING)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('TP_LAB_LIB'):
	from photometry_amd import _lib
	_lib.LIB_PATH = os.environ['TP_LAB_LIB']
import numpy as np
from photometry_amd import simulate, engine, pipeline
from photometry_amd.device import Context

Nt = int(os.environ.get('NT', 2048))
ctx = Context(0)
scene = simulate.make_scene(Nt, 64, 15, 15, seed=1000)
simulate.fill_cubes(scene)
scene.aperture = None
batch = pipeline.ApertureBatch(ctx, scene)
work = pipeline.ApertureWork(ctx, batch)
work.diag = ctx.zeros((Nt, 16), 'float64')   # the timing build writes 16 doubles per target
engine.sumimage(ctx, batch.images, batch.quality, out=work.sumimage)
engine.k2p2_masks(ctx, batch, work)
ctx.sync()
d = work.diag.to_host()
names = {1: 'threshold (KDE, Powell, MAD)', 2: 'DBSCAN labels', 3: 'saturated pre-pass', 4: 'gaussian filter + peaks', 5: 'catalogue / local maxima',
	6: 'marker labels', 7: 'watershed', 8: 'relabel', 9: 'cluster sizes', 10: 'mask assembly', 11: 'hole fill / overflow', 12: 'selection, flags, contamination'}
tot = d[:, 1:13].sum()
for i in range(1, 13):
	print(f'{names[i]:34s} {d[:, i].mean():10.0f} cycles per target  {100 * d[:, i].sum() / tot:5.1f} %')
print('total', d[:, 1:13].sum(axis=1).mean(), 'cycles per target (clock64 ticks)')
