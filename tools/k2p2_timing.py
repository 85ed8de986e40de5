#!/usr/bin/env python3
"""Diagnostic: cycles per phase of the mask builder, summed over the 10 000 targets of the BASELINE batch, in the stand-alone
kernel (tp_k2p2_masks).  Needs a scratch build with the in-kernel clocks, which are NOT in the product sources:
    git apply tools/lab/clock_hooks.patch
    SRC=k2p2.hip bash tools/lab/build_variants.sh "k2clk:-DTP_LAB_K2P2_CLOCK"
    git apply -R tools/lab/clock_hooks.patch
    TP_LAB_LIB=tools/lab/lib_k2clk.so python tools/k2p2_timing.py        (on the GPU box)"""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('TP_LAB_LIB'):
	from photometry_amd import _lib
	_lib.LIB_PATH = os.environ['TP_LAB_LIB']
from photometry_amd import simulate, engine, pipeline
from photometry_amd.device import Context
ctx = Context(0)
Nt, T, H, W = int(os.environ.get('NT', 10000)), 200, int(os.environ.get('H', 15)), int(os.environ.get('W', 15))
scene = simulate.make_scene(Nt, T, H, W, seed=1000)
scene.aperture = None
cubes = engine.synth_fill(ctx, scene)
batch = pipeline.ApertureBatch(ctx, scene, cubes=cubes)
work = pipeline.ApertureWork(ctx, batch)
engine.sumimage(ctx, batch.images, batch.quality, out=work.sumimage)
engine.k2p2_masks(ctx, batch, work); ctx.sync()
buf = (ctypes.c_ulonglong * 24)()
ctx.lib.tp_lab_k2p2_clocks(buf, 1)
engine.k2p2_masks(ctx, batch, work); ctx.sync()
ctx.lib.tp_lab_k2p2_clocks(buf, 0)
c = np.array(list(buf), dtype='float64') / Nt
names = ['sort', 'bandwidth', 'KDE by DFT + argmax', 'Powell / Brent on the Gaussian sum', 'MAD + CUT', 'idx, DBSCAN core, labels', 'saturated pre-pass', 'blur + peaks',
	'label the markers', 'watershed + relabel', 'mask assembly', 'contamination + outputs', 'peak list', 'copy of the selected peaks', 'dedupe in saturated patches', 'minimum aperture + edges', 'star match loop', 'cat_in_mask', 'contamination: serial star loop', 'contamination: log10f / pow']
tot = c.sum()
for n, v in zip(names, c):
	print('%-40s %9.0f ticks  %5.1f %%' % (n, v, 100 * v / tot))
print('total per target %.0f ticks' % tot)
