#!/usr/bin/env python3
"""Diagnostic: cycles per phase of the mask builder, summed over the 10 000 targets of the BASELINE batch, in the stand-alone
kernel (tp_k2p2_masks).  Needs a scratch build with the in-kernel clocks (the hooks TP_K2P2_CLOCK of k2p2_core.h are empty in the product):
    SRC=k2p2.hip bash tools/lab/build_variants.sh "k2clk:-DTP_LAB_K2P2_CLOCK"
    TP_LAB_LIB=tools/lab/lib_k2clk.so H=25 W=25 NT=2000 python tools/k2p2_timing.py        (on the GPU box)"""
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
names = ['A2 threshold (sort, KDE, Powell / Brent, MAD)', 'idx, DBSCAN core, labels', 'per cluster: saturated pre-pass', 'per cluster: blur + peaks', 'per cluster: star match', 'per cluster: dedupe in saturated patches',
	'per cluster: label the markers', 'per cluster: watershed', 'per cluster: relabel', 'mask assembly', 'minimum aperture, edges, contamination, outputs']
print('clusters per target %.1f' % c[11])
print('A2 in detail (ticks per target): sort %.0f, bandwidth %.0f, KDE grid + argmax %.0f, Powell / Brent %.0f, MAD + CUT %.0f' % (c[12], c[13], c[14], c[15], c[0]))
c[0] += c[12] + c[13] + c[14] + c[15]
c = c[:len(names)]
tot = c.sum()
for n, v in zip(names, c):
	print('%-40s %9.0f ticks  %5.1f %%' % (n, v, 100 * v / tot))
print('total per target %.0f ticks' % tot)
import time
t0 = time.perf_counter()
for _ in range(5):
	engine.k2p2_masks(ctx, batch, work)
ctx.sync()
print('launch of %d targets of %d x %d: %.3f ms; statuses' % (Nt, H, W, (time.perf_counter() - t0) / 5 * 1e3), np.bincount(work.status.to_host(), minlength=4))
