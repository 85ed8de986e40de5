import sys, os, numpy as np
os.environ['TP_K2P2_TIMING'] = '1'
sys.path.insert(0, '.')
from photometry_amd import simulate, engine, pipeline
from photometry_amd.device import Context
ctx = Context(0)
Nt, T, H, W = 10000, 200, 15, 15
scene = simulate.make_scene(Nt, T, H, W, seed=1000)
scene.aperture = None
cubes = engine.synth_fill(ctx, scene)
batch = pipeline.ApertureBatch(ctx, scene, cubes=cubes)
work = pipeline.ApertureWork(ctx, batch)
work.diag = ctx.zeros((Nt, 16), 'float64')
engine.sumimage(ctx, batch.images, batch.quality, out=work.sumimage)
engine.k2p2_masks(ctx, batch, work); ctx.sync()
t = work.diag.to_host()
names = ['-', 'threshold', 'idx+core+label', 'sat prepass', 'blur+peaks', 'star match', 'dedupe+markers', 'watershed', 'relabel', 'A5 count', 'hole fill', 'A5 sat', 'target check', 'final']
tot = t.sum(axis=1).mean()
for i, n in enumerate(names):
    if i < 16 and t[:, i].mean() > 0: print('%-16s %9.0f cycles  %5.1f %%' % (n, t[:, i].mean(), 100*t[:, i].mean()/tot))
print('total per target %.0f cycles' % tot)
