import sys, time, numpy as np
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
engine.sumimage(ctx, batch.images, batch.quality, out=work.sumimage)
def run(cut=None, n=5):
    engine.k2p2_masks(ctx, batch, work, cut_override=cut); ctx.sync()
    ctx.profile(True); ctx.profile_reset()
    for _ in range(n): engine.k2p2_masks(ctx, batch, work, cut_override=cut)
    ctx.sync(); r = ctx.profile_report(); ctx.profile(False)
    return r['tp_k2p2_kernel'][1] / r['tp_k2p2_kernel'][0]
t_full = run()
diag = work.diag.to_host()
cut = ctx.array(np.ascontiguousarray(diag[:, 0]))
t_cut = run(cut)
print('full %.3f ms   with CUT given (no A2) %.3f ms   -> A2 = %.3f ms' % (t_full, t_cut, t_full - t_cut))
huge = ctx.array(np.full(Nt, 1e30))
print('CUT=1e30 (no stars -> min aperture only) %.3f ms' % run(huge))
