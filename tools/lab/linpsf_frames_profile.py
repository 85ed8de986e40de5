import os, sys, time, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import numpy as np
from photometry_amd import pipeline, simulate, psf as hpsf
from photometry_amd.device import Context, bind_host_to_device
from benchlib.legs.frames import synthetic_region
bind_host_to_device(0)
ctx = Context(0)
N, FR, T = 2000, 512, 1300
frames, tstamp, quality, cat, targets = synthetic_region(np, N, FR, T, 12)
prf = simulate.synthetic_prf(seed=1)
model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
stack = pipeline.FrameStack(ctx, frames, 0, 44)
ctx.sync()
for _ in range(2):
	pipeline.linpsf_frames(ctx, stack, targets, cat, tstamp, quality, model)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
lin = pipeline.linpsf_frames(ctx, stack, targets, cat, tstamp, quality, model)
pr.disable()
print('linpsf_frames %.2f ms' % ((time.perf_counter() - t0) * 1e3))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(25); print(s.getvalue()[:4500])
