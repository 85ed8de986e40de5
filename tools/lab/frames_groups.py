import os, sys, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from photometry_amd import pipeline
from photometry_amd.device import Context
from benchlib.legs.frames import synthetic_region
frames, tstamp, quality, cat, targets = synthetic_region(np, 2500, 512, 200, 8)
ctx = Context(0)
stack = pipeline.FrameStack(ctx, frames, 0, 44)
res = pipeline.aperture_frames(ctx, stack, targets, cat, tstamp, quality)
for g in res.groups:
    m = g['mask']
    print('group: targets', m.shape[0], 'stamp', m.shape[1], 'x', m.shape[2], 'ncat', len(g['cat_starid']))
print('resizes histogram', np.bincount(res.stamp_resizes))
