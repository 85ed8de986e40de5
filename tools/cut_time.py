#!/usr/bin/env python3
"""Diagnostic: time of the stamp cutter on 10 000 15x15 stamps x 1300 frames (frame side FR, default 1024 and 2048)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photometry_amd import engine
from photometry_amd.device import Context, DeviceCube

ctx = Context(0)
Nt, T, H, W = 10000, 1300, 15, 15
out = DeviceCube(ctx, Nt, T, H, W)
for FR in (512, 1024, 2048):
	frames = ctx.zeros((T, FR, FR), 'float32')
	rng = np.random.default_rng(1)
	r0 = rng.integers(0, FR - H, Nt); c0 = rng.integers(0, FR - W, Nt)
	for order in ('random', 'sorted'):
		st = np.stack((r0, r0 + H, c0 + 44, c0 + 44 + W), axis=1).astype('int32')
		if order == 'sorted':
			st = st[np.lexsort((st[:, 2], st[:, 0]))]
		d = ctx.array(st)
		for _ in range(2):
			engine.cut_stamps(ctx, frames, d, H, W, 0, 44, out=out)
		ctx.sync()
		t0 = time.perf_counter()
		for _ in range(5):
			engine.cut_stamps(ctx, frames, d, H, W, 0, 44, out=out)
		ctx.sync()
		ms = (time.perf_counter() - t0) / 5 * 1e3
		print(f'frame {FR} stamps {order}: {ms:.3f} ms  ({2*Nt*H*W*T*4/ms/1e6:.0f} GB/s of stamp bytes)', flush=True)
	frames.free()
