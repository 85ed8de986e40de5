# -*- coding: utf-8 -*-
import time


def leg_end_to_end(ctx, scene, cubes, args, T, H, W, np, engine, pipeline, Context, DeviceCube):
	"""
	SURVEY 8d timing (ii): the cubes start in HOST memory, as a drop-in plugin receives them from BasePhotometry._load_cube.
	Pinned staging buffers, chunks of targets, the upload of chunk i+1 on a second stream while chunk i is processed, the
	light curves copied back asynchronously.  PCIe-bound by construction (2.35 MB in per target, 52 KB out).
	"""
	n = min(args.e2e_targets, scene.n_targets)
	chunk = 256
	n = max(chunk, n // chunk * chunk)
	sub = scene.subset(slice(0, n))
	up = Context(ctx.device, high_priority=False)
	host = {}
	for key in ('raw', 'raw_err'):
		host[key] = ctx.pinned((n, H, W, T), 'float32')
		full = np.empty((chunk, H, W, cubes[key].t_pitch), dtype='float32')
		for a in range(0, n, chunk):
			ctx._check(ctx.lib.tp_memcpy_d2h(ctx.handle, full.ctypes.data, cubes[key].slice0(a, chunk).ptr, full.nbytes))
			host[key].array[a:a + chunk] = full[..., :T]
	out_host = ctx.pinned((n // chunk, 5, chunk, T), 'float64')
	bufs = []
	for _ in range(2):
		dc = {k: DeviceCube(ctx, chunk, T, H, W) for k in ('raw', 'raw_err')}
		for c in dc.values():
			c.data.fill_bytes(0)
		bufs.append(dc)
	batches, works = [], []
	for a in range(0, n, chunk):
		b = pipeline.ApertureBatch(ctx, sub.subset(slice(a, a + chunk)), cubes=bufs[(a // chunk) % 2])
		batches.append(b)
		works.append(pipeline.ApertureWork(ctx, b))
	ev_up = [ctx.event() for _ in range(2)]
	ev_used = [ctx.event() for _ in range(2)]

	def run():
		for i, a in enumerate(range(0, n, chunk)):
			s = i % 2
			if i >= 2:
				up.wait_event(ev_used[s])
			for key in ('raw', 'raw_err'):
				bufs[s][key].upload_async(up, host[key], first_target=a)
			up.record(ev_up[s])
			ctx.wait_event(ev_up[s])
			pipeline.aperture_step(ctx, batches[i], works[i])
			ctx.record(ev_used[s])
			ctx.download_async(out_host, works[i].lc.block, host_offset=i * 5 * chunk * T * 8)
		up.sync()
		ctx.sync()

	run()
	t0 = time.perf_counter()
	reps = 2
	for _ in range(reps):
		run()
	dt = (time.perf_counter() - t0) / reps
	in_bytes = 2 * n * H * W * T * 4
	res = {'what': f'{n} targets whose raw + error cubes start in pinned host memory (reference (H, W, T) layout): chunks of {chunk} targets, '
		'H2D on a second stream overlapped with the step of the previous chunk, light curves copied back; file I/O excluded',
		'targets_per_s': n / dt, 'h2d_GBps': in_bytes / dt / 1e9, 'bound': 'PCIe (2.35 MB in per target)', 'seconds': dt}
	for h in list(host.values()) + [out_host]:
		h.free()
	up.close()
	return res
