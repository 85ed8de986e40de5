# -*- coding: utf-8 -*-
import time


def leg_frames(ctx, args, T, np, pipeline):
	"""
	The batched drop-in entry as a scheduler would call it: a CCD region's frame stacks (images, errors, backgrounds) resident
	in HBM, ``tessphot_frames`` from target list to per-target results -- default stamps, catalogue selection, stamp cuts on
	the device, the fused pass, the stamp-resize rounds, diagnostics, download and the host-side bookkeeping per target.
	"""
	from photometry_amd import tessphot_frames
	N, FR, Tn = args.frames_targets, 512, 100
	rng = np.random.default_rng(args.seed + 7)
	rows, cols, tmag = rng.uniform(12, FR - 12, N), rng.uniform(12, FR - 12, N), rng.uniform(9.0, 14.0, N)
	img = np.zeros((FR, FR))
	yy, xx = np.mgrid[-4:5, -4:5]
	for r, c, m in zip(rows, cols, tmag):
		ri, ci = int(round(r)), int(round(c))
		img[ri - 4:ri + 5, ci - 4:ci + 5] += 10**(-0.4 * (m - 20.451)) * np.exp(-0.5 * ((yy + ri - r)**2 + (xx + ci - c)**2) / 0.81) / (2 * np.pi * 0.81)
	reps = (T + Tn - 1) // Tn
	base = (img[None] * (1 + 1e-3 * rng.normal(size=Tn))[:, None, None]).astype('float32')
	noise = np.sqrt(np.abs(base) + 200.0).astype('float32')
	images = (base + 30.0 + rng.standard_normal(base.shape).astype('float32') * noise).astype('float32')
	frames = {'images': np.tile(images, (reps, 1, 1))[:T], 'images_err': np.tile(noise, (reps, 1, 1))[:T],
		'backgrounds': np.full((T, FR, FR), 100.0, dtype='float32')}
	del base, noise, images
	tstamp = 1500.0 + np.arange(T) * 1800.0 / 86400.0
	quality = np.zeros(T, dtype='int32')
	cat = {'starid': np.arange(N, dtype='int64') + 1, 'tmag': tmag.astype('float32'), 'row': rows.astype('float32'), 'column': (cols + 44).astype('float32')}
	targets = {'starid': cat['starid'].copy(), 'tmag': tmag, 'row': rows, 'column': cols + 44}
	stack = pipeline.FrameStack(ctx, frames, 0, 44)
	del frames
	ctx.sync()
	# the steady state of a scheduler that calls the entry CCD after CCD: the page-locked result buffers and the device blocks of
	# the first call are in the context's pools when the timed one runs
	tessphot_frames(ctx, stack, targets, cat, tstamp, quality)
	t0 = time.perf_counter()
	out = tessphot_frames(ctx, stack, targets, cat, tstamp, quality)
	good = int(np.sum((out.status == 1) | (out.status == 3)))
	resized = int(np.sum(out.stamp_resizes > 0))
	dt = time.perf_counter() - t0
	# every per-target object as well (what the list-based entry of round 2 built unconditionally)
	t1 = time.perf_counter()
	n_obj = sum(1 for b in out if b.status.value in (1, 3))
	dobj = time.perf_counter() - t1
	return {'what': f'tessphot_frames: {N} targets on a {FR} x {FR} x {T} region resident in HBM (three frame stacks) -> columnar results '
		'(status, stamp, resizes, diagnostics, light curves, masks in arrays; per-target objects on demand): stamp cuts, fused pass, '
		'stamp-resize rounds and diagnostics on the device, default stamps / catalogue selection / decisions on the host (one Python process)',
		'targets_per_s': N / dt, 'seconds': dt, 'ok_or_warning': good, 'targets_resized': resized,
		'with_every_per_target_object': {'targets_per_s': N / (dt + dobj), 'seconds_for_the_objects': dobj, 'ok_or_warning': n_obj}}
