# -*- coding: utf-8 -*-
import time


def synthetic_region(np, N, FR, T, seed):
	"""A CCD region of FR x FR pixels with N stars (Gaussian PRF, sigma 0.9 px), T frames: frame stacks, time, quality, catalogue, targets."""
	Tn = 100
	rng = np.random.default_rng(seed)
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
	tstamp = 1500.0 + np.arange(T) * 1800.0 / 86400.0
	quality = np.zeros(T, dtype='int32')
	cat = {'starid': np.arange(N, dtype='int64') + 1, 'tmag': tmag.astype('float32'), 'row': rows.astype('float32'), 'column': (cols + 44).astype('float32')}
	targets = {'starid': cat['starid'].copy(), 'tmag': tmag, 'row': rows, 'column': cols + 44}
	return frames, tstamp, quality, cat, targets


def leg_frames(ctx, args, T, np, pipeline, N=None, FR=512, NB=12, runs=14):
	"""
	The batched drop-in entry as a scheduler would call it: a CCD region's frame stacks (images, errors, backgrounds) resident
	in HBM, ``tessphot_frames`` from target list to per-target results -- default stamps, catalogue selection, stamp cuts on
	the device, the fused pass, the stamp-resize rounds, diagnostics, download and the host-side bookkeeping per target.
	"""
	from photometry_amd import tessphot_frames
	N = args.frames_targets if N is None else N
	frames, tstamp, quality, cat, targets = synthetic_region(np, N, FR, T, args.seed + 7)
	stack = pipeline.FrameStack(ctx, frames, 0, 44)
	del frames
	ctx.sync()
	# the steady state of a scheduler that calls the entry batch after batch: the page-locked result buffers and the device blocks of
	# the earlier calls are in the engine's pools (the allocation caches need two calls to hold every block a call takes: the third
	# call of a process is the first that allocates nothing); the median of three timed calls
	for _ in range(2):
		tessphot_frames(ctx, stack, targets, cat, tstamp, quality)
	times = []
	out = None
	for _ in range(3):
		out = None
		t0 = time.perf_counter()
		out = tessphot_frames(ctx, stack, targets, cat, tstamp, quality)
		good = int(np.sum((out.status == 1) | (out.status == 3)))
		resized = int(np.sum(out.stamp_resizes > 0))
		times.append(time.perf_counter() - t0)
	dt = sorted(times)[1]
	# every per-target object as well (what the list-based entry of round 2 built unconditionally)
	t1 = time.perf_counter()
	n_obj = sum(1 for b in out if b.status.value in (1, 3))
	dobj = time.perf_counter() - t1
	# consecutive batches (what a run over a CCD does): the same stars in another order per batch, four batches on the device at a
	# time, each result consumed (its counts read) and let go before the next is taken
	from photometry_amd import tessphot_frames_pipelined
	rng = np.random.default_rng(args.seed + 8)
	batches = []
	for _b in range(NB):
		sel = rng.permutation(N)
		batches.append({k: np.asarray(v)[sel] for k, v in targets.items()})
	out = None
	piped = {}
	import gc
	gc.collect()   # (the legs before this one leave a large heap: a full collection in the middle of a 70 ms timing is 10 % of it)
	reps = []
	okc = 0
	for rep in range(runs):
		t2 = time.perf_counter()
		okc = 0
		for res in tessphot_frames_pipelined(ctx, stack, iter(batches), cat, tstamp, quality, in_flight=4):
			okc += int(np.sum((res.status == 1) | (res.status == 3)))
			res = None
		reps.append(time.perf_counter() - t2)
	dp = sorted(reps[-5:])[2]      # median of the last five runs (the allocation pools of the four jobs' sixteen contexts fill over the first ones)
	piped = {'what': f'tessphot_frames_pipelined: {NB} consecutive batches of {N} targets of the same region, four jobs of the native engine in flight '
		'(submit / collect: the rounds of a batch are driven by a worker thread of the library; the first round of a batch runs under the '
		'latency-bound stamp-resize rounds of the others; every batch equals a call of its own: tests/test_gpu_resize.py)',
		'targets_per_s': NB * N / dp, 'seconds': dp, 'seconds_all_runs': reps, 'batches': NB, 'in_flight': 4, 'ok_or_warning': okc}
	return {'what': f'tessphot_frames: {N} targets on a {FR} x {FR} x {T} region resident in HBM (three frame stacks) -> columnar results '
		'(status, stamp, resizes, diagnostics, light curves, masks in arrays; per-target objects on demand): sum images cropped from the region\'s '
		'(BasePhotometry.py:1001-1006), masks, ONE cut of the in-mask rows of the three stacks, extraction, stamp-resize rounds and diagnostics on the device; '
		'catalogue selection and the plugin rules by a worker thread of the library (csrc/frames.cpp): the host submits the batch and collects it',
		'targets_per_s': N / dt, 'seconds': dt, 'seconds_all_calls': times, 'ok_or_warning': good, 'targets_resized': resized, 'engine': 'native (csrc/frames.cpp)',
		'with_every_per_target_object': {'targets_per_s': N / (dt + dobj), 'seconds_for_the_objects': dobj, 'ok_or_warning': n_obj},
		'pipelined': piped}


def leg_psf_frames(ctx, args, T, np, pipeline):
	"""
	The batched PSF entries: ``pipeline.linpsf_frames`` (LinPSFPhotometry for every target of a region resident in HBM) and
	``pipeline.psf_frames`` (PSFPhotometry), from the target list to columnar results -- default stamps, catalogue and star
	selection, stamp cuts on the device, P1 blend + fit, download.  cpu_baseline / parity: the oracle on the host copy of two targets'
	stamps, first cadences.
	"""
	from photometry_amd import simulate, psf as hpsf
	from oracle import psf as opsf, linpsf as olin
	N, FR = min(args.frames_targets, 2000), 512
	frames, tstamp, quality, cat, targets = synthetic_region(np, N, FR, T, args.seed + 11)
	prf = simulate.synthetic_prf(seed=1)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	stack = pipeline.FrameStack(ctx, frames, 0, 44)
	ctx.sync()
	out = {}
	pipeline.linpsf_frames(ctx, stack, targets, cat, tstamp, quality, model)
	t0 = time.perf_counter()
	lin = pipeline.linpsf_frames(ctx, stack, targets, cat, tstamp, quality, model)
	dt = time.perf_counter() - t0
	out['linpsf_frames'] = {'what': f'pipeline.linpsf_frames: {N} targets on a {FR} x {FR} x {T} region resident in HBM -> columnar results',
		'targets_per_s': N / dt, 'seconds': dt, 'ok_or_warning': int(np.sum((lin.status == 1) | (lin.status == 3)))}
	if args.cpu_sample > 0:
		ns, tsub = 2, min(T, 40)
		t1 = time.perf_counter()
		bad = 0
		for i in range(ns):
			b = lin[i]
			r1, r2, c1, c2 = b['stamp']
			img = frames['images'][:tsub, r1:r2, c1 - 44:c2 - 44].transpose(1, 2, 0)
			c = pipeline._catalog_of_stamp(cat, b['stamp'])
			positions = np.empty((tsub, len(c['starid']), 2))
			positions[:, :, 0] = c['row_stamp'][None, :]
			positions[:, :, 1] = c['column_stamp'][None, :]
			p = opsf.PSF(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'], b['stamp'])
			ref = olin.do_photometry(img, p, c, targets['starid'][i], positions, b['stamp'], targets['row'][i], targets['column'][i],
				np.ones((r2 - r1, c2 - c1), dtype='int32'))
			bad += not np.allclose(b['flux'][:tsub], ref['flux'], rtol=1e-7, atol=1e-8 * np.nanmax(np.abs(ref['flux'])))
		dc = time.perf_counter() - t1
		out['linpsf_frames']['cpu_baseline'] = {'value': ns / (dc * T / tsub), 'unit': 'targets/s', 'cores': 1, 'kind': 'port',
			'sample': f'{ns} targets x first {tsub} cadences (oracle restatement of LinPSFPhotometry.do_photometry on the host copy of their stamps), extrapolated to {T} cadences'}
		out['linpsf_frames']['parity_sample'] = {'targets': ns, 'cadences': tsub, 'mismatches': int(bad), 'rtol': 1e-7}
	# PSFPhotometry: a serial chain per target (tp_psf_fit), one workgroup per target, 512 - 768 of them resident: a batch below
	# that leaves workgroup slots empty for the length of the longest chain (round 5 ran 256 targets: half the chip idle)
	n2 = min(N, 2000)
	sub = {k: v[:n2] for k, v in targets.items()}
	pipeline.psf_frames(ctx, stack, {k: v[:16] for k, v in targets.items()}, cat, tstamp, quality, model)
	t0 = time.perf_counter()
	ps = pipeline.psf_frames(ctx, stack, sub, cat, tstamp, quality, model)
	dt = time.perf_counter() - t0
	out['psf_frames'] = {'what': f'pipeline.psf_frames: {n2} targets x {T} cadences of the same region (Nelder-Mead fit per star and cadence, warm-started)',
		'targets_per_s': n2 / dt, 'seconds': dt, 'finite_fraction': float(np.mean(np.isfinite(ps.flux)))}
	for a in stack.dev.values():
		a.free()
	return out
