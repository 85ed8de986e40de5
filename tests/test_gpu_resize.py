# -*- coding: utf-8 -*-
"""
The stamp-resize retry loop of the aperture plugin (photometry/AperturePhotometry/photometry.py:75-170,
BasePhotometry.resize_stamp / _set_stamp, BasePhotometry.py:567-693) for a batch: targets of one CCD region resident in HBM,
stamps re-cut on the device round after round (``pipeline.aperture_frames`` / ``tessphot_frames``).

Three-way parity on a region with faint stars, bright stars with bleed trails, and trails running into the frame limit:
batch == the per-target plugin (``AperturePhotometry`` over a ``MemoryStampSource`` of the same frames) == the oracle's
restatement of the whole loop (``oracle.aperture.photometry_on_frames``): statuses, final stamps, resize counts, messages,
masks bit for bit, float32 sums bit for bit.
"""
import numpy as np
import pytest
from scipy.special import erf

pytestmark = pytest.mark.gpu


def _region(seed=3, R=110, C=96, T=24):
	rng = np.random.default_rng(seed)
	row0, col0 = 200, 300
	# (row, column, tmag, trail half-length in rows): CCD coordinates
	stars = [
		(row0 + 30.3, col0 + 25.6, 11.0, 0), (row0 + 31.9, col0 + 60.2, 9.5, 0), (row0 + 70.4, col0 + 20.7, 12.5, 0),   # faint, 15x15
		(row0 + 60.2, col0 + 50.4, 6.5, 15),     # bleed trail longer than the 19x19 default stamp: one or two resizes
		(row0 + 85.7, col0 + 75.1, 6.8, 30),     # trail into the upper frame limit: "Could not resize stamp any further."
		(row0 + 12.1, col0 + 80.3, 5.5, 40),     # bright (Tmag < 6): 10 attempts, trail into the lower limit: haloswitch quick break
		(row0 + 45.5, col0 + 8.2, 10.2, 0),      # close to the left limit: default stamp clipped
		(row0 + 33.0, col0 + 27.9, 12.0, 0),     # neighbour inside the first target's stamp (skip_targets / contamination)
	]
	rr, cc = np.arange(R) + row0, np.arange(C) + col0
	img = np.zeros((R, C))
	for (r, c, tmag, trail) in stars:
		flux = 10**(-0.4 * (tmag - 20.451))
		# a narrow core for the trailed stars: star + trail stay within the 15 % of pixels K2P2 trims before estimating the sky mode
		sig = 0.6 if trail else 0.9
		pr = 0.5 * (erf((rr + 0.5 - r) / (np.sqrt(2) * sig)) - erf((rr - 0.5 - r) / (np.sqrt(2) * sig)))
		pc = 0.5 * (erf((cc + 0.5 - c) / (np.sqrt(2) * sig)) - erf((cc - 0.5 - c) / (np.sqrt(2) * sig)))
		img += flux * np.outer(pr, pc)
		if trail:
			ri, ci = int(round(r)) - row0, int(round(c)) - col0
			lo, hi = max(ri - trail, 0), min(ri + trail + 1, R)
			img[lo:hi, ci:ci + 2] += 0.02 * flux      # two pixels wide: DBSCAN core pixels need 4 neighbours
	bkg = 100.0
	cube = img[:, :, None] * (1 + 1e-3 * rng.normal(size=T))[None, None, :]
	noise = np.sqrt(np.abs(cube) + bkg + 100.0)
	# a small positive residual sky: K2P2 estimates the sky mode from the positive pixels of the sum image only
	images = (cube + 30.0 + rng.normal(size=cube.shape) * noise).astype('float32')
	images[rng.random(images.shape) < 5e-4] = np.nan
	frames = {'images': images, 'images_err': noise.astype('float32'), 'backgrounds': np.full(images.shape, bkg, dtype='float32')}
	time = 1500.0 + np.arange(T) * 1800.0 / 86400.0
	quality = np.zeros(T, dtype='int32')
	quality[5] = 32
	cat = {'starid': np.arange(len(stars), dtype='int64') + 101, 'tmag': np.array([s[2] for s in stars], dtype='float32'),
		'row': np.array([s[0] for s in stars], dtype='float32'), 'column': np.array([s[1] for s in stars], dtype='float32')}
	targets = {'starid': cat['starid'].copy(), 'tmag': np.array([s[2] for s in stars]), 'row': np.array([s[0] for s in stars]),
		'column': np.array([s[1] for s in stars])}
	return frames, row0, col0, time, quality, cat, targets


def test_batched_resize_equals_plugin_equals_oracle(tmp_path):
	from photometry_amd import pipeline, tessphot_frames, STATUS
	from photometry_amd.device import Context
	from photometry_amd.plugins import AperturePhotometry
	from photometry_amd.tessphot import run_plugin
	from photometry_amd.source import MemoryStampSource
	from oracle import aperture as oap
	frames, row0, col0, time, quality, cat, targets = _region()
	T = len(time)
	ctx = Context(0)
	stack = pipeline.FrameStack(ctx, {k: np.moveaxis(v, 2, 0) for k, v in frames.items()}, row0, col0)
	batch = tessphot_frames(ctx, stack, targets, cat, time, quality)
	src = MemoryStampSource(frames, row0, col0, time, np.zeros(T), np.arange(T), quality, cat, targets=targets)
	n_resized = n_final = 0
	seen = set()
	for i in range(len(targets['starid'])):
		b = batch[i]
		# --- oracle: the reference's loop restated on the same frames
		o = oap.photometry_on_frames(oap.FrameTarget(frames, row0, col0, quality, cat, int(targets['starid'][i]), float(targets['tmag'][i]),
			float(targets['row'][i]), float(targets['column'][i])))
		assert b.status.value == o['status'], (i, b.status, o['status'], b._details.get('errors'), o['errors'])
		assert tuple(b._details['stamp']) == tuple(o['stamp']), (i, b._details['stamp'], o['stamp'])
		assert b._details['stamp_resizes'] == o['stamp_resizes']
		assert [e for e in b._details.get('errors', []) if not e.startswith('WARNING: Could not detrend')] == o['errors'], (i, b._details.get('errors'), o['errors'])
		if 'edge_flux' in o['details']:
			np.testing.assert_allclose(b._details['edge_flux'], o['details']['edge_flux'], rtol=1e-12)
		if 'mask' in o:
			np.testing.assert_array_equal(b.final_phot_mask, o['mask'])
			np.testing.assert_array_equal(b.lightcurve['flux'], o['flux'])
			np.testing.assert_array_equal(b.lightcurve['flux_err'], o['flux_err'])
			np.testing.assert_array_equal(b.lightcurve['flux_background'], o['flux_background'])
			np.testing.assert_allclose(b.lightcurve['pos_centroid'], o['pos_centroid'], rtol=1e-12, equal_nan=True)
			assert b._details.get('skip_targets', []) == o['skip_targets']
			n_final += 1
		# --- the per-target plugin over the same frames
		p = run_plugin(AperturePhotometry, int(targets['starid'][i]), src, str(tmp_path), ctx=ctx)
		assert p.status == b.status, (i, p.status, b.status, p._details.get('errors'))
		assert tuple(p._details['stamp']) == tuple(b._details['stamp'])
		assert p._details.get('stamp_resizes', 0) == b._details['stamp_resizes']
		if b.final_phot_mask is not None and p.status != STATUS.ERROR:
			np.testing.assert_array_equal(p.final_phot_mask, b.final_phot_mask)
			np.testing.assert_array_equal(p.lightcurve['flux'], b.lightcurve['flux'])
			for key in ('mean_flux', 'variance', 'rms_hour', 'ptp', 'variability', 'edge_flux', 'mask_size'):
				assert p._details[key] == b._details[key], key
		n_resized += b._details['stamp_resizes'] > 0
		seen.update(e for e in b._details.get('errors', []))
	assert n_resized >= 3 and n_final >= 5
	assert 'WARNING: Could not resize stamp any further.' in seen
	assert 'ERROR: Stamp resize hit limit. Haloswitch quick break.' in seen or 'ERROR: Too many stamp resizes.' in seen
	ctx.close()


def test_bright_star_large_stamp_three_way():
	"""A Tmag 3.2 star: default stamp 67 x 45 (BasePhotometry.py:541-564), beyond the LDS-resident mask builder -- cutter, the
	mask builder on HBM work arrays, extraction and diagnostics of a 3 000-pixel stamp in the batched loop, the per-target plugin
	and the oracle's restatement of the loop: same statuses, stamps, masks and float32 sums."""
	from photometry_amd import pipeline, tessphot_frames, STATUS
	from photometry_amd.device import Context
	from photometry_amd.plugins import AperturePhotometry
	from photometry_amd.tessphot import run_plugin
	from photometry_amd.source import MemoryStampSource
	from oracle import aperture as oap
	rng = np.random.default_rng(11)
	R, C, T = 150, 120, 12
	row0, col0 = 400, 500
	stars = [(row0 + 75.4, col0 + 60.7, 3.2), (row0 + 20.2, col0 + 20.9, 10.5), (row0 + 120.6, col0 + 95.1, 9.0), (row0 + 70.8, col0 + 75.3, 8.5)]
	rr, cc = np.arange(R) + row0, np.arange(C) + col0
	img = np.zeros((R, C))
	for (r, c, tmag) in stars:
		flux = 10**(-0.4 * (tmag - 20.451))
		sig = 1.4 if tmag < 5 else 0.9
		pr = 0.5 * (erf((rr + 0.5 - r) / (np.sqrt(2) * sig)) - erf((rr - 0.5 - r) / (np.sqrt(2) * sig)))
		pc = 0.5 * (erf((cc + 0.5 - c) / (np.sqrt(2) * sig)) - erf((cc - 0.5 - c) / (np.sqrt(2) * sig)))
		img += flux * np.outer(pr, pc)
	cube = img[:, :, None] * (1 + 1e-3 * rng.normal(size=T))[None, None, :]
	noise = np.sqrt(np.abs(cube) + 200.0)
	images = (cube + 30.0 + rng.normal(size=cube.shape) * noise).astype('float32')
	frames = {'images': images, 'images_err': noise.astype('float32'), 'backgrounds': np.full(images.shape, 100.0, dtype='float32')}
	time = 1500.0 + np.arange(T) * 1800.0 / 86400.0
	quality = np.zeros(T, dtype='int32')
	cat = {'starid': np.arange(len(stars), dtype='int64') + 201, 'tmag': np.array([s[2] for s in stars], dtype='float32'),
		'row': np.array([s[0] for s in stars], dtype='float32'), 'column': np.array([s[1] for s in stars], dtype='float32')}
	targets = {k: np.array(v[:2], dtype=v.dtype if k == 'starid' else 'float64') for k, v in cat.items()}
	targets['row'] = np.array([s[0] for s in stars[:2]])
	targets['column'] = np.array([s[1] for s in stars[:2]])
	targets['tmag'] = np.array([s[2] for s in stars[:2]])
	ctx = Context(0)
	stack = pipeline.FrameStack(ctx, {k: np.moveaxis(v, 2, 0) for k, v in frames.items()}, row0, col0)
	batch = tessphot_frames(ctx, stack, targets, cat, time, quality)
	src = MemoryStampSource(frames, row0, col0, time, np.zeros(T), np.arange(T), quality, cat, targets=targets)
	import tempfile
	with tempfile.TemporaryDirectory() as tmp:
		for i in range(2):
			b = batch[i]
			o = oap.photometry_on_frames(oap.FrameTarget(frames, row0, col0, quality, cat, int(targets['starid'][i]), float(targets['tmag'][i]),
				float(targets['row'][i]), float(targets['column'][i])))
			assert b.status.value == o['status'], (i, b.status, o['status'], b._details.get('errors'), o['errors'])
			assert tuple(b._details['stamp']) == tuple(o['stamp'])
			assert b._details['stamp_resizes'] == o['stamp_resizes']
			if i == 0:
				st = b._details['stamp']
				assert (st[1] - st[0]) * (st[3] - st[2]) > 54 * 54 # really the HBM path
			if 'mask' in o:
				np.testing.assert_array_equal(b.final_phot_mask, o['mask'])
				np.testing.assert_array_equal(b.lightcurve['flux'], o['flux'])
				np.testing.assert_array_equal(b.lightcurve['flux_err'], o['flux_err'])
				np.testing.assert_array_equal(b.lightcurve['flux_background'], o['flux_background'])
				assert b._details.get('skip_targets', []) == o['skip_targets']
			p = run_plugin(AperturePhotometry, int(targets['starid'][i]), src, tmp, ctx=ctx)
			assert p.status == b.status, (i, p.status, b.status, p._details.get('errors'))
			assert tuple(p._details['stamp']) == tuple(b._details['stamp'])
			if b.final_phot_mask is not None and p.status != STATUS.ERROR:
				np.testing.assert_array_equal(p.final_phot_mask, b.final_phot_mask)
				np.testing.assert_array_equal(p.lightcurve['flux'], b.lightcurve['flux'])
				for key in ('mean_flux', 'variance', 'rms_hour', 'ptp', 'edge_flux', 'mask_size'):
					assert p._details[key] == b._details[key], key
	assert batch[0].status in (STATUS.OK, STATUS.WARNING)
	ctx.close()


@pytest.mark.parametrize("time_major,budget_gb", [('1', '0.00004'), ('0', '0.004')])
def test_aperture_frames_in_parts_under_a_memory_budget(monkeypatch, time_major, budget_gb):
	"""A budget of device memory far below the batch cuts the groups of a round into parts (and a group into chunks of targets)
	that are processed one after the other: same results as in one piece.  With the time-major stacks (the default) a target costs
	its output block only (40 KB of budget: eight targets per chunk); with the stamp cubes cut per pass 4 MB are a dozen targets."""
	monkeypatch.setenv('TESSPHOT_FRAMES_TIME_MAJOR', time_major)
	from photometry_amd import pipeline, tessphot_frames
	from photometry_amd.device import Context
	rng = np.random.default_rng(12)
	N, FR, T = 150, 128, 40
	rows, cols, tmag = rng.uniform(12, FR - 12, N), rng.uniform(12, FR - 12, N), rng.uniform(8.0, 13.5, N)
	img = np.zeros((FR, FR))
	yy, xx = np.mgrid[-4:5, -4:5]
	for r, c, m in zip(rows, cols, tmag):
		ri, ci = int(round(r)), int(round(c))
		img[ri - 4:ri + 5, ci - 4:ci + 5] += 10**(-0.4 * (m - 20.451)) * np.exp(-0.5 * ((yy + ri - r)**2 + (xx + ci - c)**2) / 0.81) / (2 * np.pi * 0.81)
	base = (img[None] * (1 + 1e-3 * rng.normal(size=T))[:, None, None]).astype('float32')
	noise = np.sqrt(np.abs(base) + 200.0).astype('float32')
	frames = {'images': (base + 30.0 + rng.standard_normal(base.shape).astype('float32') * noise).astype('float32'), 'images_err': noise,
		'backgrounds': np.full((T, FR, FR), 100.0, dtype='float32')}
	tstamp = 1500.0 + np.arange(T) * 1800.0 / 86400.0
	quality = np.zeros(T, dtype='int32')
	cat = {'starid': np.arange(N, dtype='int64') + 1, 'tmag': tmag.astype('float32'), 'row': rows.astype('float32'), 'column': (cols + 44).astype('float32')}
	targets = {'starid': cat['starid'].copy(), 'tmag': tmag, 'row': rows, 'column': cols + 44}
	ctx = Context(0)
	stack = pipeline.FrameStack(ctx, frames, 0, 44)
	whole = tessphot_frames(ctx, stack, targets, cat, tstamp, quality)
	monkeypatch.setenv('TESSPHOT_FRAMES_BUDGET_GB', budget_gb)
	parts = tessphot_frames(ctx, stack, targets, cat, tstamp, quality)
	assert len(parts.frames.groups) > len(whole.frames.groups) + 3
	np.testing.assert_array_equal(parts.status, whole.status)
	np.testing.assert_array_equal(parts.stamp, whole.stamp)
	for name in ('mask_size', 'contamination', 'mean_flux', 'variance'):
		np.testing.assert_array_equal(parts.column(name), whole.column(name))
	for i in (0, 7, 77, N - 1):
		a, b = whole[i], parts[i]
		assert a.status == b.status
		if a.lightcurve is not None:
			np.testing.assert_array_equal(a.lightcurve['flux'], b.lightcurve['flux'])
			np.testing.assert_array_equal(a.final_phot_mask, b.final_phot_mask)
	ctx.close()


def test_pipelined_batches_equal_separate_calls():
	"""Consecutive batches of one region with two (and three) of them on the device at a time: every batch gives exactly what a
	call of its own gives -- statuses, stamps, resize counts, messages, masks, light curves -- in order."""
	from photometry_amd import pipeline, tessphot_frames, tessphot_frames_pipelined
	from photometry_amd.device import Context
	frames, row0, col0, time, quality, cat, targets = _region()
	rng = np.random.default_rng(5)
	order = rng.permutation(len(targets['starid']))
	cuts = [order[:3], order[3:4], order[4:], order[:0], order[::2]]   # ragged batches, an empty one, a repeated target
	batches = [{k: np.asarray(v)[sel] for k, v in targets.items()} for sel in cuts]
	ctx = Context(0)
	stack = pipeline.FrameStack(ctx, {k: np.moveaxis(v, 2, 0) for k, v in frames.items()}, row0, col0)
	alone = [tessphot_frames(ctx, stack, b, cat, time, quality) for b in batches]
	for in_flight in (2, 3):
		together = list(tessphot_frames_pipelined(ctx, stack, iter(batches), cat, time, quality, in_flight=in_flight))
		assert len(together) == len(batches)
		for a, b in zip(alone, together):
			assert len(a) == len(b)
			np.testing.assert_array_equal(a.status, b.status)
			np.testing.assert_array_equal(a.stamp, b.stamp)
			np.testing.assert_array_equal(a.frames.stamp_resizes, b.frames.stamp_resizes)
			for i in range(len(a)):
				x, y = a[i], b[i]
				assert x.status == y.status and x._details.get("errors") == y._details.get("errors") and x.starid == y.starid
				if x.lightcurve is not None:
					for key in ('flux', 'flux_err', 'flux_background', 'pos_centroid'):
						np.testing.assert_array_equal(x.lightcurve[key], y.lightcurve[key])
					np.testing.assert_array_equal(x.final_phot_mask, y.final_phot_mask)
	ctx.close()


def _compare_frames_results(a, b, same_groups=True):
	"""Two pipeline.FramesResult of the same batch: everything a caller can see, target by target."""
	assert a.n == b.n
	np.testing.assert_array_equal(a.status, b.status)
	np.testing.assert_array_equal(a.stamp, b.stamp)
	np.testing.assert_array_equal(a.stamp_resizes, b.stamp_resizes)
	np.testing.assert_array_equal(a.has_result, b.has_result)
	assert a.errors == b.errors
	assert a.edge_flux == b.edge_flux                       # (float64 sums in numpy's pairwise order on both sides: equal, not close)
	assert len(a.groups) == len(b.groups) or not same_groups
	for i in range(a.n):
		x, y = a[i], b[i]
		assert set(x) == set(y), (i, set(x) ^ set(y))
		for k in x:
			if isinstance(x[k], np.ndarray):
				np.testing.assert_array_equal(x[k], y[k], err_msg=f'target {i}: {k}')
			elif k == 'diagnostics':
				np.testing.assert_array_equal(np.array(list(x[k].values())), np.array(list(y[k].values())), err_msg=f'target {i}: diagnostics')
			elif isinstance(x[k], float):
				assert x[k] == y[k] or (x[k] != x[k] and y[k] != y[k]), (i, k, x[k], y[k])
			else:
				assert x[k] == y[k], (i, k, x[k], y[k])


def test_native_engine_equals_the_python_rounds():
	"""The native job engine (csrc/frames.cpp: rounds driven by a worker thread of the library) against the Python generator of the
	same rounds, on the region with bleed trails / frame limits / the haloswitch quick break, on a crowded region whose targets
	resize in several size groups, with invalid stamps and an empty batch, and under a memory budget that cuts the rounds in parts."""
	import os
	from photometry_amd import pipeline
	from photometry_amd.device import Context
	ctx = Context(0)
	frames, row0, col0, time, quality, cat, targets = _region()
	stack = pipeline.FrameStack(ctx, {k: np.moveaxis(v, 2, 0) for k, v in frames.items()}, row0, col0)
	# a target whose default stamp lies outside the region ("Invalid stamp selected") among the others
	t2 = {k: np.concatenate((np.asarray(v), np.asarray(v)[:1])) for k, v in targets.items()}
	t2['row'][-1] = row0 - 500.0
	for tg in (targets, t2, {k: np.asarray(v)[:0] for k, v in targets.items()}):
		py = pipeline.aperture_frames(ctx, stack, tg, cat, time, quality, engine='python')
		nat = pipeline.aperture_frames(ctx, stack, tg, cat, time, quality, engine='native')
		_compare_frames_results(py, nat)
	assert int(py.n) == 0
	# crowded region
	rng = np.random.default_rng(44)
	N, FR, T = 700, 192, 30
	rows, cols = rng.uniform(10, FR - 10, N), rng.uniform(10, FR - 10, N)
	tmag = np.where(rng.random(N) < 0.06, rng.uniform(5.0, 7.5, N), rng.uniform(8.0, 14.0, N))
	img = np.zeros((FR + 16, FR + 16))
	yy, xx = np.mgrid[-8:9, -8:9]
	for r, c, m in zip(rows, cols, tmag):
		ri, ci = int(round(r)) + 8, int(round(c)) + 8
		img[ri - 8:ri + 9, ci - 8:ci + 9] += 10**(-0.4 * (m - 20.451)) * np.exp(-0.5 * ((yy + ri - 8 - r)**2 + (xx + ci - 8 - c)**2) / 0.81) / (2 * np.pi * 0.81)
	img = img[8:-8, 8:-8]
	base = (img[None] * (1 + 1e-3 * rng.normal(size=T))[:, None, None]).astype('float32')
	noise = np.sqrt(np.abs(base) + 200.0).astype('float32')
	fr = {'images': (base + 30.0 + rng.standard_normal(base.shape).astype('float32') * noise).astype('float32'), 'images_err': noise,
		'backgrounds': np.full((T, FR, FR), 100.0, dtype='float32')}
	tstamp = 1500.0 + np.arange(T) * 1800.0 / 86400.0
	q = np.zeros(T, dtype='int32')
	cat2 = {'starid': np.arange(N, dtype='int64') + 1, 'tmag': tmag.astype('float32'), 'row': rows.astype('float32'), 'column': (cols + 44).astype('float32')}
	tg2 = {'starid': cat2['starid'].copy(), 'tmag': tmag, 'row': rows, 'column': cols + 44}
	stack2 = pipeline.FrameStack(ctx, fr, 0, 44)
	py = pipeline.aperture_frames(ctx, stack2, tg2, cat2, tstamp, q, engine='python')
	nat = pipeline.aperture_frames(ctx, stack2, tg2, cat2, tstamp, q, engine='native')
	_compare_frames_results(py, nat)
	assert int((nat.stamp_resizes > 0).sum()) >= 10 and len(nat.groups) >= 4 and len(nat.errors) >= 1
	print('crowded region:', int((nat.stamp_resizes > 0).sum()), 'targets resized,', len(nat.groups), 'device passes,', len(nat.errors), 'targets with messages')
	try:
		# (the native engine reads the time-major stacks and cuts no cubes: a target costs it its output block only)
		os.environ['TESSPHOT_FRAMES_BUDGET_GB'] = '0.00003'
		parts = pipeline.aperture_frames(ctx, stack2, tg2, cat2, tstamp, q, engine='native')
		os.environ['TESSPHOT_FRAMES_BUDGET_GB'] = '0.003'
		parts_py = pipeline.aperture_frames(ctx, stack2, tg2, cat2, tstamp, q, engine='python')
	finally:
		del os.environ['TESSPHOT_FRAMES_BUDGET_GB']
	assert len(parts.groups) > len(nat.groups) + 3
	_compare_frames_results(parts_py, parts, same_groups=False)   # (cut into parts by different budgets)
	# the group arrays are read-only views that go with the result
	g0 = nat.groups[0]
	with pytest.raises(ValueError):
		g0['status'][0] = 7
	nat.release()
	assert nat.groups == []
	# pipelined, more batches than slots, results held by the caller while later batches run
	batches = [{k: np.asarray(v)[rng.permutation(N)[:200]] for k, v in tg2.items()} for _ in range(7)]
	alone = [pipeline.aperture_frames(ctx, stack2, b, cat2, tstamp, q, engine='python') for b in batches]
	held = list(pipeline.aperture_frames_pipelined(ctx, stack2, iter(batches), cat2, tstamp, q, in_flight=3))
	for a, b in zip(alone, held):
		_compare_frames_results(a, b)
	ctx.close()


@pytest.mark.parametrize("engine_kind", ['native', 'python'])
def test_sum_images_are_crops_of_the_regions_sum_image(engine_kind):
	"""BasePhotometry.sumimage, FFI branch (BasePhotometry.py:1001-1006): ``self._sumimage_full[ir1:ir2, ic1:ic2]`` -- the sum image a
	target works with is a crop of the one prepare.py accumulated for the whole frame (prepare.py:450-453, 459: float64 sums of the
	finite float32 pixels of the good frames, in cadence order, over their count).  Bit for bit, for the final (resized) stamps,
	whether the stack computes the region's sum image itself or is handed the file's."""
	from photometry_amd import pipeline
	from photometry_amd.device import Context
	frames, row0, col0, time, quality, cat, targets = _region()
	fr = {k: np.moveaxis(v, 2, 0) for k, v in frames.items()}
	# prepare.py:450-453, 459 on the host
	full = np.zeros(fr['images'].shape[1:], dtype='float64')
	nimg = np.zeros(full.shape, dtype='int64')
	for k in range(len(time)):
		if quality[k] & 4335 == 0:      # TESSQualityFlags.filter with the default bitmask (quality.py)
			f = fr['images'][k].astype('float64')
			ok = np.isfinite(f)
			nimg += ok
			full += np.where(ok, f, 0.0)
	with np.errstate(invalid='ignore'):
		full = full / nimg
	ctx = Context(0)
	for given in (False, True):
		stack = pipeline.FrameStack(ctx, fr, row0, col0, sumimage=full if given else None)
		np.testing.assert_array_equal(stack.sumimage_for(quality).to_host(), full)
		res = pipeline.aperture_frames(ctx, stack, targets, cat, time, quality, engine=engine_kind)
		n = 0
		for i in range(len(targets['starid'])):
			b = res[i]
			if 'sumimage' not in b:
				continue
			r1, r2, c1, c2 = b['stamp']
			np.testing.assert_array_equal(b['sumimage'], full[r1 - row0:r2 - row0, c1 - col0:c2 - col0])
			n += 1
		assert n >= 5
	# another quality series: the stack forms the sum image anew
	q2 = quality.copy(); q2[3] = 1
	stack = pipeline.FrameStack(ctx, fr, row0, col0)
	a = stack.sumimage_for(quality).to_host()
	b = stack.sumimage_for(q2).to_host()
	assert np.any(a != b)
	np.testing.assert_array_equal(stack.sumimage_for(quality).to_host(), full)
	ctx.close()


def test_crop_sumimage_outside_the_frame_is_nan():
	from photometry_amd import engine
	from photometry_amd.device import Context
	ctx = Context(0)
	rng = np.random.default_rng(1)
	full = rng.normal(size=(40, 37))
	stamps = np.array([[95, 106, 200, 212], [130, 141, 230, 242], [100, 111, 195, 207], [120, 131, 210, 222]], dtype='int32')   # (row0, col0) = (100, 200)
	out = engine.crop_sumimage(ctx, ctx.array(full), ctx.array(stamps), 11, 12, 100, 200).to_host().reshape(4, 11, 12)
	for t, (r1, r2, c1, c2) in enumerate(stamps):
		ref = np.full((11, 12), np.nan)
		for i in range(11):
			for j in range(12):
				r, c = r1 - 100 + i, c1 - 200 + j
				if 0 <= r < 40 and 0 <= c < 37:
					ref[i, j] = full[r, c]
		np.testing.assert_array_equal(out[t], ref)
	ctx.close()


def test_postage_stamp_datasource_sums_its_own_stamps():
	"""``datasource='tpf:...'``: the reference sums the stamp's own cube (BasePhotometry.py:1007-1019) instead of cropping the
	region's sum image (:1001-1006), and the haloswitch quick break does not apply (photometry.py:146).  The native engine then
	gets no region sum image (``tp_frames_stack.d_sumimage = NULL``: every pass cuts the images and runs ``tp_sumimage``) -- equal
	to the Python rounds target for target, and its sum images equal to the oracle's sum over the cut stamp."""
	from photometry_amd import pipeline
	from photometry_amd.device import Context
	from oracle import sumimage as osum
	ctx = Context(0)
	frames, row0, col0, time, quality, cat, targets = _region()
	stack = pipeline.FrameStack(ctx, {k: np.moveaxis(v, 2, 0) for k, v in frames.items()}, row0, col0)
	py = pipeline.aperture_frames(ctx, stack, targets, cat, time, quality, engine='python', datasource='tpf:1234')
	nat = pipeline.aperture_frames(ctx, stack, targets, cat, time, quality, engine='native', datasource='tpf:1234')
	_compare_frames_results(py, nat)
	ffi = pipeline.aperture_frames(ctx, stack, targets, cat, time, quality, engine='native')
	n_checked = 0
	for i in range(nat.n):
		r = nat[i]
		if 'sumimage' not in r:
			continue
		r1, r2, c1, c2 = (int(v) for v in r['stamp'])
		cube = frames['images'][r1 - row0:r2 - row0, c1 - col0:c2 - col0, :]
		np.testing.assert_allclose(r['sumimage'], osum.sumimage(cube, quality), rtol=1e-12, equal_nan=True)
		n_checked += 1
	assert n_checked >= 3
	# no quick break for a postage stamp: the bright target keeps resizing where the FFI target stops
	assert (nat.stamp_resizes >= ffi.stamp_resizes).all()
	ctx.close()


def test_time_major_stacks_change_nothing():
	"""The native engine on the time-major stacks (no cut: tp_aperture_extract_stack reads a mask pixel's series as a row of the
	transposed stack) against the same engine cutting the in-mask rows per pass (TESSPHOT_FRAMES_TIME_MAJOR=0): every array equal."""
	import os
	from photometry_amd import pipeline
	from photometry_amd.device import Context
	ctx = Context(0)
	frames, row0, col0, time, quality, cat, targets = _region()
	fr = {k: np.moveaxis(v, 2, 0) for k, v in frames.items()}
	stack_a = pipeline.FrameStack(ctx, fr, row0, col0)
	a = pipeline.aperture_frames(ctx, stack_a, targets, cat, time, quality, engine='native')
	assert stack_a._time_major not in (None, False)
	os.environ['TESSPHOT_FRAMES_TIME_MAJOR'] = '0'
	try:
		stack_b = pipeline.FrameStack(ctx, fr, row0, col0)
		b = pipeline.aperture_frames(ctx, stack_b, targets, cat, time, quality, engine='native')
		assert stack_b._time_major is None
	finally:
		del os.environ['TESSPHOT_FRAMES_TIME_MAJOR']
	_compare_frames_results(a, b)
	assert int(((a.status == 1) | (a.status == 3)).sum()) >= 3
	ctx.close()


def test_frames_transpose_and_extract_stack_equal_the_cut_cubes():
	"""tp_frames_transpose is the transposition (padding zero), and tp_aperture_extract_stack on the transposed stacks gives, bit for
	bit, what tp_aperture_extract gives on the cubes cut from the frames -- masks below and above 128 pixels (both kernels), stamps
	of two sizes, a stack with an odd number of cadences."""
	import ctypes
	from photometry_amd import engine
	from photometry_amd.device import Context
	ctx = Context(0)
	rng = np.random.default_rng(5)
	T, R, C = 37, 70, 90
	t_pitch = 64
	fr = [rng.normal(100.0, 10.0, (T, R, C)).astype('float32') for _ in range(3)]
	fr[0][3, 10:14, 20] = np.nan
	d = [ctx.array(x) for x in fr]
	dt = [ctx.empty((R * C, t_pitch), 'float32') for _ in range(3)]
	for x, y in zip(d, dt):
		ctx._check(ctx.lib.tp_frames_transpose(ctx.handle, x.ptr, T, R * C, R * C, y.ptr, t_pitch))
	ctx.sync()
	for x, y in zip(fr, dt):
		got = y.to_host()
		np.testing.assert_array_equal(got[:, :T], x.reshape(T, R * C).T)
		assert not got[:, T:].any()
	row0, col0 = 100, 44
	for (H, W, n) in [(15, 15, 40), (24, 31, 9)]:
		r0 = rng.integers(0, R - H + 1, n); c0 = rng.integers(0, C - W + 1, n)
		stamps = np.stack([row0 + r0, row0 + r0 + H, col0 + c0, col0 + c0 + W], axis=1).astype('int32')
		mask = (rng.random((n, H, W)) < 0.25).astype('uint8')
		mask[0] = 0
		mask[1] = 1                                   # every pixel: above 128 -> the recursive pairwise kernel
		status = np.ones(n, dtype='int32'); status[2] = 2   # STATUS.ERROR: the target is skipped
		ds, dm, dst = ctx.array(stamps), ctx.array(mask), ctx.array(status)
		cubes = engine.cut_stamps_multi(ctx, d, ds, H, W, row0, col0)
		outs = {}
		for kind in ('cube', 'stack'):
			lc = [ctx.empty((n, T), 'float64') for _ in range(5)]
			for a in lc:
				ctx._check(ctx.lib.tp_memset(ctx.handle, a.ptr, 0x7f, a.nbytes))
			if kind == 'cube':
				desc = cubes[0].desc
				ctx._check(ctx.lib.tp_aperture_extract(ctx.handle, ctypes.byref(desc), cubes[0].ptr, cubes[1].ptr, cubes[2].ptr, 0, 0, None, 0,
					dm.ptr, ds.ptr, dst.ptr, lc[0].ptr, lc[1].ptr, lc[2].ptr, lc[3].ptr, lc[4].ptr, T))
			else:
				ctx._check(ctx.lib.tp_aperture_extract_stack(ctx.handle, n, T, H, W, dt[0].ptr, dt[1].ptr, dt[2].ptr, t_pitch, R, C, row0, col0,
					dm.ptr, ds.ptr, dst.ptr, lc[0].ptr, lc[1].ptr, lc[2].ptr, lc[3].ptr, lc[4].ptr, T))
			ctx.sync()
			outs[kind] = [a.to_host() for a in lc]
		for a, b in zip(outs['cube'], outs['stack']):
			np.testing.assert_array_equal(a.view('uint64'), b.view('uint64'))
		assert np.isfinite(outs['stack'][0][1]).all() and np.isnan(outs['stack'][0][0]).all()
	ctx.close()
