# -*- coding: utf-8 -*-
"""
The batched counterparts of the PSF plugins over a CCD region resident in HBM (``pipeline.linpsf_frames`` /
``pipeline.psf_frames``): for every target the SAME light curve, contamination / centroids, status and messages as the
per-target plugin (``LinPSFPhotometry.do_photometry``, linpsf_photometry.py:79-219; ``PSFPhotometry.do_photometry``,
psf_photometry.py:111-196) over a ``MemoryStampSource`` of the same frames -- stamps of different sizes, neighbours inside the
5-pixel selection radius, a target whose default stamp is clipped by the frame limit, per-cadence jitter.
"""
import numpy as np
import pytest
from scipy.special import erf

pytestmark = pytest.mark.gpu


def _region(seed=5, R=90, C=84, T=18):
	rng = np.random.default_rng(seed)
	row0, col0 = 120, 260
	# (row, column, tmag): CCD coordinates
	stars = [
		(row0 + 30.3, col0 + 25.6, 11.0), (row0 + 32.4, col0 + 28.1, 12.3),     # a blend: both fitted beside each other
		(row0 + 60.7, col0 + 50.2, 9.4), (row0 + 63.9, col0 + 47.6, 10.8), (row0 + 58.1, col0 + 53.4, 12.9),   # three stars
		(row0 + 20.5, col0 + 66.8, 7.2),                                           # brighter: a larger default stamp
		(row0 + 75.2, col0 + 6.4, 10.1),                                           # at the left limit: clipped stamp
		(row0 + 45.0, col0 + 70.5, 13.5),                                          # alone and faint
	]
	rr, cc = np.arange(R) + row0, np.arange(C) + col0
	jitter = rng.normal(scale=0.03, size=(T, 2))
	images = np.empty((R, C, T), dtype='float32')
	noise = np.empty((R, C, T), dtype='float32')
	for k in range(T):
		img = np.zeros((R, C))
		for (r, c, tmag) in stars:
			flux = 10**(-0.4 * (tmag - 20.451))
			sig = 0.8
			pr = 0.5 * (erf((rr + 0.5 - (r + jitter[k, 1])) / (np.sqrt(2) * sig)) - erf((rr - 0.5 - (r + jitter[k, 1])) / (np.sqrt(2) * sig)))
			pc = 0.5 * (erf((cc + 0.5 - (c + jitter[k, 0])) / (np.sqrt(2) * sig)) - erf((cc - 0.5 - (c + jitter[k, 0])) / (np.sqrt(2) * sig)))
			img += flux * np.outer(pr, pc)
		nz = np.sqrt(np.abs(img) + 150.0)
		images[:, :, k] = img + rng.normal(size=img.shape) * nz
		noise[:, :, k] = nz
	images[rng.random(images.shape) < 3e-4] = np.nan
	frames = {'images': images, 'images_err': noise, 'backgrounds': np.full(images.shape, 100.0, dtype='float32')}
	time = 1500.0 + np.arange(T) * 1800.0 / 86400.0
	quality = np.zeros(T, dtype='int32')
	cat = {'starid': np.arange(len(stars), dtype='int64') + 501, 'tmag': np.array([s[2] for s in stars], dtype='float32'),
		'row': np.array([s[0] for s in stars], dtype='float32'), 'column': np.array([s[1] for s in stars], dtype='float32')}
	targets = {'starid': cat['starid'].copy(), 'tmag': np.array([s[2] for s in stars]), 'row': np.array([s[0] for s in stars]),
		'column': np.array([s[1] for s in stars])}
	return frames, row0, col0, time, quality, cat, targets, jitter


def _setup(T=18):
	from photometry_amd import pipeline, psf as hpsf, simulate
	from photometry_amd.device import Context
	from photometry_amd.source import MemoryStampSource
	frames, row0, col0, time, quality, cat, targets, jitter = _region(T=T)
	prf = simulate.synthetic_prf(seed=3)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	ctx = Context(0)
	stack = pipeline.FrameStack(ctx, {k: np.moveaxis(v, 2, 0) for k, v in frames.items()}, row0, col0)
	src = MemoryStampSource(frames, row0, col0, time, np.zeros(T), np.arange(T), quality, cat, targets=targets, jitter=jitter, prf=model)
	return ctx, stack, src, model, time, quality, cat, targets, jitter


def test_linpsf_frames_equals_plugin(tmp_path):
	from photometry_amd import pipeline, STATUS
	from photometry_amd.plugins import LinPSFPhotometry
	ctx, stack, src, model, time, quality, cat, targets, jitter = _setup()
	batch = pipeline.linpsf_frames(ctx, stack, targets, cat, time, quality, model, jitter=jitter)
	sizes = set()
	n_multi = 0
	for i in range(len(targets['starid'])):
		b = batch[i]
		with LinPSFPhotometry(int(targets['starid'][i]), src, str(tmp_path), ctx=ctx) as pho:
			status = pho.do_photometry()
			assert tuple(pho.stamp) == b['stamp']
			assert status.value == b['status'], (i, status, b['status'])
			assert pho._details.get('errors', []) == b['errors']
			np.testing.assert_array_equal(pho.lightcurve['flux'], b['flux'])
			np.testing.assert_array_equal(pho.lightcurve['flux_err'], b['flux_err'])
			if status != STATUS.ERROR:
				assert pho.additional_headers['PSF_CONT'][0] == b['contamination']
				n_multi += b['contamination'] > 0
			assert np.isfinite(b['flux']).sum() >= len(time) - 1
			sizes.add((pho.stamp[1] - pho.stamp[0], pho.stamp[3] - pho.stamp[2]))
	assert len(sizes) >= 3 and n_multi >= 4   # several stamp groups; blends with a fitted neighbour
	ctx.close()


def test_psf_frames_equals_plugin(tmp_path):
	from photometry_amd import pipeline
	from photometry_amd.plugins import PSFPhotometry
	ctx, stack, src, model, time, quality, cat, targets, jitter = _setup(T=4)
	batch = pipeline.psf_frames(ctx, stack, targets, cat, time, quality, model, readnoise=10, gain=100, n_readout=src.n_readout)
	n_fin = 0
	for i in range(len(targets['starid'])):
		b = batch[i]
		with PSFPhotometry(int(targets['starid'][i]), src, str(tmp_path), ctx=ctx) as pho:
			status = pho.do_photometry()
			assert tuple(pho.stamp) == b['stamp'] and status.value == b['status']
			np.testing.assert_array_equal(pho.lightcurve['flux'], b['flux'])
			np.testing.assert_array_equal(pho.lightcurve['flux_err'], b['flux_err'])
			np.testing.assert_array_equal(pho.lightcurve['pos_centroid'], b['pos_centroid'])
			n_fin += int(np.isfinite(b['flux']).sum())
	assert n_fin >= 3 * len(targets['starid'])
	ctx.close()
