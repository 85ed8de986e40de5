# -*- coding: utf-8 -*-
"""
BASELINE.json configurations at their stated shapes, through the C ABI.

* configs[1]: synthetic 1 000 targets x 200 cadences x 11x11, seed 0, APERTURE-ONLY (images + errors, no background
  cube: ``d_backgrounds = NULL``) -- every target against the oracle, bit for bit.
* configs[4]: one rank's share (12 500 targets, aperture + PSF into the packed gathered block), by size-independent properties.
* configs[3]: the 10 000 x 1 300 x 15x15 cube through the LinPSF path, by size-independent properties
  (chunked == whole, second run == first, a seeded sample == the oracle).
configs[2] has its full-size test in tests/test_gpu_fullsize.py; configs[0] (bundled TIC 182092046) cannot run: the
reference's test data are git-LFS pointers.
"""
import hashlib
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config1_aperture_only_every_target():
	from photometry_amd import simulate, pipeline
	from photometry_amd.device import Context
	from oracle import aperture as oap
	from k2p2_common import own_chain_check
	ctx = Context(0)
	Nt, T, H, W = 1000, 200, 11, 11
	s = simulate.make_scene(Nt, T, H, W, seed=0)           # seed 0 for this configuration
	simulate.fill_cubes(s)
	res = pipeline.run_aperture(ctx, s, cubes='host_aperture_only')
	assert np.all(np.isnan(res['flux_background']))         # no background cube -> nothing to sum
	n_ok = n_err = 0
	for i in range(Nt):
		# the oracle runs on the DEVICE sum image so that last-bit differences of A1 (checked to 1e-12 elsewhere) do not enter
		try:
			ref = oap.do_photometry(res['sumimage'][i], s.images[i], s.images_err[i], None, tuple(s.stamps[i]),
				s.target_pos_row[i], s.target_pos_column[i], s.target_tmag[i], s.target_starid[i], s.catalog_of(i), s.aperture[i])
		except Exception as e: # noqa: B902 -- any exception in the plugin is STATUS.ERROR upstream (tessphot.py:37-49)
			ref = {'status': oap.STATUS_ERROR, 'exception': repr(e)}
		assert int(res['status'][i]) == ref['status'], (i, res['status'][i], ref)
		if ref['status'] == oap.STATUS_ERROR or 'mask' not in ref:
			n_err += 1
			continue
		np.testing.assert_array_equal(res['mask'][i].astype(bool), ref['mask'], err_msg=f"mask {i}")
		np.testing.assert_array_equal(res['flux'][i], ref['flux'], err_msg=f"flux {i}")
		np.testing.assert_array_equal(res['flux_err'][i], ref['flux_err'], err_msg=f"flux_err {i}")
		np.testing.assert_allclose(res['pos_centroid'][i], ref['pos_centroid'], rtol=1e-12, equal_nan=True)
		c = ref['contamination']
		assert (np.isnan(c) and np.isnan(res['contamination'][i])) or abs(res['contamination'][i] - c) < 2e-6
		n_ok += 1
	print(f"configs[1]: {n_ok} targets bit-exact, {n_err} ERROR targets agree")
	assert n_ok >= 0.9 * Nt
	# the chain closed for EVERY target: the oracle's own sum image -> the oracle's own mask and light curve against the device's
	verdicts = [own_chain_check(res['sumimage'][i], s.images[i], s.images_err[i], None, s.quality, tuple(s.stamps[i]), s.target_pos_row[i],
		s.target_pos_column[i], s.target_tmag[i], s.target_starid[i], s.catalog_of(i), s.aperture[i], res['mask'][i], res['status'][i],
		res['flux'][i], res['flux_err'][i]) for i in range(Nt)]
	print('configs[1], own-sum-image chain:', verdicts.count('exact'), 'exact,', verdicts.count('razor'), 'razor of', Nt)
	assert verdicts.count('razor') <= 2
	ctx.close()


def _digest(res):
	return {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in res.items()}


def test_config3_linpsf_full_size_properties():
	from photometry_amd import simulate, engine, pipeline, psf as hpsf
	from photometry_amd.device import Context
	from oracle import psf as opsf, linpsf as olin
	ctx = Context(0)
	if ctx.info()['hbm_bytes'] < 60e9:
		pytest.skip("needs the 288 GB device")
	Nt, T, H, W = 10000, 1300, 15, 15
	scene = simulate.make_scene(Nt, T, H, W, seed=1)        # seed 1 for C3 / C4
	scene.aperture = None
	cubes = engine.synth_fill(ctx, scene, images_err=False, backgrounds=False)
	prf = opsf.synthetic_prf(seed=7)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	batch = pipeline.LinPSFBatch(ctx, scene, model, images=cubes['images'])
	pipeline.linpsf_step(ctx, batch)
	ctx.sync()
	a = batch.out.to_host()
	da = _digest(a)
	assert np.isin(a['status'], (1, 3)).mean() > 0.95 and np.isfinite(a['flux']).mean() > 0.99

	# a second run reproduces the first bit for bit
	pipeline.linpsf_step(ctx, batch)
	ctx.sync()
	assert _digest(batch.out.to_host()) == da

	# two unequal chunks of targets (views of the same buffers) == one launch
	out2 = engine.LinPSFResult(ctx, Nt, batch.n_fit_stars, T)
	for start, count in ((0, 3333), (3333, Nt - 3333)):
		view = engine.LinPSFResult.__new__(engine.LinPSFResult)
		view.n_cad = out2.n_cad
		for k in ('flux', 'flux_err', 'contamination', 'status'):
			setattr(view, k, getattr(out2, k).slice0(start, count))
		view.fluxes_all, view.fluxes_mean = out2.fluxes_all, out2.fluxes_mean   # indexed by the absolute star offsets
		engine.linpsf_fit(ctx, batch.images.slice0(start, count), batch.coef.slice0(start, count), batch.tx, batch.ty,
			batch.star_offsets.slice0(start, count + 1), batch.target_index.slice0(start, count), batch.pos_row, batch.pos_col,
			batch.max_stars, out=view)
	ctx.sync()
	assert _digest(out2.to_host()) == da

	# a seeded sample spread over the batch against the oracle (about 1 s of CPU per target)
	rng = np.random.default_rng(11)
	for i in np.sort(rng.choice(Nt, 5, replace=False)):
		i = int(i)
		img = cubes['images'].slice0(i, 1).to_host()[0]
		cat = scene.catalog_of(i)
		positions = np.empty((T, len(cat['starid']), 2))
		positions[:, :, 0] = cat['row_stamp'][None, :] + scene.jitter[:, 1][:, None]
		positions[:, :, 1] = cat['column_stamp'][None, :] + scene.jitter[:, 0][:, None]
		p = opsf.PSF(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'], tuple(scene.stamps[i]))
		ref = olin.do_photometry(img, p, cat, scene.target_starid[i], positions, tuple(scene.stamps[i]),
			scene.target_pos_row[i], scene.target_pos_column[i], np.ones((H, W), dtype='int32'))
		scale = np.nanmax(np.abs(ref['flux']))
		np.testing.assert_allclose(a['flux'][i], ref['flux'], rtol=1e-8, atol=1e-9*scale)   # north_star: 1e-5 relative
		assert int(a['status'][i]) == ref['status']
		np.testing.assert_allclose(a['contamination'][i], ref['contamination'], rtol=1e-7, atol=1e-11)
	ctx.close()


def test_config4_one_rank_share_aperture_plus_psf():
	"""
	BASELINE configs[4] is 100 000 targets over 8 GPUs: what ONE rank does per step -- 12 500 targets x 1 300 x 15x15, stamp
	background + aperture photometry + the LinPSF fit, every per-target result in the packed block that is gathered -- at its
	full per-GPU size, by size-independent properties: a second step reproduces the block bit for bit; the block unpacks
	(``comm.unpack_block``, the layout the gloo test gathers) into the arrays the work object holds; a seeded sample of targets
	equals the oracle from the raw cube alone (aperture: bit for bit, both background series included; LinPSF: 1e-8).  The cross-rank part (sharding, the
	gather of the blocks, reassembly in global order) is covered on CPU by tests/test_distributed_gloo.py.
	"""
	from photometry_amd import simulate, engine, pipeline, psf as hpsf, comm as tpcomm
	from photometry_amd.device import Context
	from oracle import psf as opsf, linpsf as olin, sumimage as osum, aperture as oap, backgrounds as ob
	ctx = Context(0)
	if ctx.info()['hbm_bytes'] < 100e9:
		pytest.skip("needs the 288 GB device")
	Nt, T, H, W = 12500, 1300, 15, 15
	scene = simulate.make_scene(Nt, T, H, W, seed=1003)      # rank 3 of the bench's seeding (seed * 1000 + rank)
	scene.aperture = None
	cubes = engine.synth_fill(ctx, scene, images=False, images_err=True, backgrounds=False, raw=True)
	batch = pipeline.ApertureBatch(ctx, scene, cubes={'raw': cubes['raw'], 'raw_err': cubes['images_err']})
	work = pipeline.ApertureWork(ctx, batch, packed=True, psf=True)
	prf = simulate.synthetic_prf(seed=1)
	model = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	lin = pipeline.LinPSFBatch(ctx, scene, model, images=cubes['raw'], subtract=work.bkg, work=work)

	def step():
		pipeline.aperture_step(ctx, batch, work)
		pipeline.linpsf_step(ctx, lin)
		ctx.sync()
		return work.block.to_host()

	block = step()
	assert hashlib.sha256(step().tobytes()).hexdigest() == hashlib.sha256(block.tobytes()).hexdigest()
	got = tpcomm.unpack_block(block, work.block_layout)
	assert set(got) == {'lc', 'contamination', 'status', 'flags', 'mask', 'psf_flux', 'psf_contamination', 'psf_status'}
	np.testing.assert_array_equal(got['psf_flux'], lin.out.flux.to_host())
	np.testing.assert_array_equal(got['status'], work.status.to_host())
	assert np.isin(got['status'], (1, 3)).mean() > 0.9 and np.isin(got['psf_status'], (1, 3)).mean() > 0.95
	assert np.isfinite(got['psf_flux']).mean() > 0.99

	bkg = work.bkg.to_host()[:, :T]
	rng = np.random.default_rng(4)
	for i in np.sort(rng.choice(Nt, 4, replace=False)):
		i = int(i)
		raw = cubes['raw'].slice0(i, 1).to_host()[0]
		err = cubes['images_err'].slice0(i, 1).to_host()[0]
		braw = ob.background_series(raw)                         # the oracle's own B* and B2: nothing of the device on its side
		np.testing.assert_array_equal(work.bkg_raw.slice0(i, 1).to_host()[0, :T], braw)
		bsm = ob.smooth_time(braw, 3)
		np.testing.assert_array_equal(bkg[i], bsm)
		series = bsm[None, None, :]
		img, e2 = ob.subtract_background(raw, err, series)
		S = osum.sumimage(img, scene.quality)
		ref = oap.do_photometry(S, img, e2, np.broadcast_to(series.astype('float32'), img.shape), tuple(scene.stamps[i]), scene.target_pos_row[i],
			scene.target_pos_column[i], scene.target_tmag[i], scene.target_starid[i], scene.catalog_of(i), np.ones((H, W), dtype='int32'))
		assert int(got['status'][i]) == ref['status']
		if 'mask' in ref:
			np.testing.assert_array_equal(got['mask'][i].astype(bool), ref['mask'])
			np.testing.assert_array_equal(got['lc'][0][i], ref['flux'])
			np.testing.assert_array_equal(got['lc'][1][i], ref['flux_err'])
			np.testing.assert_array_equal(got['lc'][2][i], ref['flux_background'])
		cat = scene.catalog_of(i)
		positions = np.empty((T, len(cat['starid']), 2))
		positions[:, :, 0] = cat['row_stamp'][None, :] + scene.jitter[:, 1][:, None]
		positions[:, :, 1] = cat['column_stamp'][None, :] + scene.jitter[:, 0][:, None]
		p = opsf.PSF(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'], tuple(scene.stamps[i]))
		rl = olin.do_photometry(img, p, cat, scene.target_starid[i], positions, tuple(scene.stamps[i]),
			scene.target_pos_row[i], scene.target_pos_column[i], np.ones((H, W), dtype='int32'))
		scale = np.nanmax(np.abs(rl['flux']))
		np.testing.assert_allclose(got['psf_flux'][i], rl['flux'], rtol=1e-8, atol=1e-9*scale)
		assert int(got['psf_status'][i]) == rl['status']
		np.testing.assert_allclose(got['psf_contamination'][i], rl['contamination'], rtol=1e-7, atol=1e-11)
	ctx.close()
