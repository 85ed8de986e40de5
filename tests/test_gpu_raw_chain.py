# -*- coding: utf-8 -*-
"""
The parity chain of BASELINE configs[2] (aperture + background on RAW cubes) closed from the raw cube with NOTHING of the
device on the oracle's side:

    raw cube --oracle B*--> per-cadence background --oracle B2--> smoothed series --oracle B3--> subtracted cube
             --oracle A1--> sum image --oracle K2P2 / A5b--> mask --oracle A6 / A7--> light curve

against what ``tp_background_sumimage`` + ``tp_aperture_photometry_from_sumimage`` (the raw-cube step of the bench) return.
B* is build-defined and defined to the last bit (oracle/backgrounds.py), so both background series are compared BIT FOR BIT; the
float32 light-curve columns and the masks are bit for bit as well.  The one float64 quantity whose summation order differs
between the sides is the sum image (1e-12): a mask may therefore differ only where a sum-image pixel sits within rounding of
K2P2's threshold -- reported as a razor case by ``k2p2_common.own_chain_check`` and bounded (none on the committed seed).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_target(job):
	"""One target, oracle only, from its raw cube.  Runs in a worker process."""
	import numpy as np
	from oracle import backgrounds as ob, sumimage as osum, aperture as oap
	raw, err, quality, time_smooth, stamp, pos_row, pos_col, tmag, starid, catalog, H, W = job
	bkg_raw = ob.background_series(raw)                                    # B*
	bkg = ob.smooth_time(bkg_raw, time_smooth)                             # B2 (prepare.py:317-335)
	series = bkg[None, None, :]
	img, e2 = ob.subtract_background(raw, err, series)                     # B3 (prepare.py:419-425)
	S = osum.sumimage(img, quality)                                        # A1
	try:
		ref = oap.do_photometry(S, img, e2, np.broadcast_to(series.astype('float32'), img.shape), stamp, pos_row, pos_col, tmag, starid, catalog,
			np.ones((H, W), dtype='int32'))
	except Exception as e: # noqa: B902 -- tessphot.py:37-49: any exception is STATUS.ERROR
		ref = {'status': oap.STATUS_ERROR, 'exception': repr(e)}
	return {'bkg_raw': bkg_raw, 'bkg': bkg, 'S': S, 'ref': {k: ref.get(k) for k in ('status', 'mask', 'flux', 'flux_err', 'flux_background', 'pos_centroid', 'contamination')}}


def test_raw_chain_oracle_only():
	import multiprocessing as mp
	from photometry_amd import simulate, engine, pipeline
	from photometry_amd.device import Context
	from oracle import aperture as oap, k2p2 as ok2p2
	ctx = Context(0)
	Nt, T, H, W = 256, 1300, 15, 15
	time_smooth = 3
	scene = simulate.make_scene(Nt, T, H, W, seed=2)          # the bench's scene generator, configs[2]'s shape
	scene.aperture = None
	cubes = engine.synth_fill(ctx, scene, images=False, images_err=True, backgrounds=False, raw=True)
	batch = pipeline.ApertureBatch(ctx, scene, cubes={'raw': cubes['raw'], 'raw_err': cubes['images_err']})
	work = pipeline.ApertureWork(ctx, batch)
	pipeline.aperture_step(ctx, batch, work)                  # tp_background_sumimage + tp_aperture_photometry_from_sumimage
	ctx.sync()
	raw = cubes['raw'].to_host()
	err = cubes['images_err'].to_host()
	got = {'bkg_raw': work.bkg_raw.to_host()[:, :T], 'bkg': work.bkg.to_host()[:, :T], 'S': work.sumimage.to_host().reshape(Nt, H, W),
		'mask': work.mask.to_host().reshape(Nt, H, W), 'status': work.status.to_host(), 'lc': work.lc.to_host()}
	ctx.close()

	jobs = [(raw[i], err[i], scene.quality, time_smooth, tuple(scene.stamps[i]), scene.target_pos_row[i], scene.target_pos_column[i],
		scene.target_tmag[i], scene.target_starid[i], scene.catalog_of(i), H, W) for i in range(Nt)]
	with mp.get_context('fork').Pool(8) as pool:
		orc = pool.map(_oracle_target, jobs, chunksize=4)

	n_exact = n_razor = n_err = 0
	for i, o in enumerate(orc):
		# the two background series: bit for bit, NaN positions included
		np.testing.assert_array_equal(got['bkg_raw'][i], o['bkg_raw'], err_msg=f"B* of target {i}")
		np.testing.assert_array_equal(got['bkg'][i], o['bkg'], err_msg=f"B2 of target {i}")
		# the sum image: float64 sums of the same float32 differences in another order
		np.testing.assert_allclose(got['S'][i], o['S'], rtol=1e-12, atol=0, equal_nan=True)
		ref = o['ref']
		same = int(got['status'][i]) == ref['status'] and (ref['mask'] is None or np.array_equal(got['mask'][i].astype(bool), ref['mask']))
		if not same:
			thr = ok2p2.threshold(o['S'], 0.8, full_output=True)
			with np.errstate(invalid='ignore'):
				margin = np.nanmin(np.abs(o['S'] - thr['CUT']))
			assert margin <= 8e-6 * max(1.0, abs(thr['CUT'])), f"target {i}: masks differ off the razor's edge (margin {margin}, CUT {thr['CUT']}, status {got['status'][i]} vs {ref['status']})"
			n_razor += 1
			continue
		if ref['status'] == oap.STATUS_ERROR or ref['mask'] is None:
			n_err += 1
			continue
		np.testing.assert_array_equal(got['lc']['flux'][i], ref['flux'], err_msg=f"flux {i}")
		np.testing.assert_array_equal(got['lc']['flux_err'][i], ref['flux_err'], err_msg=f"flux_err {i}")
		np.testing.assert_array_equal(got['lc']['flux_background'][i], ref['flux_background'], err_msg=f"flux_background {i}")
		np.testing.assert_allclose(got['lc']['pos_centroid'][i], ref['pos_centroid'], rtol=1e-12, equal_nan=True)
		n_exact += 1
	print(f"raw chain, oracle only: {n_exact} of {Nt} targets bit-exact (masks, flux, flux_err, flux_background; B* and B2 series of all {Nt}), "
		f"{n_err} without a mask on both sides, {n_razor} razor cases")
	assert n_razor == 0, "a razor case on the committed seed"
	assert n_exact >= 0.9 * Nt
