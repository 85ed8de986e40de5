# -*- coding: utf-8 -*-
"""
Golden vectors produced by the reference's own code, run through the HIP path (not only through the oracle):

* ``golden_k2p2.npz``: the reference's ``k2p2FixFromSum`` (k2p2v2.py:344-623) on 56 sum images -> every mask it returned
  must come out of ``tp_k2p2_masks`` bit for bit when the target sits on a pixel of that mask;
* ``golden_psf.npz``: the reference's ``PSF.integrate_to_image`` (psf.py:122-148) -> an image built by the reference from
  known fluxes must be decomposed by ``tp_linpsf_fit`` (device P2 design matrix + P3 solve) into exactly those fluxes.
"""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
	from photometry_amd.device import Context
	c = Context(0)
	yield c
	c.close()


def _k2p2_golden_batch(ctx, g, cases):
	"""One device target per (case, mask) of same-shaped golden cases; returns (n targets, n bit-exact)."""
	from photometry_amd import simulate
	from test_gpu_k2p2 import run_device
	S, cats, want, pos = [], [], [], []
	for n in cases:
		masks = g[f'k{n}_masks'].astype(bool)
		cover = masks.sum(axis=0)
		for m in masks:
			own = np.argwhere(m & (cover == 1))
			if len(own) == 0:
				continue
			# the pixel of the mask with the largest flux (any pixel selects the mask, photometry.py:107-120)
			r, c = own[np.argmax(g[f'k{n}_sumimage'][own[:, 0], own[:, 1]])]
			S.append(g[f'k{n}_sumimage']); cats.append(g[f'k{n}_catalog']); want.append(m); pos.append((r, c))
	Nt = len(S)
	H, W = S[0].shape
	s = simulate.make_scene(Nt, 4, H, W, seed=1)
	s.aperture = np.ones((Nt, H, W), dtype='int32')
	s.stamps[:] = np.array([0, H, 44, 44 + W], dtype='int32')
	offs, col = [0], {k: [] for k in ('starid', 'tmag', 'row', 'column', 'row_stamp', 'column_stamp')}
	for i, cat in enumerate(cats):
		k = len(cat)
		col['starid'].append(np.arange(k, dtype='int64') + 100 * (i + 1))
		col['column_stamp'].append(cat[:, 0]); col['row_stamp'].append(cat[:, 1]); col['tmag'].append(cat[:, 2])
		col['column'].append(cat[:, 0] + 44); col['row'].append(cat[:, 1])
		offs.append(offs[-1] + k)
	s.catalog = {k: np.concatenate(v) for k, v in col.items()}
	s.cat_offsets = np.asarray(offs, dtype='int64')
	s.target_pos_row = np.array([p[0] for p in pos], dtype='float64')
	s.target_pos_column = np.array([p[1] + 44 for p in pos], dtype='float64')
	s.target_starid = np.array([100 * (i + 1) for i in range(Nt)], dtype='int64')
	s.target_tmag = np.array([c[0, 2] if len(c) else 10.0 for c in cats], dtype='float64')
	got = run_device(ctx, s, np.stack(S))
	for i in range(Nt):
		# ERROR is allowed only as "no catalog star in mask" (photometry.py:243-246), which is decided after the mask exists
		assert not (int(got['flags'][i]) & 1), (i, got['status'][i], hex(int(got['flags'][i])))
		np.testing.assert_array_equal(got['mask'][i].astype(bool), want[i], err_msg=f"golden mask {i} of shape {H}x{W}")
	return Nt


def test_golden_k2p2_masks_on_device(ctx, golden_dir):
	g = np.load(os.path.join(golden_dir, 'golden_k2p2.npz'))
	by_shape = {}
	for n in range(int(g['n_cases'])):
		if g[f'k{n}_masks'].shape[0]:
			by_shape.setdefault(g[f'k{n}_sumimage'].shape, []).append(n)
	total = sum(_k2p2_golden_batch(ctx, g, cases) for cases in by_shape.values())
	assert total >= 60 and len(by_shape) >= 3


def test_golden_psf_images_decompose_on_device(ctx, golden_dir):
	from photometry_amd import engine, psf as hpsf
	from photometry_amd.device import DeviceCube
	from scipy.interpolate import RectBivariateSpline
	g = np.load(os.path.join(golden_dir, 'golden_psf.npz'))
	x = g['prf_x']
	model = hpsf.PRFModel.from_spline(RectBivariateSpline(x, x, g['prf_img']))
	n_done = 0
	for n in range(int(g['n_cases'])):
		cutoff = float(g[f'p{n}_cutoff'])
		if not np.isfinite(cutoff):
			continue    # integrate_to_image without a cut-off radius: LinPSFPhotometry never calls it that way (:63, :146)
		params = g[f'p{n}_params']           # (row, column, flux) per star
		img = g[f'p{n}_img']                 # the reference's pixel-integrated image of those stars
		ns = len(params)
		T = 3
		cube = np.repeat(img[None, :, :, None], T, axis=3).astype('float32')
		star_offsets = np.array([0, ns], dtype='int64')
		pos_row = np.repeat(params[:, 0:1], 32, axis=1)
		pos_col = np.repeat(params[:, 1:2], 32, axis=1)
		coef = engine.linpsf_prf(ctx, ctx.array(model.base_coef), ctx.array(model.weights(np.zeros((1, 4)))))
		res = engine.linpsf_fit(ctx, DeviceCube.from_host(ctx, cube), coef, ctx.array(model.tx), ctx.array(model.ty),
			ctx.array(star_offsets), ctx.array(np.zeros(1, dtype='int32')), ctx.array(pos_row), ctx.array(pos_col), ns,
			cutoff_radius=cutoff).to_host()
		# float32 image: the fitted fluxes carry its 6e-8 rounding; stars whose cut-off disc holds no pixel stay 0
		for k in range(T):
			np.testing.assert_allclose(res['fluxes_all'][:ns, k], params[:, 2], rtol=2e-6)
		n_done += 1
	assert n_done >= 4
