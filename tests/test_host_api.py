# -*- coding: utf-8 -*-
"""
CPU tests of the host-side plugin API: stamp geometry (known answers of the reference's
tests/test_basephotometry.py), dispatch and error conventions (tessphot.py), table helpers,
target sharding and skip-target replay.  No GPU needed: numerics run only in the gpu tests.
"""
import os
import numpy as np
import pytest
import photometry_amd
from photometry_amd import STATUS, tessphot, simulate
from photometry_amd.plugins import BasePhotometry, AperturePhotometry, Table
from photometry_amd.source import MemoryStampSource, source_from_scene
from photometry_amd import comm as tpcomm


def _region_source(R=60, C=80, T=6, row0=30, col0=44, tmag=10.0):
	frames = {k: np.zeros((R, C, T), dtype='float32') for k in ('images', 'images_err', 'backgrounds')}
	cat = {'starid': np.array([1, 2]), 'tmag': np.array([tmag, 12.0], dtype='float32'),
		'row': np.array([55.2, 57.0], dtype='float32'), 'column': np.array([60.7, 63.0], dtype='float32')}
	return MemoryStampSource(frames, row0, col0, np.arange(T)*0.02, np.zeros(T), np.arange(T), np.zeros(T, dtype='int32'), cat)


def test_status_values():
	# photometry/BasePhotometry.py:48-59
	assert [STATUS.UNKNOWN.value, STATUS.OK.value, STATUS.ERROR.value, STATUS.WARNING.value, STATUS.ABORT.value,
		STATUS.SKIPPED.value, STATUS.STARTED.value] == [0, 1, 2, 3, 4, 5, 6]


def test_stamp_geometry_known_answers():
	"""tests/test_basephotometry.py:59-173 of the reference: default stamp, 1-based pixel grid, resize."""
	src = _region_source()
	with BasePhotometry(1, src, None, datasource='ffi') as pho:
		assert pho.stamp == (48, 63, 54, 69) # round(55.2)=55, round(60.7)=61; 15 rows / cols: pos - 7 .. pos + 8
		cols, rows = pho.get_pixel_grid()
		assert rows.shape == (15, 15) and cols.shape == (15, 15)
		assert rows[0, 0] == 49 and cols[0, 0] == 55 and rows[-1, 0] == 63 and cols[0, -1] == 69 # 1-based
		assert abs(pho.target_pos_row_stamp - (55.2 - 48)) < 1e-5 and abs(pho.target_pos_column_stamp - (60.7 - 54)) < 1e-5
		# resize (BasePhotometry.py:567-613)
		assert pho.resize_stamp(up=12)
		assert pho.stamp == (48, 75, 54, 69)
		assert pho.resize_stamp(down=2, left=3, right=4)
		assert pho.stamp == (46, 75, 51, 73)
		assert pho._details['stamp_resizes'] == 2
		assert pho.resize_stamp(width=11, height=17)
		assert pho.stamp == (55 - 8, 55 + 9, 61 - 5, 61 + 6)
		# growing beyond the frame is clipped; no change -> False
		assert pho.resize_stamp(up=1000) and pho.stamp[1] == 90
		assert pho.resize_stamp(up=10) is False
		# the explicit stamp of the reference's test: (50, 60, 50, 70) -> rows 51..60, cols 51..70
		pho._stamp = (50, 60, 50, 70)
		cols, rows = pho.get_pixel_grid()
		assert rows[0, 0] == 51 and rows[-1, 0] == 60 and cols[0, 0] == 51 and cols[0, -1] == 70


def test_default_stamp_bright_star():
	src = _region_source(tmag=2.0)
	with BasePhotometry(1, src, None) as pho:
		Nrows, Ncols = pho.default_stamp()
		assert Nrows > 100 and Ncols > 50 # BasePhotometry.py:541-564 lookup table
		assert pho.stamp[:2] == tuple(src.max_stamp[:2]) # rows clipped to the frame
		assert pho.stamp[2] == 44 and pho.stamp[3] == 61 + int(Ncols)//2 + 1


def test_catalog_and_table():
	src = _region_source()
	with BasePhotometry(1, src, None) as pho:
		cat = pho.catalog
		assert len(cat) == 2 and cat['row_stamp'].dtype == np.float32
		np.testing.assert_allclose(cat['row_stamp'], cat['row'] - pho.stamp[0], atol=1e-4)
		rows = list(cat)
		assert rows[0]['starid'] == 1
		sub = cat[[1]]
		assert len(sub) == 1 and sub[0]['starid'] == 2
		assert len(cat[[]]) == 0
	t = Table(a=np.arange(3))
	t['b'] = np.arange(3) * 2
	assert 'b' in t and t[2]['b'] == 4


def test_invalid_inputs():
	src = _region_source()
	with pytest.raises(ValueError):
		tessphot('nonexistent', 1, src, None) # tessphot.py:128-129
	with pytest.raises(ValueError):
		BasePhotometry(1, src, None, datasource='invalid')
	with pytest.raises(FileNotFoundError):
		BasePhotometry(1, '/not/a/source', None)
	with pytest.raises(RuntimeError):
		BasePhotometry(12345, src, None) # star not in catalog (BasePhotometry.py:413-414)
	with BasePhotometry(1, src, None) as pho:
		with pytest.raises(NotImplementedError):
			pho.do_photometry()


def test_errors_become_status_error(tmp_path):
	"""tessphot.py:37-49: any exception in the plugin -> STATUS.ERROR + traceback in details."""
	src = _region_source()
	pho = tessphot('halo', 1, src, str(tmp_path))
	assert pho.status == STATUS.ERROR and pho.method == 'halo'
	assert any('NotImplementedError' in e for e in pho._details['errors'])
	# constructor failure -> error dummy
	pho = tessphot('aperture', 999, src, str(tmp_path))
	assert pho.status == STATUS.ERROR and pho.method == 'error'


def test_aperture_without_gpu_is_error_not_fallback(tmp_path):
	import ctypes
	from photometry_amd import _lib
	n = ctypes.c_int(0)
	_lib.load().tp_device_count(ctypes.byref(n))
	if n.value > 0:
		pytest.skip("a GPU is visible")
	s = simulate.make_scene(1, 8, 11, 11, seed=1)
	simulate.fill_cubes(s)
	pho = tessphot('aperture', int(s.target_starid[0]), source_from_scene(s, 0), str(tmp_path))
	assert pho.status == STATUS.ERROR
	assert any('libtessphot_hip' in e or 'HIP' in e for e in pho._details['errors'])


def test_shard_ranges_and_assembly():
	for n, w in ((10, 2), (10, 3), (100000, 8), (5, 8), (0, 2)):
		rs = [tpcomm.shard_range(n, w, r) for r in range(w)]
		assert rs[0][0] == 0 and rs[-1][1] == n
		assert all(rs[i][1] == rs[i+1][0] for i in range(w-1))
		assert sum(tpcomm.shard_sizes(n, w)) == n
	rng = np.random.default_rng(0)
	full = rng.normal(size=(5, 11, 7))
	sizes = tpcomm.shard_sizes(11, 3)
	cap = max(sizes)
	blocks = []
	for r in range(3):
		a, b = tpcomm.shard_range(11, 3, r)
		blk = np.zeros((5, cap, 7))
		blk[:, :b-a] = full[:, a:b]
		blocks.append(blk)
	np.testing.assert_array_equal(tpcomm.assemble_gathered(blocks, sizes), full)


def test_replay_skip_targets():
	"""photometry.py:244-250 -> taskmanager.py:460-532: the fainter star inside a brighter star's mask is skipped."""
	starids = [10, 11, 12, 13]
	tmags = [9.0, 12.0, 8.0, 11.0]
	skip = [[11], [10], [], []]       # 10 and 11 sit in each other's masks
	st = tpcomm.replay_skip_targets(starids, tmags, skip, [1, 1, 1, 3])
	assert list(st) == [1, 5, 1, 3]
	st = tpcomm.replay_skip_targets(starids, tmags, [[], [10], [], []], [1, 1, 2, 1])
	assert list(st) == [1, 5, 2, 1] # 11 lies in the mask of the brighter 10 -> 11 is skipped


def test_replay_skip_targets_reference_known_answers():
	"""tests/test_taskmanager.py:304-400 of the reference (FFI cases): the bright star (Tmag 2.216) with the faint one (14.574)
	in its skip list keeps OK and the faint one is SKIPPED; the faint one saved first with the bright one in its list is
	SKIPPED itself and the bright one stays to be processed."""
	starids, tmags = [267211065, 261522674], [2.216, 14.574]
	st, ran = tpcomm.replay_skip_targets(starids, tmags, [[261522674], []], [1, 1], return_ran=True)
	assert list(st) == [1, 5] and ran == [0]
	# the reference's second test saves the faint star first (it asked for that task explicitly): priorities reversed
	st, ran = tpcomm.replay_skip_targets(starids, tmags, [[], [267211065]], [1, 1], priorities=[2, 1], return_ran=True)
	assert list(st) == [1, 5] and ran == [1, 0]


def test_replay_skip_targets_golden():
	"""The replay against the reference's own TaskManager (get_task / start_task / save_result on sqlite todo-lists,
	tests/golden/make_golden.py:golden_skiptargets): final statuses and the order in which targets ran, 60 todo-lists with
	Tmag ties, arbitrary priorities, self-references and names that are no targets."""
	import os
	g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'golden_skiptargets.npz'))
	n_skipped = 0
	for c in range(int(g['n_cases'])):
		off, flat = g[f's{c}_skip_offsets'], g[f's{c}_skip_flat']
		skip = [list(flat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
		st, ran = tpcomm.replay_skip_targets(g[f's{c}_starid'], g[f's{c}_tmag'], skip, g[f's{c}_status_in'],
			priorities=g[f's{c}_priority'], return_ran=True)
		np.testing.assert_array_equal(st, g[f's{c}_status_out'], err_msg=f'case {c}')
		assert ran == list(g[f's{c}_ran']), c
		n_skipped += int((st == 5).sum())
	assert n_skipped > 100


def test_batch_helpers_equal_per_target_helpers():
	"""The vectorised host helpers of the batched path (stamps.default_stamps, pipeline._catalogs_of_stamps) against the
	per-target functions the plugin uses, on random regions: half-integer positions (rounding mode), stars exactly on the
	half-pixel limits of the 5-pixel catalogue buffer, stamps clipped by the frame, invalid stamps."""
	from photometry_amd import stamps as st, pipeline as pl
	rng = np.random.default_rng(0)
	n_stamps = 0
	for trial in range(25):
		n = int(rng.integers(5, 300))
		R, C = int(rng.integers(30, 400)), int(rng.integers(30, 400))
		limits = (100, 100 + R, 200, 200 + C)
		rows, cols, tm = rng.uniform(90, 110 + R, n), rng.uniform(190, 210 + C, n), rng.uniform(1.5, 15, n)
		rows[:3] = np.round(rows[:3]) + 0.5
		S, valid = st.default_stamps(rows, cols, tm, limits)
		for i in range(n):
			try:
				ref = st.default_stamp(rows[i], cols[i], tm[i], limits)
				assert valid[i] and tuple(S[i]) == ref
			except ValueError:
				assert not valid[i]
		cat = {'starid': np.arange(n) + 7, 'tmag': tm.astype('float32'), 'row': rows.astype('float32'), 'column': cols.astype('float32')}
		cat['row'][:5] = np.array([S[0][0] - 5.5, S[0][1] + 4.5, S[0][0] - 5.5, S[0][1] + 4.49, S[0][0]], dtype='float32')
		cat['column'][:2] = np.array([S[0][2] - 5.5, S[0][3] + 4.5], dtype='float32')
		sel = [tuple(s) for s, v in zip(S.tolist(), valid) if v]
		if not sel:
			continue
		off, arr = pl._catalogs_of_stamps(pl._CatalogIndex(cat), sel)
		for j, s in enumerate(sel):
			ref = pl._catalog_of_stamp(cat, s)
			for k in ref:
				a = arr[k][off[j]:off[j + 1]]
				assert a.dtype == ref[k].dtype
				np.testing.assert_array_equal(a, ref[k], err_msg=f'{trial} {j} {k}')
		n_stamps += len(sel)
	assert n_stamps > 1000


def _write_prf_mat(path, prf):
	"""A MATLAB file with the structure of the SPOC PRF files (prfStruct: one struct entry per PRF sample)."""
	from scipy.io import savemat
	n = len(prf['ccdRow'])
	dt = [('values', 'O'), ('ccdRow', 'O'), ('ccdColumn', 'O'), ('prfRow', 'O'), ('prfColumn', 'O')]
	st = np.empty((1, n), dtype=dt)
	for i in range(n):
		st[0, i] = (prf['values'][i], np.array([[prf['ccdRow'][i]]]), np.array([[prf['ccdColumn'][i]]]),
			prf['prfRow'][None, :], prf['prfColumn'][None, :])
	savemat(path, {'prfStruct': st})


def test_prf_mat_loader_and_file_rule(tmp_path):
	"""PRFModel.from_mat / for_ccd: the file rule of psf.py:66-72 and the prfStruct unpacking of :81-104, on a file written
	with scipy.io.savemat from the synthetic PRF; the loaded model equals the one built from the arrays."""
	from photometry_amd import simulate, psf as hpsf
	prf = simulate.synthetic_prf(seed=4, n_side=3)
	for d, cam, ccd in (('start_s0001', 2, 3), ('start_s0004', 2, 3)):
		os.makedirs(tmp_path / d, exist_ok=True)
		# the two characterisations differ, so that the sector rule is visible in the result
		p2 = dict(prf)
		p2['values'] = prf['values'] * (1.0 if d == 'start_s0001' else 2.0)
		_write_prf_mat(str(tmp_path / d / f'tess2018243163600-prf-{cam}-{ccd}-characterized-prf.mat'), p2)
	assert 'start_s0001' in hpsf.prf_file(str(tmp_path), 3, 2, 3) and 'start_s0004' in hpsf.prf_file(str(tmp_path), 4, 2, 3)
	with pytest.raises(ValueError):
		hpsf.prf_file(str(tmp_path), 0, 2, 3)
	with pytest.raises(ValueError):
		hpsf.prf_file(str(tmp_path), 1, 5, 3)
	with pytest.raises(ValueError):
		hpsf.prf_file(str(tmp_path), 1, 2, 0)
	with pytest.raises(FileNotFoundError):
		hpsf.prf_file(str(tmp_path), 1, 1, 1)
	ref = hpsf.PRFModel(prf['values'], prf['ccdColumn'], prf['ccdRow'], prf['prfColumn'], prf['prfRow'])
	m1 = hpsf.PRFModel.for_ccd(str(tmp_path), 2, 2, 3)
	m4 = hpsf.PRFModel.for_ccd(str(tmp_path), 11, 2, 3)
	np.testing.assert_array_equal(m1.base_coef, ref.base_coef)
	np.testing.assert_array_equal(m1.ccd_column, ref.ccd_column)
	np.testing.assert_array_equal(m1.ccd_row, ref.ccd_row)
	np.testing.assert_array_equal(m1.tx, ref.tx)
	np.testing.assert_allclose(m4.sums, 2.0 * ref.sums, rtol=1e-15)
	stamps = np.array([[100, 115, 300, 315]])
	np.testing.assert_allclose(m1.weights(stamps), ref.weights(stamps), rtol=1e-15)
	# the normalised blend does not care about the scale of the file: same coefficient table
	np.testing.assert_allclose(m4.weights(stamps) @ m4.base_coef, m1.weights(stamps) @ m1.base_coef, rtol=1e-12, atol=1e-18)


def test_plugin_psf_from_the_prf_directory(tmp_path, monkeypatch):
	"""BasePhotometry.psf without a model on the source: the CCD's PRF file from TESSPHOT_PSF_DIR (psf.py:66-72)."""
	from photometry_amd import simulate, plugins, psf as hpsf
	prf = simulate.synthetic_prf(seed=5, n_side=2)
	os.makedirs(tmp_path / 'start_s0001')
	_write_prf_mat(str(tmp_path / 'start_s0001' / 'tess2018243163600-prf-1-1-characterized-prf.mat'), prf)

	class Src: prf = None; psf_dir = None
	pho = plugins.BasePhotometry.__new__(plugins.BasePhotometry)
	pho.source, pho._psf, pho.sector, pho.camera, pho.ccd = Src(), None, 1, 1, 1
	monkeypatch.delenv('TESSPHOT_PSF_DIR', raising=False)
	with pytest.raises(FileNotFoundError):
		pho.psf
	monkeypatch.setenv('TESSPHOT_PSF_DIR', str(tmp_path))
	model = pho.psf
	assert isinstance(model, hpsf.PRFModel) and model.n_hdu == 4
	pho2 = plugins.BasePhotometry.__new__(plugins.BasePhotometry)
	pho2.source, pho2._psf, pho2.sector, pho2.camera, pho2.ccd = Src(), None, 1, 1, 1
	assert pho2.psf is model   # fitted once per process


def test_binned_catalogue_selection_equals_the_per_stamp_function():
	"""pipeline._catalogs_of_stamps (all stamps of a group at once, candidates from the cells of a binned catalogue) against
	pipeline._catalog_of_stamp (the reference's selection, BasePhotometry.py:1094-1181, one stamp at a time): same stars in the same
	(catalogue) order with the same float32 stamp coordinates -- stamps of every size, stars on the stamp's buffer limits, stars
	without a position."""
	from photometry_amd import pipeline as pl
	rng = np.random.default_rng(3)
	N = 3000
	cat = {'starid': np.arange(N) + 1, 'tmag': rng.uniform(8, 15, N).astype('float32'), 'row': rng.uniform(-3, 520, N).astype('float32'),
		'column': rng.uniform(40, 560, N).astype('float32')}
	cat['row'][5] = np.nan
	cat['row'][6], cat['column'][6] = 94.5, 139.5          # exactly on the lower limits of the first stamp's buffer
	cat['row'][7], cat['column'][7] = 119.5, 169.5         # exactly on the (excluded) upper limits
	st = [(100, 115, 145, 165)]
	for _ in range(600):
		r, c, h, w = rng.integers(0, 500), rng.integers(44, 540), rng.integers(5, 40), rng.integers(5, 40)
		st.append((r, r + h, c, c + w))
	st = np.array(st)
	off, arr = pl._catalogs_of_stamps(pl._CatalogIndex(cat), st)
	assert off[0] == 0 and len(off) == len(st) + 1
	for i in range(len(st)):
		ref = pl._catalog_of_stamp(cat, tuple(st[i]))
		for k in ref:
			np.testing.assert_array_equal(arr[k][off[i]:off[i + 1]], ref[k], err_msg=f'stamp {i} {k}')
	first = set(arr['starid'][off[0]:off[1]])
	assert 7 in first and 8 not in first
	# an empty catalogue and an empty group
	e = {k: v[:0] for k, v in cat.items()}
	off0, arr0 = pl._catalogs_of_stamps(pl._CatalogIndex(e), st[:3])
	assert list(off0) == [0, 0, 0, 0] and len(arr0['starid']) == 0


def test_bind_host_to_device_leaves_the_affinity_alone_without_a_gpu():
	"""device.bind_host_to_device: the NUMA node of the GPU's PCIe slot when there is one (the process is then restricted to that
	node's cores), None -- and an untouched affinity -- when the node is unknown (no GPU here)."""
	import ctypes
	from photometry_amd import _lib
	from photometry_amd.device import bind_host_to_device
	n = ctypes.c_int(0)
	_lib.load().tp_device_count(ctypes.byref(n))
	before = os.sched_getaffinity(0)
	node = bind_host_to_device(0)
	after = os.sched_getaffinity(0)
	try:
		if n.value <= 0:
			assert node is None and after == before
		else:
			assert node is None or (isinstance(node, int) and node >= 0 and after <= before and len(after) > 0)
	finally:
		os.sched_setaffinity(0, before)
