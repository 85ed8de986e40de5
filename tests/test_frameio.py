# -*- coding: utf-8 -*-
"""The build's on-disk frame-stack container (photometry_amd/frameio.py): header / layout round trip on CPU."""
import os
import numpy as np
import pytest
from photometry_amd import frameio


def test_tpstack_round_trip(tmp_path):
	rng = np.random.default_rng(0)
	T, R, C = 7, 33, 41
	groups = {'images': rng.normal(size=(T, R, C)).astype('float32'), 'images_err': rng.random((T, R, C)).astype('float32'),
		'pixel_flags': rng.integers(0, 8, (T, R, C)).astype('uint8')}
	groups['images'][2, 3, 4] = np.nan
	path = str(tmp_path / 'ccd.tpstack')
	frameio.write_stack(path, groups, row_offset=0, col_offset=44, time=1325.0 + np.arange(T) / 48, cadenceno=np.arange(T) + 4697,
		quality=np.array([0, 0, 32, 0, 0, 4, 0]), attrs={'sector': 1, 'camera': 3, 'ccd': 2, 'cadence': 1800})
	meta = frameio.read_header(path)
	assert meta['shape'] == [T, R, C] and meta['col_offset'] == 44 and meta['attrs']['camera'] == 3
	assert meta['quality'] == [0, 0, 32, 0, 0, 4, 0] and abs(meta['time'][1] - (1325.0 + 1 / 48)) < 1e-12
	for name, a in groups.items():
		g = next(x for x in meta['groups'] if x['name'] == name)
		assert g['offset'] % 4096 == 0
		np.testing.assert_array_equal(frameio.open_group(path, name, meta), a)
	with pytest.raises(KeyError):
		frameio.open_group(path, 'backgrounds', meta)
	# a long time base does not fit the 4 KiB header: the metadata moves behind the data
	T2 = 1300
	big = {'images': np.zeros((T2, 2, 2), dtype='float32')}
	p2 = str(tmp_path / 'long.tpstack')
	frameio.write_stack(p2, big, time=np.arange(T2) * 0.0208333, cadenceno=np.arange(T2), quality=np.zeros(T2, dtype=int))
	m2 = frameio.read_header(p2)
	assert m2['shape'] == [T2, 2, 2] and len(m2['time']) == T2
	np.testing.assert_array_equal(frameio.open_group(p2, 'images', m2), big['images'])
	with open(str(tmp_path / 'junk.bin'), 'wb') as fh:
		fh.write(b'not a stack')
	with pytest.raises(ValueError):
		frameio.read_header(str(tmp_path / 'junk.bin'))
