# -*- coding: utf-8 -*-
"""
The bench's stdout line must stay machine-sized (round 4's 21 KB line could not be read by the driver): benchlib.line.compact
builds it from the full result, benchlib.line.check_line is the self-check bench.py runs before printing.  Canned full results:
the committed full line of round 4 (profiles/r4_bench_steps20_warmup5_unprofiled.json), as it is and with every string blown up.
"""
import copy
import io
import json
import os
import pytest
from benchlib import line as bl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANNED = os.path.join(ROOT, 'profiles', 'r4_bench_steps20_warmup5_unprofiled.json')


def _canned():
	return json.loads(open(CANNED).read().strip().splitlines()[-1])


def _inflate(o):
	"""Every string ten times as long, every dict with a 2 000-character note: the compact line must not grow with the prose."""
	if isinstance(o, dict):
		d = {k: _inflate(v) for k, v in o.items()}
		d['note'] = 'x' * 2000
		return d
	if isinstance(o, list):
		return [_inflate(v) for v in o]
	if isinstance(o, str) and len(o) > 40:
		return o * 10
	return o


def test_compact_line_of_a_canned_result():
	full = _canned()
	assert len(json.dumps(full)) > 20000                   # the line the driver could not read
	s = json.dumps(bl.compact(full), separators=(',', ':'))
	d = bl.check_line(s)
	assert len(s) < bl.MAX_LINE
	# the contract's fields survive with their values
	for k in ('metric', 'unit', 'n_gpus', 'steps', 'warmup', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data'):
		assert d[k] == full[k]
	assert d['value'] == pytest.approx(full['value'], rel=1e-5)
	assert d['ms_per_step'] == pytest.approx(full['ms_per_step'], rel=1e-5)
	assert d['config']['baseline_config'] == 'configs[2]'
	r, rf = d['roofline'], full['roofline']
	assert r['kernel'] == rf['kernel'] and r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000
	assert r['frac'] == pytest.approx(rf['frac'], rel=1e-5) and r['frac'] == pytest.approx(r['achieved'] / r['peak'], rel=1e-4)
	assert r['traffic'] == pytest.approx(rf['traffic'], rel=1e-5)
	assert r['traffic_source'].startswith('profiles/')
	c = d['cpu_baseline']
	assert c['kind'] == 'port' and c['cores'] == 16 and c['unit'] == 'targets/s' and c['value'] == pytest.approx(full['cpu_baseline']['value'], rel=1e-5)
	assert len(c['sample']) <= bl.MAX_STRING
	assert d['parity_sample'] == {'targets': 512, 'mismatches': 0, 'background_mismatches': 0, 'background_max_rel_err': 0}
	# one number per leg
	legs = d['legs']
	assert legs['linpsf']['ms_per_step'] == pytest.approx(full['linpsf']['ms_per_step'], rel=1e-5)
	assert legs['frames_to_results']['pipelined']['targets_per_s'] == pytest.approx(full['frames_to_results']['pipelined']['targets_per_s'], rel=1e-5)
	assert legs['psf_fit']['ns_per_simplex_iteration_chipwide'] > 0
	assert legs['fit_background_frames']['tess']['kernel_ms_per_frame'] > 0
	assert d['details'] == bl.LEGS_FILE


def test_compact_line_does_not_grow_with_the_prose():
	full = _inflate(_canned())
	assert len(json.dumps(full)) > 200000
	s = json.dumps(bl.compact(full), separators=(',', ':'))
	bl.check_line(s)
	assert len(s) < bl.MAX_LINE


def test_check_line_refuses_what_the_driver_cannot_read():
	good = bl.compact(_canned())
	with pytest.raises(AssertionError):
		bl.check_line(json.dumps(dict(good, note='y' * (bl.MAX_STRING + 1))))
	with pytest.raises(AssertionError):
		bl.check_line(json.dumps(dict(good, pad=['z' * 60] * 100)))
	bad = copy.deepcopy(good)
	del bad['roofline']['traffic']
	with pytest.raises(AssertionError):
		bl.check_line(json.dumps(bad))
	with pytest.raises(AssertionError):
		bl.check_line(json.dumps(good, indent=1))          # more than one line


def test_emit_writes_the_legs_file_and_returns_the_line(tmp_path):
	full = _canned()
	os.mkdir(tmp_path / 'gpurun_out')
	err = io.StringIO()
	s = bl.emit(full, str(tmp_path), stream=err)
	assert json.loads(s)['value'] == pytest.approx(full['value'], rel=1e-5)
	for d in (tmp_path, tmp_path / 'gpurun_out'):
		assert json.load(open(d / bl.LEGS_FILE)) == full   # nothing is lost: the full result is beside the line
	assert json.loads(err.getvalue()) == full


def test_multi_gpu_line_carries_the_gather_numbers():
	full = _canned()
	full['n_gpus'] = 8
	full['gather'].update(mode='rccl', mean_ms=3.2, final_ms=11.0, step_ms_without_gather=14.1, bytes_per_rank_per_step=783000000)
	d = bl.check_line(json.dumps(bl.compact(full), separators=(',', ':')))
	assert d['gather']['mean_ms'] == 3.2 and d['gather']['step_ms_without_gather'] == 14.1 and d['gather']['final_ms'] == 11
