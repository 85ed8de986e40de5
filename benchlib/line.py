# -*- coding: utf-8 -*-
"""
The ONE JSON line bench.py prints, kept machine-sized.

The full result of a run (every leg with its notes, per-kernel tables, samples) goes to ``bench_legs.json`` beside bench.py
(and under ``gpurun_out/`` when that directory exists) and to stderr.  The stdout line carries the contract's fields, ``config``,
``roofline`` (dominant kernel), ``rooflines`` (numbers only), ``cpu_baseline``, ``parity_sample`` and one or two numbers per leg;
no string in it is longer than ``MAX_STRING`` characters and the whole line stays below ``MAX_LINE`` -- round 4's line had grown
to 21 KB of prose and the driver could not read it.  ``emit`` checks both before anything is printed.
"""
import json
import os

MAX_LINE = 4000
MAX_STRING = 80
LEGS_FILE = 'bench_legs.json'


def _num(x):
	"""Numbers at six significant digits (a float printed in full is 18 characters), everything else unchanged."""
	if isinstance(x, bool) or x is None:
		return x
	if isinstance(x, int):
		return x
	if isinstance(x, float):
		if x != x or x in (float('inf'), float('-inf')):
			return None
		if x == int(x) and abs(x) < 1e15:
			return int(x)
		return float('%.6g' % x)
	return x


def _pick(d, keys, rename=None):
	"""The listed keys of ``d`` that are present and scalar, numbers rounded, strings cut at MAX_STRING."""
	out = {}
	if not isinstance(d, dict):
		return out
	for k in keys:
		if k not in d:
			continue
		v = d[k]
		if isinstance(v, (dict, list, tuple)):
			continue
		if isinstance(v, str):
			v = v[:MAX_STRING]
		out[(rename or {}).get(k, k)] = _num(v)
	return out


_ROOF = ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_kernel_ms', 'bytes_per_launch')


def _roofline(r):
	out = _pick(r, _ROOF)
	if out.get('bound') and len(out['bound']) > 12:
		out['bound'] = out['bound'].split(' ')[0]
	src = (r or {}).get('traffic_source')
	if src:
		out['traffic_source'] = src.split(' ')[0]
	return out


def _cpu(c, extra=()):
	out = _pick(c, ('value', 'unit', 'cores', 'kind') + tuple(extra))
	if isinstance(c, dict) and c.get('sample'):
		out['sample'] = c.get('sample_short') or c['sample'][:MAX_STRING]
	return out


def compact(result):
	"""The short form of bench.py's full result (see the module docstring)."""
	r = result
	out = _pick(r, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data'))
	out['metric'] = r['metric'][:MAX_STRING]
	cfg = r.get('config', {})
	out['config'] = _pick(cfg, ('baseline_config', 'targets_per_gpu', 'targets_total', 'cadences'))
	out['config']['workload'] = cfg.get('workload_short') or cfg.get('workload', '')[:MAX_STRING]
	out['config']['stamp'] = cfg.get('stamp')
	out['config']['parallelism'] = cfg.get('parallelism_short') or cfg.get('parallelism', '')[:MAX_STRING]
	out['roofline'] = _roofline(r.get('roofline'))
	out['rooflines'] = [_pick(x, ('kernel', 'achieved', 'frac', 'frac_counter_bytes', 'traffic', 'avg_kernel_ms')) for x in r.get('rooflines', [])]
	if 'step_hbm' in r:
		out['step_hbm'] = _pick(r['step_hbm'], ('necessary_bytes_per_step', 'GBps_over_whole_step', 'frac_of_hbm_peak', 'mean_mask_pixels'))
	if 'cpu_baseline' in r:
		out['cpu_baseline'] = _cpu(r['cpu_baseline'], ('single_core_targets_per_s',))
		out['parity_sample'] = _pick(r.get('parity_sample'), ('targets', 'mismatches', 'background_mismatches', 'background_max_rel_err'))
		out.update(_pick(r, ('speedup_vs_cpu_baseline', 'speedup_vs_one_core')))
	g = r.get('gather')
	if g and r.get('n_gpus', 1) > 1:
		out['gather'] = _pick(g, ('mode', 'issued_short', 'bytes_per_rank_per_step', 'mean_ms', 'final_ms', 'step_ms_without_gather', 'ideal_ms_one_xgmi_link', 'measured_8gpu', 'compact'))
	if 'linpsf_roofline' in r:
		out['linpsf_roofline'] = _pick(r['linpsf_roofline'], ('achieved', 'peak', 'unit', 'frac', 'kernel_ms_per_step', 'fitted_stars'))
	if 'warning' in r:
		out['warning'] = r['warning'][:MAX_STRING]
	legs = {}
	p = r.get('aperture_premade_cubes')
	if p:
		legs['aperture_premade_cubes'] = dict(_pick(p, ('targets_per_s', 'ms_per_step')), **_pick(p.get('roofline'), ('frac', 'avg_kernel_ms', 'traffic')))
	s = r.get('stages')
	if s:
		legs['stages'] = {k: _pick(v, ('avg_ms', 'frac_of_hbm_peak')) for k, v in s.items() if isinstance(v, dict)}
	e = r.get('end_to_end')
	if e:
		legs['end_to_end'] = _pick(e, ('targets_per_s', 'h2d_GBps'))
	lp = r.get('linpsf')
	if lp:
		d = _pick(lp, ('value', 'ms_per_step', 'fitted_stars'), {'value': 'targets_per_s'})
		d.update(_pick(lp.get('roofline'), ('frac', 'achieved', 'kernel_ms_per_step'), {'achieved': 'tflops'}))
		d['fit_kernel_ms'] = _num((lp.get('kernels', {}).get('tp_linpsf_fitm_kernel') or {}).get('ms_per_step'))
		d['cpu_targets_per_s'] = _num((lp.get('cpu_baseline') or {}).get('value'))
		d['parity_mismatches'] = (lp.get('parity_sample') or {}).get('mismatches')
		if lp.get('drift'):
			d['drift'] = _pick(lp['drift'], ('ms_per_step', 'ratio_to_no_drift'))
		legs['linpsf'] = d
	f = r.get('frames_to_results')
	if f:
		d = _pick(f, ('targets_per_s', 'seconds', 'targets_resized'))
		if f.get('pipelined'):
			d['pipelined'] = _pick(f['pipelined'], ('targets_per_s', 'batches', 'in_flight'))
		if f.get('with_every_per_target_object'):
			d['with_objects_targets_per_s'] = _num(f['with_every_per_target_object'].get('targets_per_s'))
		legs['frames_to_results'] = d
	fl = r.get('frames_to_results_large_batch')
	if fl:
		d = _pick(fl, ('targets_per_s', 'seconds', 'targets_resized'))
		if fl.get('pipelined'):
			d['pipelined'] = _pick(fl['pipelined'], ('targets_per_s', 'batches', 'in_flight'))
		legs['frames_to_results_large_batch'] = d
	pf = r.get('psf_frames_to_results')
	if pf:
		legs['psf_frames_to_results'] = {k: _pick(v, ('targets_per_s', 'finite_fraction')) for k, v in pf.items() if isinstance(v, dict)}
	ps = r.get('psf_fit')
	if ps:
		d = _pick(ps, ('targets_per_s_at_1300_cadences', 'ns_per_simplex_iteration_chipwide', 'finite_fraction', 'kernel_ms'))
		d['frac'] = _num((ps.get('roofline') or {}).get('frac'))
		d['parity_mismatches'] = (ps.get('parity_sample') or {}).get('mismatches')
		legs['psf_fit'] = d
	fb = r.get('fit_background_frames')
	if fb:
		d = {}
		for k in ('plain', 'tess', 'shenanigans'):
			if isinstance(fb.get(k), dict):
				d[k] = _pick(fb[k], ('kernel_ms_per_frame', 'frames_per_s'))
				d[k]['frac'] = _num((fb[k].get('roofline') or {}).get('frac'))
				d[k]['cpu_frames_per_s'] = _num((fb[k].get('cpu_baseline') or {}).get('value'))
		d['parity'] = _pick(fb.get('parity_sample'), ('plain_max_rel_err', 'tess_max_rel_err'))
		legs['fit_background_frames'] = d
	if legs:
		out['legs'] = legs
	out['details'] = LEGS_FILE
	return out


def check_line(line):
	"""What the driver needs of the line: one line, parseable, short, no long strings, the contract's objects present."""
	assert '\n' not in line
	d = json.loads(line)
	assert len(line) < MAX_LINE, f'bench line is {len(line)} characters (limit {MAX_LINE})'

	def walk(o, path):
		if isinstance(o, str):
			assert len(o) <= MAX_STRING, f'string of {len(o)} characters at {path}'
		elif isinstance(o, dict):
			for k, v in o.items():
				walk(v, path + '.' + k)
		elif isinstance(o, (list, tuple)):
			for i, v in enumerate(o):
				walk(v, f'{path}[{i}]')
	walk(d, '')
	for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
		assert k in d, k
	assert 'workload' in d['config']
	for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
		assert k in d['roofline'], k
	return d


def emit(result, root, stream=None):
	"""Write the full result to LEGS_FILE (beside bench.py and under gpurun_out/ if present), return the checked compact line."""
	full = json.dumps(result, indent=1, sort_keys=False)
	for d in (root, os.path.join(root, 'gpurun_out')):
		if os.path.isdir(d):
			try:
				with open(os.path.join(d, LEGS_FILE), 'w') as fh:
					fh.write(full + '\n')
			except OSError:
				pass
	if stream is not None:
		stream.write(json.dumps(result) + '\n')
		stream.flush()
	line = json.dumps(compact(result), separators=(',', ':'))
	check_line(line)
	return line
