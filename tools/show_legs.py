#!/usr/bin/env python3
"""A few numbers of the last bench run (bench_legs.json) on one screen."""
import json, os
d = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench_legs.json')))
print('step', d['ms_per_step'], d['value'], 'numa', d['config'].get('host_numa_node'))
for r in d['rooflines']:
	print('  ', r['kernel'], r['avg_kernel_ms'], r['frac'])
if 'frames_to_results' in d:
	f = d['frames_to_results']
	print('frames', f['targets_per_s'], f.get('seconds_all_calls'), 'pipelined', f['pipelined']['targets_per_s'], f['pipelined'].get('seconds_all_runs'))
if 'frames_to_results_large_batch' in d:
	f = d['frames_to_results_large_batch']
	print('frames (10 000 targets)', f['targets_per_s'], f.get('seconds_all_calls'), 'pipelined', f['pipelined']['targets_per_s'], f['pipelined'].get('seconds_all_runs'))
if 'fit_background_frames' in d:
	fb = d['fit_background_frames']
	print('tess', fb['tess']['kernel_ms_per_frame'], 'plain', fb['plain']['kernel_ms_per_frame'], fb.get('parity_sample'))
if 'linpsf' in d:
	print('linpsf', d['linpsf']['ms_per_step'], d['linpsf']['roofline']['frac'], 'drift', d['linpsf'].get('drift', {}).get('ms_per_step'))
	print('linpsf_frames', d['psf_frames_to_results']['linpsf_frames']['targets_per_s'], 'psf_frames', d['psf_frames_to_results']['psf_frames']['targets_per_s'])
if 'psf_fit' in d:
	print('psf_fit ns/iter', d['psf_fit']['ns_per_simplex_iteration_chipwide'])
if 'cpu_baseline' in d:
	print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['parity_sample'])
if 'aperture_premade_cubes' in d:
	print('premade', d['aperture_premade_cubes']['ms_per_step'], d['aperture_premade_cubes']['roofline']['frac'])
print('stages', {k: v.get('avg_ms') for k, v in d.get('stages', {}).items()}, 'e2e', d.get('end_to_end', {}).get('targets_per_s'))
