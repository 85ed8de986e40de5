#!/usr/bin/env python3
"""
Summarise rocprofv3 CSV output (kernel stats + PMC counters) of profiles/run_profile.sh into a small text table and a
traffic JSON (HBM bytes per launch of every kernel, read by bench.py).

FETCH_SIZE on gfx950 reports half the bytes of SOME access patterns (MI355X_MICROARCH.md: wide coalesced streaming reads) and
the full bytes of others (e.g. the read-modify-write stream of the materialised subtraction).  The factor is therefore not
assumed: for every kernel whose necessary bytes are known (bench.py prints them) the factor f in {1, 2} is the smallest
one with f x FETCH_SIZE + WRITE_SIZE >= 0.97 x necessary bytes -- a kernel cannot have moved less than it needs -- and it
is recorded next to the number.  Kernels without a known byte count get no traffic figure, only the raw counters.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
	return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def short(name):
	"""tp_xxx_kernel plus its template arguments (the fused kernel runs in two configurations)."""
	m = re.search(r'(tp_\w+)(<[^>]*>)?', name)
	return (m.group(1) + (m.group(2) or '')) if m else name


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find('trace/**/*kernel_stats.csv'):
	with open(f) as fh:
		for r in csv.DictReader(fh):
			if 'tp_' in r.get('Name', ''):
				print(f"{short(r['Name']):64s} calls={r.get('Calls'):>4s} avg_ns={r.get('AverageNs'):>12s} total_ns={r.get('TotalDurationNs'):>14s} pct={r.get('Percentage')}")

means = defaultdict(dict)
for counter, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
	acc = defaultdict(list)
	for f in find(f'{sub}/**/*counter_collection.csv'):
		with open(f) as fh:
			for r in csv.DictReader(fh):
				if r.get('Counter_Name') == counter and 'tp_' in r.get('Kernel_Name', ''):
					acc[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
	for k, v in acc.items():
		means[k][counter] = sum(v) / len(v) * 1024.0     # reported in KiB
		means[k]['dispatches_' + counter] = len(v)

# necessary bytes per launch from the bench line of the trace run
need = {}
for f in ('bench_trace.json',):
	try:
		r = json.loads(open(os.path.join(out, f)).read().strip().splitlines()[-1])
	except Exception: # noqa: B902
		continue
	for k, v in r.get('kernels', {}).items():
		if 'necessary_bytes_per_launch' in v:
			need[('step', k)] = v['necessary_bytes_per_launch']
	for v in r.get('stages', {}).values():
		if 'necessary_bytes_per_launch' in v:
			need[('step', v['kernel'])] = v['necessary_bytes_per_launch']
	for k, v in r.get('aperture_premade_cubes', {}).get('kernels', {}).items():
		if 'necessary_bytes_per_launch' in v:
			need[('premade', k)] = v['necessary_bytes_per_launch']

print("== HBM traffic per launch ==")
traffic = {'traffic_bytes_per_launch': {}, 'traffic_bytes_per_launch_premade': {}, 'detail': {}}
for k in sorted(means):
	v = means[k]
	if 'FETCH_SIZE' not in v or 'WRITE_SIZE' not in v:
		continue
	base = re.sub(r'<.*', '', k)
	# the fused kernel: <.., HAS_SUB = true, BKG = 1> is the timed step (raw cubes), <.., false, 0> the premade-cube leg
	leg = 'premade' if (base == 'tp_aperture_fused_kernel' and re.search(r'false,\s*0>', k)) else 'step'
	nb = need.get((leg, base))
	d = {'fetch_size_bytes_as_reported': v['FETCH_SIZE'], 'write_size_bytes': v['WRITE_SIZE'], 'necessary_bytes': nb}
	if nb:
		f = 1 if (v['FETCH_SIZE'] + v['WRITE_SIZE'] >= 0.97 * nb) else 2
		d['fetch_factor'] = f
		d['traffic_bytes'] = f * v['FETCH_SIZE'] + v['WRITE_SIZE']
		d['traffic_over_necessary'] = d['traffic_bytes'] / nb
		traffic['traffic_bytes_per_launch_premade' if leg == 'premade' else 'traffic_bytes_per_launch'][base] = d['traffic_bytes']
		print(f"{k:64s} FETCH {v['FETCH_SIZE']/1e9:8.3f} GB (x{f})  WRITE {v['WRITE_SIZE']/1e9:7.3f} GB  traffic {d['traffic_bytes']/1e9:8.3f} GB  necessary {nb/1e9:8.3f} GB  ratio {d['traffic_over_necessary']:.3f}")
	else:
		print(f"{k:64s} FETCH {v['FETCH_SIZE']/1e9:8.3f} GB (as reported, uncalibrated)  WRITE {v['WRITE_SIZE']/1e9:7.3f} GB")
	traffic['detail'][k] = d
if len(sys.argv) > 2:
	with open(sys.argv[2], 'w') as fh:
		json.dump(traffic, fh, indent=1, sort_keys=True)
