#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC counters) into a small text table."""
import csv
import re
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
	return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find('trace/**/*kernel_stats.csv'):
	with open(f) as fh:
		rows = list(csv.DictReader(fh))
	for r in rows:
		name = r.get('Name', '')
		if 'tp_' not in name:
			continue
		print(f"{name[:70]:70s} calls={r.get('Calls')} avg_ns={r.get('AverageNs')} total_ns={r.get('TotalDurationNs')} pct={r.get('Percentage')}")

means = {}
for counter, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
	print(f"== {counter} per dispatch (KiB as reported; FETCH_SIZE reads 1/2 of the bytes of wide streaming reads on gfx950) ==")
	acc = defaultdict(list)
	for f in find(f'{sub}/**/*counter_collection.csv'):
		with open(f) as fh:
			for r in csv.DictReader(fh):
				if r.get('Counter_Name') == counter and 'tp_' in r.get('Kernel_Name', ''):
					acc[r['Kernel_Name']].append(float(r['Counter_Value']))
	for k, v in acc.items():
		print(f"{k[:70]:70s} dispatches={len(v)} mean={sum(v)/len(v):.1f} min={min(v):.1f} max={max(v):.1f}")
		means.setdefault(re.search(r'(tp_\w+)', k).group(1), {})[counter] = sum(v) / len(v)

# HBM bytes per launch: FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests
# of wide streaming reads at 64 bytes, so it is doubled (MI355X_MICROARCH.md, "HBM"); WRITE_SIZE is exact.
if len(sys.argv) > 2:
	import json
	traffic = {}
	for k, v in means.items():
		if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
			traffic[k] = 2.0 * v['FETCH_SIZE'] * 1024.0 + v['WRITE_SIZE'] * 1024.0
	with open(sys.argv[2], 'w') as fh:
		json.dump(traffic, fh, indent=1, sort_keys=True)
	print("== HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) ==")
	for k, v in sorted(traffic.items()):
		print(f"{k:40s} {v/1e9:.3f} GB")
