#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC counters) into a small text table."""
import csv
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
