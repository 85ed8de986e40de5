#!/usr/bin/env python3
"""
Summarise rocprofv3 CSV output (kernel stats + PMC counters) of profiles/run_profile.sh into a small text table and a
traffic JSON (HBM bytes per launch of every kernel, read by bench.py).

FETCH_SIZE on gfx950 = 64 B x (read requests of the L2 to the fabric): rocprofv3 applies the gfx94x formula, whose count of
128-byte requests (TCC_BUBBLE) stays zero on this chip.  What a read request carries depends on the kernel
(tools/fetchcal.hip, tools/fetchcal2.sh, measured): pure read streams of any load width (4 / 8 / 16 bytes per lane, coalesced
or in 32-byte segments) issue ONE 128-byte request per missed line -- FETCH_SIZE reports exactly half their bytes --, a stream
that also writes at the same rate (the materialised subtraction) issues TWO 64-byte requests per line -- FETCH_SIZE is exact.
The factor is therefore measured per kernel from a third PMC pass: read lines = TCC_MISS - TCC_EA0_WRREQ (every L2 miss that
is not a write request fetches one 128-byte line; the write requests of these streaming kernels all miss), so

    read bytes = 128 x max(TCC_MISS - TCC_EA0_WRREQ, TCC_EA0_RDREQ / 2)        (between 64 and 128 bytes per read request)
    traffic    = read bytes + WRITE_SIZE

and the bytes per read request this implies are printed next to every kernel.  No byte count of the kernel enters.
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


# the step kernels launch by launch (the stats below average the warm-up launches of the run in: the first launches of a process run
# at lower clocks; bench.py's HIP-event average covers the timed steps only -- the last TIMED_STEPS launches of the trace)
TIMED = int(os.environ.get('TIMED_STEPS', 5))
for f in find('trace/**/*kernel_trace.csv'):
	with open(f) as fh:
		rows = list(csv.DictReader(fh))
	print("== the step kernels launch by launch (rocprofv3 --kernel-trace), ms ==")
	for name in ('tp_bkg_stamp_sum_kernel', 'tp_aperture_fused_kernel<2, true, true, 1, false>'):
		r = sorted((x for x in rows if name in x.get('Kernel_Name', '')), key=lambda x: int(x['Start_Timestamp']))
		d = [(int(x['End_Timestamp']) - int(x['Start_Timestamp'])) / 1e6 for x in r]
		if d:
			print(f"{name:52s} " + ' '.join(f'{v:.3f}' for v in d) + f"   average of the last {min(TIMED, len(d))} (the timed steps): {sum(d[-TIMED:]) / len(d[-TIMED:]):.3f}")
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find('trace/**/*kernel_stats.csv'):
	with open(f) as fh:
		for r in csv.DictReader(fh):
			if 'tp_' in r.get('Name', ''):
				print(f"{short(r['Name']):64s} calls={r.get('Calls'):>4s} avg_ns={r.get('AverageNs'):>12s} total_ns={r.get('TotalDurationNs'):>14s} pct={r.get('Percentage')}")

# the step kernels dispatch by dispatch (kernel trace): the --stats average above includes the untimed warm-up steps, whose
# launches are slower (the first ones of the process); bench.py's own HIP events cover the timed steps only -- the same launches
# as the last `steps` rows here
print("== the step's kernels, every launch in order (ms; rocprofv3 --kernel-trace), and the mean of the timed steps ==")
try:
	timed = int(json.load(open(os.path.join(out, 'bench_trace.json')))['steps'])
except Exception: # noqa: B902
	timed = 0
for f in find('trace/**/*kernel_trace.csv'):
	per = defaultdict(list)
	with open(f) as fh:
		for r in csv.DictReader(fh):
			n = short(r.get('Kernel_Name', ''))
			if n.startswith(('tp_bkg_stamp_sum_kernel', 'tp_aperture_fused_kernel<2, true, true, 1, false>')):
				per[n].append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
	for n, v in per.items():
		d = [x[1] for x in sorted(v)]
		tail = d[-timed:] if timed and len(d) >= timed else d
		print(f"{n:64s} {' '.join('%.3f' % x for x in d)}   mean of the last {len(tail)}: {sum(tail) / len(tail):.3f}")

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

# raw L2 counters of the calibration pass
lines = defaultdict(lambda: defaultdict(list))
for f in find('pmc_lines/**/*counter_collection.csv'):
	with open(f) as fh:
		for r in csv.DictReader(fh):
			if 'tp_' in r.get('Kernel_Name', ''):
				lines[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))

# necessary bytes per launch from the bench line of the trace run
need = {}
for f in ('bench_trace_legs.json', 'bench_trace.json'):
	# the full result of the traced run (bench_legs.json, copied by run_profile.sh); before round 5 the stdout line was the full result
	try:
		txt = open(os.path.join(out, f)).read().strip()
		r = json.loads(txt) if f.endswith('_legs.json') else json.loads(txt.splitlines()[-1])
	except Exception: # noqa: B902
		continue
	if 'kernels' not in r:
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
	lp = r.get('linpsf', {}).get('roofline', {}).get('hbm', {})
	if lp:
		need[('step', 'tp_linpsf_fit')] = lp['necessary_bytes_per_step']

print("== HBM traffic per launch (read bytes = 128 x (L2 misses - write requests), see the header) ==")
traffic = {'traffic_bytes_per_launch': {}, 'traffic_bytes_per_launch_premade': {}, 'detail': {}, 'fullframe_frames': int(os.environ.get('FULLFRAME_FRAMES', '0'))}
linpsf_total = 0.0
for k in sorted(means):
	v = means[k]
	if 'FETCH_SIZE' not in v or 'WRITE_SIZE' not in v:
		continue
	base = re.sub(r'<.*', '', k)
	# the fused kernel: <.., HAS_SUB = true, BKG = 1> is the timed step (raw cubes), <.., false, 0> the premade-cube leg
	# the fused kernel <VEC, VEC4, HAS_SUB, BKG, A1>: <.., true, 1, false> is the timed step (raw cubes, the sum image given), <.., false, 0, true> the premade-cube leg
	leg = 'premade' if (base == 'tp_aperture_fused_kernel' and re.search(r'false,\s*0,\s*true>', k)) else 'step'
	nb = need.get((leg, base))
	d = {'fetch_size_bytes_as_reported': v['FETCH_SIZE'], 'write_size_bytes': v['WRITE_SIZE'], 'necessary_bytes': nb}
	ln = {c: sum(x) / len(x) for c, x in lines.get(k, {}).items()}
	if 'TCC_MISS_sum' in ln and 'TCC_EA0_RDREQ_sum' in ln and ln['TCC_EA0_RDREQ_sum'] > 0:
		read_lines = max(ln['TCC_MISS_sum'] - ln.get('TCC_EA0_WRREQ_sum', 0.0), ln['TCC_EA0_RDREQ_sum'] / 2)
		read_bytes = min(128.0 * read_lines, 128.0 * ln['TCC_EA0_RDREQ_sum'])
		d['bytes_per_read_request'] = read_bytes / ln['TCC_EA0_RDREQ_sum']
		d['fetch_factor'] = d['bytes_per_read_request'] / 64.0
		d['traffic_bytes'] = read_bytes + v['WRITE_SIZE']
		d['raw'] = ln
		txt = f"{k:64s} read {read_bytes/1e9:8.3f} GB ({d['bytes_per_read_request']:5.1f} B/request; FETCH_SIZE says {v['FETCH_SIZE']/1e9:7.3f})  WRITE {v['WRITE_SIZE']/1e9:7.3f} GB  traffic {d['traffic_bytes']/1e9:8.3f} GB"
		if nb:
			d['traffic_over_necessary'] = d['traffic_bytes'] / nb
			txt += f"  necessary {nb/1e9:8.3f} GB  ratio {d['traffic_over_necessary']:.3f}"
		print(txt)
		traffic['traffic_bytes_per_launch_premade' if leg == 'premade' else 'traffic_bytes_per_launch'][base if '<' not in k or base != 'tp_linpsf_fit2_kernel' else k] = d['traffic_bytes']
		if base in ('tp_linpsf_fit2_kernel', 'tp_linpsf_plan_kernel', 'tp_linpsf_coef_kernel', 'tp_linpsf_fitm_kernel'):
			linpsf_total += d['traffic_bytes']
	else:
		print(f"{k:64s} FETCH {v['FETCH_SIZE']/1e9:8.3f} GB (as reported, no calibration pass)  WRITE {v['WRITE_SIZE']/1e9:7.3f} GB")
	traffic['detail'][k] = d
if linpsf_total:
	nb = need.get(('step', 'tp_linpsf_fit'))
	traffic['traffic_bytes_per_launch']['tp_linpsf_fit'] = linpsf_total
	print(f"{'LinPSF fit (plan + coefficient store + every fit launch)':64s} traffic {linpsf_total/1e9:8.3f} GB per step" + (f"  necessary {nb/1e9:8.3f} GB  ratio {linpsf_total/nb:.3f}" if nb else ''))
# B1 branch by branch: every dispatch of tools/radial_time.py MODE=plain / tess (two runs of NF frames), summed
b1_frames = int(os.environ.get('B1_FRAMES', '0'))
if b1_frames:
	traffic['b1_traffic_bytes_per_frame'] = {}
	traffic['b1_frames'] = b1_frames
	print("== B1 (fit_background) traffic per 2048 x 2048 frame, every dispatch of the branch summed ==")
	for mode in ('plain', 'tess'):
		written = 0.0
		for f in find(f'b1_{mode}_write/**/*counter_collection.csv'):
			with open(f) as fh:
				for r in csv.DictReader(fh):
					if r.get('Counter_Name') == 'WRITE_SIZE' and 'tp_' in r.get('Kernel_Name', ''):
						written += float(r['Counter_Value']) * 1024.0
		disp = defaultdict(dict)
		for f in find(f'b1_{mode}_lines/**/*counter_collection.csv'):
			with open(f) as fh:
				for r in csv.DictReader(fh):
					if 'tp_' in r.get('Kernel_Name', ''):
						disp[r.get('Dispatch_Id')][r['Counter_Name']] = float(r['Counter_Value'])
		read = 0.0
		for c in disp.values():
			if 'TCC_MISS_sum' in c and 'TCC_EA0_RDREQ_sum' in c:
				read += min(128.0 * max(c['TCC_MISS_sum'] - c.get('TCC_EA0_WRREQ_sum', 0.0), c['TCC_EA0_RDREQ_sum'] / 2), 128.0 * c['TCC_EA0_RDREQ_sum'])
		if disp:
			traffic['b1_traffic_bytes_per_frame'][mode] = (read + written) / b1_frames
			print(f"{mode:6s} read {read / b1_frames / 1e6:8.1f} MB  written {written / b1_frames / 1e6:8.1f} MB  traffic {(read + written) / b1_frames / 1e6:8.1f} MB per frame ({len(disp)} dispatches)")
if len(sys.argv) > 2:
	with open(sys.argv[2], 'w') as fh:
		json.dump(traffic, fh, indent=1, sort_keys=True)
