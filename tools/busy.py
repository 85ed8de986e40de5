#!/usr/bin/env python3
"""GPU-busy analysis of a rocprofv3 kernel trace database (rocpd sqlite): per kernel totals and the union of the kernel intervals
inside the last `frac` of the traced time (how much of the wall time some kernel was running).
    python3 tools/busy.py results.db [frac = 0.25] [n: timeline of the last n kernels, 0 = none] [1: concurrency histogram]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
rows = list(db.execute("select name, start, end from kernels order by start"))
t0, t1 = rows[0][1], max(r[2] for r in rows)
lo = t1 - (t1 - t0) * frac
sel = [r for r in rows if r[1] >= lo]
busy, cur_s, cur_e = 0, None, None
for _, s, e in sel:
	if cur_e is None or s > cur_e:
		if cur_e is not None: busy += cur_e - cur_s
		cur_s, cur_e = s, e
	else:
		cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(r[2] for r in sel) - sel[0][1]
print(f'window {span/1e6:.2f} ms, some kernel running {busy/1e6:.2f} ms = {busy/span:.3f}; kernels {len(sel)}')
tot = {}
for n, s, e in sel:
	k = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0][-48:]
	a = tot.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
for k, (n, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:18]:
	print(f'  {k:50s} {n:6d}  {d/1e6:9.3f} ms  ({d/span:.3f} of window)')
if len(sys.argv) > 3 and int(sys.argv[3]) > 0:   # timeline of the last `n` kernels: start offset (us), duration (us), stream, name
	n = int(sys.argv[3])
	rows2 = list(db.execute("select name, start, end, stream_id, grid_x, workgroup_x from kernels order by start"))[-n:]
	b = rows2[0][1]
	for nm, s, e, st, gx, wx in rows2:
		k = nm.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0][-40:]
		print(f'{(s - b)/1e3:9.1f} {(e - s)/1e3:8.1f}  s{st} {gx // max(wx, 1):7d} wg  {k}')
if len(sys.argv) > 4:   # concurrency: share of the window with k kernels running; with a large (>= 1500 workgroups) kernel running
	ev = []
	for nm, s, e, st, gx, wx in db.execute("select name, start, end, stream_id, grid_x, workgroup_x from kernels where start >= ?", (lo,)):
		big = 1 if gx // max(wx, 1) >= 1500 else 0
		ev.append((s, 1, big)); ev.append((e, -1, -big))
	ev.sort()
	hist, bigt, cur, curb, last = {}, 0, 0, 0, ev[0][0]
	for t, d, b in ev:
		hist[cur] = hist.get(cur, 0) + (t - last)
		if curb > 0: bigt += t - last
		cur += d; curb += b; last = t
	tot = sum(hist.values())
	print('kernels running at once:', {k: round(v / tot, 3) for k, v in sorted(hist.items())})
	print('a large kernel running:', round(bigt / tot, 3))
