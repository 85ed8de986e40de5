#!/bin/bash
# HBM traffic of the batched drop-in entry's kernels (run on the GPU box through gpurun):  bash profiles/frames_traffic.sh
# Two PMC passes (WRITE_SIZE; the raw L2 counters that calibrate the read bytes, as profiles/summarize.py does) over
# tools/frames_pipe.py with one job in flight: REPS runs of ONE batch of 2 500 targets on a 512 x 512 x 1300 region.
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_frames_traffic
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export IN_FLIGHT=1 BATCHES=1 REPS=3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/tools/frames_pipe.py > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/lines -- python3 $REPO/tools/frames_pipe.py > $OUT/lines.log 2>&1
cd $REPO
python3 - $OUT $REPS <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
out, reps = sys.argv[1], int(sys.argv[2])
def short(n):
	return re.sub(r'\(anonymous namespace\)::|void ', '', n).split('(')[0].split('<')[0]
written, nw = defaultdict(float), defaultdict(int)
for f in glob.glob(os.path.join(out, 'write/**/*counter_collection.csv'), recursive=True):
	for r in csv.DictReader(open(f)):
		if r.get('Counter_Name') == 'WRITE_SIZE':
			written[short(r['Kernel_Name'])] += float(r['Counter_Value']) * 1024.0
			nw[short(r['Kernel_Name'])] += 1
disp = defaultdict(dict)
for f in glob.glob(os.path.join(out, 'lines/**/*counter_collection.csv'), recursive=True):
	for r in csv.DictReader(open(f)):
		d = disp[(f, r.get('Dispatch_Id'))]
		d['k'] = short(r['Kernel_Name']); d[r['Counter_Name']] = float(r['Counter_Value'])
read = defaultdict(float)
for c in disp.values():
	if 'TCC_MISS_sum' in c and 'TCC_EA0_RDREQ_sum' in c:
		read[c['k']] += min(128.0 * max(c['TCC_MISS_sum'] - c.get('TCC_EA0_WRREQ_sum', 0.0), c['TCC_EA0_RDREQ_sum'] / 2), 128.0 * c['TCC_EA0_RDREQ_sum'])
print(f'# HBM traffic per batch of 2 500 targets (512 x 512 x 1300 region, 15 x 15 default stamps, every round of the batch), mean of {reps} batches')
print(f'# read: 128 B x max(L2 misses - write requests, read requests / 2), capped at 128 B x read requests (profiles/summarize.py); written: WRITE_SIZE')
tot_r = tot_w = 0.0
for k in sorted(set(read) | set(written), key=lambda k: -(read[k] + written[k])):
	print(f'{k:34s} launches {nw[k] // max(reps, 1):4d}  read {read[k] / reps / 1e9:7.3f} GB  written {written[k] / reps / 1e9:7.3f} GB')
	tot_r += read[k] / reps; tot_w += written[k] / reps
print(f'{"all kernels":34s}                read {tot_r / 1e9:7.3f} GB  written {tot_w / 1e9:7.3f} GB  = {(tot_r + tot_w) / 1e9:.2f} GB per batch')
PY
