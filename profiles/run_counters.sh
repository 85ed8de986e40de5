#!/bin/bash
# Derived PMC metrics of the default bench, one rocprofv3 pass per metric (run on the GPU box through gpurun):
#   bash profiles/run_counters.sh r4
TAG=${1:-r4}
REPO=$(pwd)
OUT=$REPO/gpurun_out/counters_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for M in MeanOccupancyPerCU VALUBusy SALUBusy LdsBankConflict MemUnitStalled VALUUtilization; do
	rocprofv3 --pmc $M --output-format csv -d $OUT/$M -- python3 $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 --e2e-targets 0 --frames-targets 0 --frame 512 --psf-targets 0 --fullframe-frames 4 --linpsf-drift 0 > $OUT/$M.json 2> $OUT/$M.log
done
cd $REPO
python3 - "$OUT" <<'PY' > $OUT/summary.txt 2>&1
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
table = defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, '*/'))):
	metric = os.path.basename(os.path.dirname(d))
	for f in glob.glob(os.path.join(d, '**/*counter_collection.csv'), recursive=True):
		acc = defaultdict(list)
		with open(f) as fh:
			for r in csv.DictReader(fh):
				m = re.search(r'(tp_\w+)', r.get('Kernel_Name', ''))
				if m and r.get('Counter_Name') == metric:
					acc[m.group(1)].append(float(r['Counter_Value']))
		for k, v in acc.items():
			table[k][metric] = sum(v) / len(v)
metrics = sorted({m for v in table.values() for m in v})
print('%-30s' % 'kernel' + ''.join('%22s' % m for m in metrics))
for k in sorted(table):
	print('%-30s' % k + ''.join('%22.3f' % table[k].get(m, float('nan')) for m in metrics))
PY
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +4M -delete
