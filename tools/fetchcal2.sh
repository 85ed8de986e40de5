#!/bin/bash
# raw L2 <-> fabric counters of tools/fetchcal and of the bench kernels: read requests, "bubbles" (128-byte requests in the
# gfx94x formula of FETCH_SIZE), L2 misses -- to decide per kernel what one read request carries
REPO=$(pwd)
OUT=$REPO/gpurun_out/fetchcal2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_BUBBLE_sum TCC_MISS_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/cal -- $REPO/tools/fetchcal > $OUT/cal.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_BUBBLE_sum TCC_MISS_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/bench -- python3 $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 --e2e-targets 0 --frames-targets 0 --frame 0 --psf-targets 0 --fullframe-frames 0 > $OUT/bench.json 2> $OUT/bench.log
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
for sub in ('cal', 'bench'):
	acc = defaultdict(lambda: defaultdict(list))
	for f in glob.glob(os.path.join(out, sub, '**/*counter_collection.csv'), recursive=True):
		for r in csv.DictReader(open(f)):
			name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
			if sub == 'bench' and 'tp_' not in name:
				continue
			acc[name[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
	for k in sorted(acc):
		v = {c: sum(x) / len(x) for c, x in acc[k].items()}
		rd = v.get('TCC_EA0_RDREQ_sum', 0)
		if rd < 1e5:
			continue
		print('%-60s RDREQ %.4g  BUBBLE %.4g  MISS %.4g  RDREQ_32B %.4g  | RDREQ/MISS %.3f  MISS*128 = %.3f GB  RDREQ*64 = %.3f GB' % (k, rd, v.get('TCC_BUBBLE_sum', 0),
			v.get('TCC_MISS_sum', 0), v.get('TCC_EA0_RDREQ_32B_sum', 0), rd / max(v.get('TCC_MISS_sum', 1), 1), v.get('TCC_MISS_sum', 0) * 128 / 1e9, rd * 64 / 1e9))
PY
find $OUT -name "*.csv" -size +4M -delete
