#!/bin/bash
# PMC counters of the LinPSF fit kernels (one rocprofv3 pass per group), run on the GPU box:  bash tools/lab/linpsf_counters.sh
REPO=$(pwd)
OUT=$REPO/gpurun_out/linpsf_counters
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
i=0
for G in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_ANY"; do
	i=$((i+1))
	NT=10000 rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 $REPO/tools/linpsf_time.py > $OUT/g$i.log 2>&1
	echo "group $i done: $G"
done
cd $REPO
python3 - "$OUT" <<'PY' > $OUT/summary.txt 2>&1
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
table = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, 'g*/**/*counter_collection.csv'), recursive=True):
	with open(f) as fh:
		for r in csv.DictReader(fh):
			m = re.search(r'(tp_linpsf\w+)(<[^>]*>)?', r.get('Kernel_Name', ''))
			if m:
				table[m.group(1) + (m.group(2) or '')][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(table):
	print(k)
	for c in sorted(table[k]):
		v = table[k][c]
		print('   %-34s n=%3d mean %.4g' % (c, len(v), sum(v) / len(v)))
PY
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +4M -delete
grep -c . $OUT/avail.txt
