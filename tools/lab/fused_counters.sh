#!/bin/bash
# PMC counters of the step's kernels (one rocprofv3 pass per group), run on the GPU box:  bash tools/lab/fused_counters.sh
REPO=$(pwd)
OUT=$REPO/gpurun_out/fused_counters
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_ANY" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR"; do
	i=$((i+1))
	STEPS=3 rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 $REPO/tools/step_time.py > $OUT/g$i.log 2>&1
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
			m = re.search(r'(tp_aperture_fused_kernel|tp_bkg_stamp_sum_kernel)', r.get('Kernel_Name', ''))
			if m:
				table[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(table):
	print(k)
	for c in sorted(table[k]):
		v = table[k][c]
		print('   %-28s launches %3d  mean per launch %.4g' % (c, len(v), sum(v) / len(v)))
PY
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +4M -delete
