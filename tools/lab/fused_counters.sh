#!/bin/bash
# lab: PMC counters of the fused aperture kernel (tools/fused_sweep.py), one rocprofv3 pass per counter group
REPO=$(pwd)
OUT=$REPO/gpurun_out/fused_counters
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH" \
         "MeanOccupancyPerCU" "GRBM_GUI_ACTIVE"; do
	i=$((i+1))
	timeout 300 rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 $REPO/tools/fused_sweep.py > $OUT/g$i.log 2>&1
done
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, '**/*counter_collection.csv'), recursive=True):
	with open(f) as fh:
		for r in csv.DictReader(fh):
			m = re.search(r'(tp_aperture_fused_kernel|tp_bkg_stamp_kernel)', r.get('Kernel_Name', ''))
			if m: acc[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
	print(k)
	for c in sorted(acc[k]):
		v = acc[k][c]
		print('   %-28s %16.4g  (n=%d)' % (c, sum(v)/len(v), len(v)))
PY
find $OUT -name "*.csv" -size +2M -delete
