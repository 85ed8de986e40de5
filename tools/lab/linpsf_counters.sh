#!/bin/bash
# lab: PMC counters of the LinPSF kernels (tools/linpsf_time.py), one rocprofv3 pass per counter group
REPO=$(pwd)
OUT=$REPO/gpurun_out/linpsf_counters
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAIT_INST_LDS" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" \
         "MeanOccupancyPerCU" "VALUBusy"; do
	i=$((i+1))
	timeout 300 rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 $REPO/tools/linpsf_time.py > $OUT/g$i.log 2>&1
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
			m = re.search(r'(tp_linpsf_\w+(<[\d, ]+>)?)', r.get('Kernel_Name', ''))
			if m: acc[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
	print(k)
	for c in sorted(acc[k]):
		v = acc[k][c]
		print('   %-28s %16.1f  (n=%d)' % (c, sum(v)/len(v), len(v)))
PY
find $OUT -name "*.csv" -size +2M -delete
