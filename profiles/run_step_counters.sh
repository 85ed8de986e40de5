#!/bin/bash
# Raw SQ counters of the two kernels of the configs[2] step (tools/step_time.py: tp_bkg_stamp_sum_kernel and
# tp_aperture_fused_kernel), one rocprofv3 pass per counter group (PMC passes only -- never together with a trace):
#   bash profiles/run_step_counters.sh r6      (on the GPU box, through gpurun)
TAG=${1:-r6}
REPO=$(pwd)
OUT=$REPO/gpurun_out/step_counters_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE" "VALUBusy" "MeanOccupancyPerCU" "MemUnitStalled"; do
	i=$((i+1))
	STEPS=4 rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 $REPO/tools/step_time.py > $OUT/g$i.log 2>&1
done
cd $REPO
python3 - "$OUT" <<'PY' > $OUT/summary.txt
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, '**/*counter_collection.csv'), recursive=True):
	with open(f) as fh:
		for r in csv.DictReader(fh):
			m = re.search(r'(tp_\w+)', r.get('Kernel_Name', ''))
			if m: acc[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
print('# raw counters per launch of the configs[2] step kernels (tools/step_time.py: 10 000 targets x 1300 x 15x15), rocprofv3 --pmc, one pass per group')
for k in sorted(acc):
	if k not in ('tp_bkg_stamp_sum_kernel', 'tp_aperture_fused_kernel'): continue
	print(k)
	c = {n: sum(v) / len(v) for n, v in acc[k].items()}
	for n in sorted(c):
		print('   %-28s launches %3d  mean per launch %.4g' % (n, len(acc[k][n]), c[n]))
	if 'SQ_INSTS_VALU' in c and 'SQ_WAVES' in c:
		print('   -> vector instructions per wavefront %.0f' % (c['SQ_INSTS_VALU'] / c['SQ_WAVES']))
	if 'SQ_ACTIVE_INST_VALU' in c and 'SQ_INSTS_VALU' in c:
		print('   -> SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU %.3f (x 4 cycles: issue cycles per vector instruction)' % (c['SQ_ACTIVE_INST_VALU'] / c['SQ_INSTS_VALU']))
	if 'SQ_WAIT_INST_ANY' in c and 'SQ_WAVE_CYCLES' in c:
		print('   -> share of wavefront cycles spent waiting for an instruction %.3f' % (c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']))
PY
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
