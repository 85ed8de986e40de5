#!/bin/bash
# lab: per-class time of the matrix-core LinPSF fit for the library variants of tools/lab/build_variants.sh (rocprofv3 kernel stats).
# The variants need the lab hooks, which are NOT in the product source: git apply tools/lab/linpsf_lab_hooks.patch, then e.g.
#   SRC=linpsf_mfma.hip bash tools/lab/build_variants.sh base: noload:-DTP_LAB_NOLOAD nomfma:-DTP_LAB_NOMFMA noacc:-DTP_LAB_NOACC stamp:-DTP_LAB_STAMP
# and git checkout photometry_amd/csrc/linpsf_mfma.hip afterwards.
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for lib in $R/tools/lab/lib_*.so; do
	v=$(basename $lib .so)
	LINPSF_PATH=1 TP_LAB_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lv_$v -- python3 $R/tools/linpsf_time.py > $R/gpurun_out/lv_$v.log 2>&1
	f=$(ls -t $(find $R/gpurun_out/lv_$v -name "*kernel_stats.csv") | head -1)
	echo "== $v"; grep -E "fitm" $f | sed -e 's/(tp_linpsf::FitArgs.*)"//' -e 's/.*fitm_kernel//' | cut -d, -f1-4
done
