#!/bin/bash
# Profiles of the default bench (run on the GPU box through gpurun):
#   bash profiles/run_profile.sh r4
# 1. rocprofv3 kernel trace + stats (per-kernel durations) of the default bench command
# 2./3. PMC passes for the HBM traffic of every dispatch (FETCH_SIZE and WRITE_SIZE need separate passes; never together with a trace)
# 4. a PMC pass with the raw L2 counters that calibrate FETCH_SIZE per kernel
TAG=${1:-r4}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --cpu-sample 0 --e2e-targets 0 --frames-targets 0 --frame 512 --psf-targets 0 --fullframe-frames 0 --linpsf-drift 0"
# (the PMC passes include FF full frames of the full-frame background / pixel-flag leg, so that its kernels -- mesh, zoom, radial,
# median filter -- get their traffic too; summarize.py divides by FF)
FF=4
PMCARGS="--steps 2 --warmup 1 --cpu-sample 0 --e2e-targets 0 --frames-targets 0 --frame 512 --psf-targets 0 --fullframe-frames $FF --linpsf-drift 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
cp $REPO/bench_legs.json $OUT/bench_trace_legs.json   # the full result of that run (stdout carries the short line only)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $PMCARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $PMCARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.log
# 4. what one read request carries: L2 misses, write requests and read requests of every dispatch (see profiles/summarize.py)
rocprofv3 --pmc TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc_lines -- python3 $REPO/bench.py $PMCARGS > $OUT/bench_pmc_lines.json 2> $OUT/pmc_lines.log
# 5. B1 alone, branch by branch (tools/radial_time.py MODE=...: two runs of NF frames each, every dispatch of the process belongs to
# the branch): total traffic per frame = sum over all dispatches / (2 NF)
NF=4
for MODE in plain tess; do
	MODE=$MODE NF=$NF rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/b1_${MODE}_write -- python3 $REPO/tools/radial_time.py > $OUT/b1_${MODE}_write.log 2>&1
	MODE=$MODE NF=$NF rocprofv3 --pmc TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/b1_${MODE}_lines -- python3 $REPO/tools/radial_time.py > $OUT/b1_${MODE}_lines.log 2>&1
done
cd $REPO
B1_FRAMES=$((2 * NF)) FULLFRAME_FRAMES=$FF python3 profiles/summarize.py $OUT $OUT/traffic.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only small files in the merged output
find $OUT -name "*.csv" -size +8M -delete
