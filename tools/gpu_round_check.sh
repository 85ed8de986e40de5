#!/bin/bash
# Round check on the GPU box: smoke, GPU suite, default bench, profile passes (run through gpurun: bash tools/gpu_round_check.sh)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2c
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2c/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r2c/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q -rs > gpurun_out/r2c/gputests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2c/gputests.log
( time timeout -k 10 600 python bench.py > gpurun_out/r2c/bench1.json 2> gpurun_out/r2c/bench1.err ) 2>&1 | grep real
tail -3 gpurun_out/r2c/bench1.err; head -c 300 gpurun_out/r2c/bench1.json; echo
rm -rf gpurun_out/prof_r2c
timeout -k 10 900 bash profiles/run_profile.sh r2c > gpurun_out/r2c/profile.log 2>&1; tail -12 gpurun_out/r2c/profile.log
