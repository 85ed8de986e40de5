#!/bin/bash
# Round-2 re-entry check: GPU suite, default bench, profile passes.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2c
timeout -k 10 900 python -m pytest tests -m gpu -x -q -rs > gpurun_out/r2c/gputests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2c/gputests.log
( time timeout -k 10 600 python bench.py > gpurun_out/r2c/bench1.json 2> gpurun_out/r2c/bench1.err ) 2>&1 | grep real
tail -3 gpurun_out/r2c/bench1.err; head -c 800 gpurun_out/r2c/bench1.json; echo
timeout -k 10 900 bash profiles/run_profile.sh r2c > gpurun_out/r2c/profile.log 2>&1; tail -30 gpurun_out/r2c/profile.log
