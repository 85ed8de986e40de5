#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2b
timeout 300 python -m pytest tests/test_gpu_comm.py -m gpu -x -q -rs > gpurun_out/r2b/comm.log 2>&1; tail -5 gpurun_out/r2b/comm.log
( time timeout 900 python bench.py > gpurun_out/r2b/bench1.json 2> gpurun_out/r2b/bench1.err ) 2>&1 | grep real
tail -3 gpurun_out/r2b/bench1.err; head -c 600 gpurun_out/r2b/bench1.json; echo
( time timeout 600 python bench.py --gpus 2 --steps 4 --warmup 1 --targets 4000 > gpurun_out/r2b/bench2.json 2> gpurun_out/r2b/bench2.err ) 2>&1 | grep real
tail -3 gpurun_out/r2b/bench2.err; head -c 400 gpurun_out/r2b/bench2.json; echo
