#!/bin/bash
# round-4 baseline on this round's box: timeline of one step, quick bench, whole gpu suite
cd $GRAFT_REPO_ROOT
bash scripts/gpu_r4_trace.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err; echo "bench $?"
tail -c 1500 gpurun_out/r4a/bench.json
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r4a/tests.log
