#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3m
for v in 0 1; do
SVSR_STEM_KEEP_WINNERS=$v timeout 900 python -m pytest tests/test_gpu_blockwise.py -q -m gpu -s > gpurun_out/r3m/log$v.txt 2>&1
echo "== keep winners $v"; grep "stem3d" gpurun_out/r3m/log$v.txt | grep "cos" | head -8; tail -1 gpurun_out/r3m/log$v.txt
done
