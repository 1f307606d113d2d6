#!/bin/bash
# A/B of tuning knobs on the op micro-benchmarks:  gpu_ab.sh "<which ops>" "<tune A>" "<tune B>" ...
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
which=$1; shift
for t in "$@"; do
  echo "=== OPB_TUNE=$t"
  OPB_TUNE=$t timeout 600 python scripts/op_bench.py $which 2>&1 | grep -v amdgpu.ids
done
