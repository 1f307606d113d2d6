#!/bin/bash
# A/B of tuning knobs on the per-op microbenchmarks: gpu_ab.sh "<op_bench sections>" "<knob=value>" "<knob=value>" ...
mkdir -p gpurun_out
secs=$1; shift
for t in "$@"; do
  OPB_TUNE=$t python scripts/op_bench.py $secs 2>&1 | grep -vE "^==|amdgpu.ids" | sed "s/^/$t /"
done | tee gpurun_out/ab.log
