#!/bin/bash
mkdir -p gpurun_out
for pad in 0 16000 30000 60000; do
  OPB_TUNE=igemm_lds_pad=$pad python scripts/op_bench.py fwd dgrad 2>&1 | grep -E "conv3x3" | sed "s/^/pad=$pad /"
done | tee gpurun_out/occ.log
