#!/bin/bash
mkdir -p gpurun_out
for t in 512 640 768 1024 1536 3072; do
  OPB_TUNE=stem_fwd_blocks=$t python scripts/op_bench.py stem 2>&1 | grep -E "stem conv fwd" | sed "s/^/blocks=$t /"
done | tee gpurun_out/occ.log
