#!/bin/bash
mkdir -p gpurun_out
for t in "wg_short_k=0" "wg_short_k=16"; do
  OPB_TUNE=$t python scripts/op_bench.py linear 2>&1 | grep -E "wgrad" | sed "s/^/$t /"
done | tee gpurun_out/occ.log
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "linear or wgrad or LINEAR" 2>&1 | tail -3
