#!/bin/bash
mkdir -p gpurun_out
for t in "igemm_bk32=0" "igemm_bk32=4" "igemm_bk32=3"; do
  OPB_TUNE=$t python scripts/op_bench.py fwd dgrad 2>&1 | grep -E "conv3x3" | sed "s/^/$t /"
done | tee gpurun_out/occ.log
SVSR_TEST_TUNE=igemm_bk32=4 timeout 900 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -5
