#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_shapes.py -m gpu -q --tb=short -p no:cacheprovider -k "stem" > gpurun_out/stem_tests.log 2>&1
tail -3 gpurun_out/stem_tests.log
python scripts/op_bench.py stem 2>&1 | grep -i stem | tee gpurun_out/stem_bench.log
