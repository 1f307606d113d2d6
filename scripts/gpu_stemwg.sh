#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_shapes.py -k "stem" -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/stemwg_tests.log 2>&1
echo "exit $?" >> gpurun_out/stemwg_tests.log
grep -E "passed|failed|^FAILED|^ERROR|Error|assert|exit" gpurun_out/stemwg_tests.log | tail -10
bash scripts/gpu_bench_ab.sh "stem_wg_pipe=0" "stem_wg_pipe=1"
