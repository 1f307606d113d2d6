#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -k "bn_backward" -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/swish_tests.log 2>&1
echo "exit $?" >> gpurun_out/swish_tests.log
timeout 1500 python -m pytest tests/test_gpu_lrs_model.py tests/test_gpu_model.py tests/test_gpu_train.py -m gpu -q --tb=short -p no:cacheprovider >> gpurun_out/swish_tests.log 2>&1
echo "exit $?" >> gpurun_out/swish_tests.log
grep -E "passed|failed|^FAILED|^ERROR|Error|assert|exit" gpurun_out/swish_tests.log | tail -20
bash scripts/gpu_lrs_ab.sh "bn_bwd_fused=0" "bn_bwd_fused=1"
bash scripts/gpu_bench_ab.sh "bn_bwd_fused=1"
