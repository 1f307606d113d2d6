#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -s ${1:-} > gpurun_out/model_tests.log 2>&1
echo "exit $?" >> gpurun_out/model_tests.log
grep -E "passed|failed|^FAILED|^ERROR|worst encoder" gpurun_out/model_tests.log | tail -20
