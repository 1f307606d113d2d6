#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_xt.py -m gpu -q --tb=short -p no:cacheprovider -s ${1:-} > gpurun_out/xt_tests.log 2>&1
echo "exit $?" >> gpurun_out/xt_tests.log
grep -E "passed|failed|^FAILED|^ERROR|worst|Error|assert" gpurun_out/xt_tests.log | tail -40
