#!/bin/bash
# usage: gpu_one.sh <test file(s) / pytest args>
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest "$@" -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/one_tests.log 2>&1
echo "exit $?" >> gpurun_out/one_tests.log
grep -E "passed|failed|^FAILED|^ERROR|Error|assert|run[0-9]:|end to end" gpurun_out/one_tests.log | tail -40
