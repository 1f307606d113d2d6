#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -k "stem" -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/stem2_tests.log 2>&1
echo "exit $?" >> gpurun_out/stem2_tests.log
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_lrs_model.py -m gpu -q --tb=short -p no:cacheprovider -x >> gpurun_out/stem2_tests.log 2>&1
echo "exit $?" >> gpurun_out/stem2_tests.log
grep -E "passed|failed|^FAILED|^ERROR|Error|assert|exit" gpurun_out/stem2_tests.log | tail -30
bash scripts/gpu_bench_ab.sh "stem_lds_bwd=1" "stem_lds_bwd=2"
