#!/bin/bash
# fused BatchNorm-backward epilogue: kernel test, model parity, whole-step A/B
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -k "dgrad_with_bn or conv_dgrad or conv_fwd or bn_act" -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/bnfuse_tests.log 2>&1
echo "exit $?" >> gpurun_out/bnfuse_tests.log
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py -m gpu -q --tb=short -p no:cacheprovider -x >> gpurun_out/bnfuse_tests.log 2>&1
echo "exit $?" >> gpurun_out/bnfuse_tests.log
grep -E "passed|failed|^FAILED|^ERROR|Error|assert|exit" gpurun_out/bnfuse_tests.log | tail -30
bash scripts/gpu_bench_ab.sh "bn_bwd_fused=0" "bn_bwd_fused=1" "bn_bwd_fused=1"
