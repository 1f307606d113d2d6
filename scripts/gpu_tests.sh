#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-level and whole-model parity tests, logs under gpurun_out/.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
rocminfo 2>/dev/null | grep -m1 -E "gfx9" > gpurun_out/gpu.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/kernels.log 2>&1
echo "kernels exit $?" >> gpurun_out/kernels.log
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -s -p no:cacheprovider > gpurun_out/model.log 2>&1
echo "model exit $?" >> gpurun_out/model.log
tail -n 60 gpurun_out/kernels.log
tail -n 80 gpurun_out/model.log
