#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3h
python -m syncvsr_amd.build > /dev/null 2>&1
timeout 600 python scripts/probes/lin_bench.py default igemm_ksplit=0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3h/lin.log
