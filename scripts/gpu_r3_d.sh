#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3d
timeout 600 python scripts/probes/native_debug.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3d/native_debug.log
