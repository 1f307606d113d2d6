#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3d
timeout 1500 python scripts/probes/emu_parity.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3d/emu.log
