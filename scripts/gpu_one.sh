#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/one
timeout 900 python -m pytest "$@" -x -q -m gpu 2>&1 | grep -v "^  \|^ \"" | tail -40 | tee gpurun_out/one/log.txt
