#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/probes/p8_check.py --ablate 2>&1 | tail -12 | tee $OUT/p8_ablate.log
