#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3j
python -m syncvsr_amd.build > /dev/null 2>&1
for i in 1 2; do
echo "== default";  timeout 300 python scripts/probes/graph_rccl_debug.py 2>&1 | grep "plain\|graph\|eager\|Error\|error" | cut -c1-300 | tee gpurun_out/r3j/dbg.log
done
