#!/bin/bash
# rocprofv3 kernel trace of a few eager steps (all launches in line: no side-stream overlap) -> gpurun_out/prof/*_kernel_stats.csv
mkdir -p gpurun_out/prof
export PYTHONUNBUFFERED=1
export SVSR_SIDE_TRUNK=0 SVSR_SIDE_ENCODER=0
tag=${1:-r2}
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o $tag -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_run.log 2>&1
echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof -name "${tag}_kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-160
find gpurun_out/prof -name "*kernel_trace.csv" -size +30M -delete
tail -1 gpurun_out/prof_run.log | cut -c1-400
