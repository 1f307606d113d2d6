#!/bin/bash
# rocprofv3 kernel trace of a few eager steps -> gpurun_out/prof/*.csv (kernel stats)
mkdir -p gpurun_out/prof
export PYTHONUNBUFFERED=1
# per-kernel numbers are taken with every launch in line (no side-stream overlap), like bench.py's roofline leg
export SVSR_SIDE_TRUNK=0 SVSR_SIDE_ENCODER=0
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o ${1:-r1} -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --profile-steps 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_run.log 2>&1
echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/prof | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -45 "$f" | cut -c1-200
find gpurun_out/prof -name "*kernel_trace.csv" -size +30M -delete
tail -2 gpurun_out/prof_run.log | cut -c1-300
