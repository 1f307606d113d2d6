#!/bin/bash
# rocprofv3 kernel trace of a few eager LRS steps -> gpurun_out/prof_lrs/*.csv (kernel stats)
mkdir -p gpurun_out/prof_lrs
export PYTHONUNBUFFERED=1
# per-kernel numbers are taken with every launch in line (no side-stream overlap), like bench.py's roofline leg
export SVSR_SIDE_TRUNK=0 SVSR_SIDE_ENCODER=0
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lrs -o ${1:-lrs} -- python $GRAFT_REPO_ROOT/bench.py --workload lrs --steps 4 --warmup 2 --no-graph --no-cpu-baseline --profile-steps 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_lrs_run.log 2>&1
echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_lrs -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -48 "$f" | cut -c1-170
find gpurun_out/prof_lrs -name "*kernel_trace.csv" -size +30M -delete
tail -1 gpurun_out/prof_lrs_run.log | cut -c1-200
