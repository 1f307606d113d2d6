#!/bin/bash
# Runs on the GPU box: parity tests, bench (graph + eager), rocprofv3 kernel stats.  Logs under gpurun_out/.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests.log 2>&1
echo "tests exit $?" >> gpurun_out/tests.log
tail -n 15 gpurun_out/tests.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err
echo "bench graph exit $?"; tail -n 3 gpurun_out/bench_graph.err; cat gpurun_out/bench_graph.json
timeout 600 python bench.py --steps 10 --warmup 2 --no-graph --no-cpu-baseline > gpurun_out/bench_eager.json 2> gpurun_out/bench_eager.err
echo "bench eager exit $?"; tail -n 3 gpurun_out/bench_eager.err; cat gpurun_out/bench_eager.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --profile-steps 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_run.log 2>&1
echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*stats*" | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f"
# keep only the small summaries
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
