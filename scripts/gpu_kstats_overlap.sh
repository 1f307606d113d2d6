#!/bin/bash
# per-kernel durations of the DEFAULT step (side stream on): which main-stream launches stretch under the side stream's kernels
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstats_ov
rm -rf $OUT; mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lrw -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-lrs-leg --profile-steps 0 --steps 8 --warmup 2 --enqueue eager > $OUT/lrw_run.log 2>&1; echo "stats $?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os, re
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "kstats_ov")
f = glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:28]:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:50]
    print(f"{int(r['Calls']):5d} calls avg {float(r['AverageNs'])/1e3:7.1f} min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f}  {name}")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
