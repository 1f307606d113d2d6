#!/bin/bash
# quick per-kernel durations of the in-line step (rocprofv3 --stats), top N rows
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstats
rm -rf $OUT; mkdir -p $OUT
export PYTHONUNBUFFERED=1 SVSR_SIDE_TRUNK=0 SVSR_SIDE_ENCODER=0
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --profile-steps 0 --steps 5 --warmup 2 $@ > $OUT/run.log 2>&1; echo "stats $?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os, re
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "kstats")
f = glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel sum {tot/7e6:.3f} ms/step")
for r in rows[:50]:
    nm = re.sub(r"\(.*", "", r["Name"])[:50]
    print(f"{nm:50s} {int(r['Calls'])/7:6.1f}/step {float(r['TotalDurationNs'])/7e3:8.1f} us/step {float(r['AverageNs'])/1e3:8.1f} us")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
