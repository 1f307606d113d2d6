#!/bin/bash
# per-kernel time table of the in-line step (no side stream), rocprofv3 --kernel-trace --stats
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstats3
rm -rf $OUT; mkdir -p $OUT
export PYTHONUNBUFFERED=1 SVSR_SIDE_TRUNK=0 SVSR_SIDE_ENCODER=0
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lrw -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-lrs-leg --profile-steps 1 --steps 5 --warmup 2 --enqueue eager > $OUT/lrw_run.log 2>&1; echo "stats $?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os, re
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "kstats3")
f = glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 9
tot = 0
lines = []
for r in rows:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:58]
    t = int(r["TotalDurationNs"]) / steps / 1e3
    tot += t
    lines.append(f"{t:8.1f} us/step {int(r['Calls'])/steps:6.1f} calls avg {float(r['AverageNs'])/1e3:7.1f}  {name}")
lines.append(f"total {tot:.1f}")
open(os.path.join(out, "table.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:70])); print(lines[-1])
PY
cp $(find $OUT -name "*kernel_stats.csv" | head -1) $OUT/lrw_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
