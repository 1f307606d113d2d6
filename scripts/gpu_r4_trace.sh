#!/bin/bash
# One step's launch timeline (default mode: native step list, side-stream weight gradients) as a compact TSV:
#   start_us  dur_us  queue  kernel            -> gpurun_out/r4trace/step.tsv  (+ the in-line variant step_inline.tsv)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4trace
rm -rf $OUT; mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
for mode in ${MODES:-side inline}; do
  if [ $mode = inline ]; then export SVSR_SIDE_TRUNK=0; else unset SVSR_SIDE_TRUNK; fi
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/$mode -o tr -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-lrs-leg --profile-steps 0 --steps 6 --warmup 3 $@ > $OUT/run_$mode.log 2>&1; echo "trace $mode $?"
  tail -c 300 $OUT/run_$mode.log
  python - $OUT $mode <<'PY'
import csv, glob, os, re, sys
out, mode = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(out, mode, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def short(n):
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:70]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows)
adam = [i for i, e in enumerate(ev) if e[2].startswith("k_adamw")]
lo, hi = adam[5], adam[6]          # a timed step of the native list (3 warm-up + 6 timed + the eager profile steps)
seg = ev[lo + 1: hi + 1]
t0 = seg[0][0]
with open(os.path.join(out, f"step_{mode}.tsv"), "w") as w:
    for s, e, n, q in seg:
        w.write(f"{(s - t0) / 1e3:.2f}\t{(e - s) / 1e3:.2f}\t{q}\t{n}\n")
print(mode, "launches", len(seg), "span us", (seg[-1][1] - t0) / 1e3)
PY
  find $OUT/$mode -name "*kernel_trace.csv" -delete; find $OUT/$mode -name "*.db" -delete
done
