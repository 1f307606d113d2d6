#!/bin/bash
# kernel trace of the default (side-stream) step: where does the time between kernel-sum and wall-clock go?
OUT=$GRAFT_REPO_ROOT/gpurun_out/gaps
rm -rf $OUT; mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --profile-steps 0 --steps 6 --warmup 3 $@ > $OUT/run.log 2>&1; echo "trace $?"
tail -c 400 $OUT/run.log
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os, re, collections
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "gaps")
f = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])[:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
ev.sort()
# steps: find k_adamw launches
adam = [i for i, e in enumerate(ev) if e[2].startswith("k_adamw")]
print("adamw launches", len(adam))
lo, hi = adam[-4], adam[-1]            # three full steps
seg = ev[lo + 1: hi + 1]
t0, t1 = seg[0][0], seg[-1][1]
span = (t1 - t0) / 3e3
ksum = sum(e[1] - e[0] for e in seg) / 3e3
# union busy
busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
gaps = []
for s, e, n, q, st in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"per step: span {span:.1f} us, kernel sum {ksum:.1f} us, union busy {busy/3e3:.1f} us, idle {span - busy/3e3:.1f} us in {len(gaps)/3:.0f} gaps")
byq = collections.defaultdict(float)
for s, e, n, q, st in seg: byq[(q, st)] += (e - s) / 3e3
print("busy per (queue, stream):", {k: round(v, 1) for k, v in byq.items()})
h = collections.Counter()
for g, n in gaps: h[min(g // 1000, 20)] += 1
print("gap histogram (us: count/step):", {k: round(v / 3, 1) for k, v in sorted(h.items())})
after = collections.defaultdict(float)
for g, n in gaps: after[n] += g / 3e3
print("idle before kernel (us/step):", sorted(((round(v, 1), k) for k, v in after.items()), reverse=True)[:25])
open(os.path.join(out, "summary.txt"), "w").write(f"span {span} ksum {ksum} busy {busy/3e3}\n")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
