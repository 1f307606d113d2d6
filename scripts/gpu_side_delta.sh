#!/bin/bash
# which main-stream kernels stretch under the side stream?  per-kernel time per step with the weight-gradient side stream on and off
# (rocprofv3 --kernel-trace of the eager step; WORKLOAD=lrw|lrs)
W=${WORKLOAD:-lrw}
cd /tmp; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for side in 1 0; do
  rm -rf /tmp/pp$side; SVSR_SIDE_TRUNK=$side SVSR_SIDE_ENCODER=$side timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp$side -o l -- python $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --no-lrs-leg --profile-steps 0 --steps ${STEPS:-6} --warmup 2 --enqueue eager > /tmp/run$side.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re, os
def load(d):
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        nm = re.sub(r"\(.*", "", r["Kernel_Name"])[:44]
        g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        k = (nm, g if g < 300 else ">=300")
        agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg
on, off = load('/tmp/pp1'), load('/tmp/pp0')
steps = float(os.environ.get("STEPS", "6")) + 3.0
rows = []
for k in on:
    if k in off and on[k][0] == off[k][0]:
        rows.append((on[k][1] - off[k][1], k, on[k][0], on[k][1], off[k][1]))
rows.sort(reverse=True)
print(f"{'kernel':44s} {'grid':>6s} {'n/step':>6s} {'on us/step':>10s} {'off us/step':>11s} {'delta':>8s} {'avg on':>7s} {'avg off':>7s}")
for d, (nm, g), n, a, b in rows[:45]:
    print(f"{nm:44s} {str(g):>6s} {n/steps:6.1f} {a/steps:10.1f} {b/steps:11.1f} {d/steps:8.1f} {a/n:7.1f} {b/n:7.1f}")
print("sum of deltas (us/step):", sum(r[0] for r in rows) / steps)
PY
