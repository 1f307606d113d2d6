#!/bin/bash
# timeline of one training step (both streams): kernel, queue, start / end relative to the step's first launch — which launches run beside which
W=${WORKLOAD:-lrw}
cd /tmp; export TMPDIR=/tmp PYTHONUNBUFFERED=1
rm -rf /tmp/tl; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o l -- python $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --no-lrs-leg --profile-steps 0 --steps 6 --warmup 3 > /tmp/tl.log 2>&1
python - <<'PY'
import csv, glob, re, os
f = glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:34], r["Queue_Id"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows)
marks = [i for i, e in enumerate(ev) if e[2].startswith("k_stem_prep") or e[2].startswith("k_clip_prep")]
lo, hi = marks[-3], marks[-2]          # one whole step, from its first launch to the next step's
seg = ev[lo:hi]
t0 = seg[0][0]
qs = sorted({e[3] for e in seg})
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", f"timeline_{os.environ.get('WORKLOAD', 'lrw')}.txt")
with open(out, "w") as fo:
    for s, e, n, q, g in seg:
        fo.write(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{qs.index(q)} g{g:<5d} {n}\n")
print("wrote", out, len(seg), "launches, span", (seg[-1][1] - t0) / 1e3, "us")
PY
