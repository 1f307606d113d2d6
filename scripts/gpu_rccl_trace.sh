#!/bin/bash
# One-rank RCCL path of the training step under rocprofv3 --kernel-trace: which kernels does the collective library launch, with what grid / LDS /
# register footprint, and beside which of the step's kernels do they run?  (8 GPUs are not available to the builder.)
cd /tmp; export TMPDIR=/tmp PYTHONUNBUFFERED=1
rm -rf /tmp/rc; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rc -o l -- python $GRAFT_REPO_ROOT/bench.py --force-collective --no-cpu-baseline --no-lrs-leg --sustained-steps 0 --profile-steps 0 --steps 6 --warmup 3 > /tmp/rc.log 2>&1
tail -1 /tmp/rc.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], json.dumps(d.get('collective'))[:1500])"
python - <<'PY'
import csv, glob, re, collections
f = glob.glob('/tmp/rc/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("columns:", [c for c in rows[0].keys()][:24])
coll = [r for r in rows if re.search(r"nccl|rccl|AllReduce|Broadcast|ncclDev", r["Kernel_Name"], re.I)]
agg = collections.defaultdict(list)
for r in coll:
    agg[(re.sub(r"\(.*", "", r["Kernel_Name"])[:70], r.get("Grid_Size_X"), r.get("Workgroup_Size_X"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")), r.get("VGPR_Count", "?"), r.get("SGPR_Count", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:72s} grid {k[1]:>6s} wg {k[2]:>4s} lds {k[3]:>6s} vgpr {k[4]:>4s} sgpr {k[5]:>4s}  n {len(v):4d}  avg {sum(v)/len(v):8.1f} us  max {max(v):8.1f}")
# what runs beside the collectives of the last step
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:40]) for r in rows)
last = [e for e in ev if re.search(r"nccl|rccl|ncclDev", e[2], re.I)][-6:]
for s, e, n in last:
    beside = sorted({x[2] for x in ev if x[0] < e and x[1] > s and x[2] != n})
    print(f"{n[:34]:36s} {(e - s) / 1e3:8.1f} us  beside: {', '.join(beside)[:220]}")
PY
python - <<'PY'
import csv, glob, re, collections
f = glob.glob('/tmp/rc/**/*kernel_trace.csv', recursive=True)[0]
c = collections.Counter(re.sub(r"\(.*", "", r["Kernel_Name"])[:60] for r in csv.DictReader(open(f)))
print("kernels that are not this library's:", {k: v for k, v in c.items() if not k.replace("void ", "").startswith("k_")})
PY
grep -o '"collective": {[^}]*}' /tmp/rc.log | tail -1
