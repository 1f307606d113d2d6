#!/bin/bash
# per-dispatch durations of selected small kernels in the LRS step, grouped by grid size (which call site is slow?)
cd /tmp; export TMPDIR=/tmp PYTHONUNBUFFERED=1
rm -rf /tmp/pp; export SVSR_SIDE_TRUNK=${SIDE:-1} SVSR_SIDE_ENCODER=${SIDE:-1}; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o l -- python $GRAFT_REPO_ROOT/bench.py --workload lrs --no-cpu-baseline --profile-steps 0 --steps 3 --warmup 1 --enqueue eager > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, re
f = glob.glob('/tmp/pp/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
agg = collections.defaultdict(list)
for r in rows:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"])[:40]
    if not any(k in nm for k in ("bn_bwd_finalize", "bn_finalize", "k_add_ln_bwd", "k_add_ln_fwd", "k_bias_act_bwd", "k_scale_bf16", "k_colsum", "reduce_flat", "bn_act_bwd_apply", "bn_act_fwd", "k_glu", "k_dw_reduce")): continue
    g = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r.get("Grid_Size_Y", 1)), int(r["Workgroup_Size_X"]))
    agg[(nm, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (nm, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{nm:40s} grid {str(g):22s} n={len(v):4d} med {v[len(v)//2]:7.1f} us max {v[-1]:7.1f} tot {sum(v)/1e3:7.2f} ms")
PY
