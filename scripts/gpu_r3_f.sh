#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
SECONDS=0; timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "wall ${SECONDS}s"
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "host_enqueue_ms", "step_mfma_frac")})
print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
print("cpu", d["cpu_baseline"])
print("lrs", {k: v for k, v in d["lrs"].items() if k not in ("roofline", "config")}, d["lrs"].get("roofline", {}).get("kernel"), d["lrs"].get("roofline", {}).get("frac"))
PY
