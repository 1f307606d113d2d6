#!/bin/bash
# the driver's own default line (LRW + LRS leg + CPU baseline), timed
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
S=$SECONDS
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
echo "bench.py default: $((SECONDS - S)) s, rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "host_enqueue_ms", "step_mfma_frac", "n_gpus", "steps", "warmup", "dtype") if k in d})
print("roofline", {k: d["roofline"][k] for k in ("kernel", "bound", "achieved", "peak", "frac", "traffic") if k in d["roofline"]})
print("cpu_baseline", d.get("cpu_baseline"))
print("lrs", {k: v for k, v in d.get("lrs", {}).items() if k != "roofline"})
PY
