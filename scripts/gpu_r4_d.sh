#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
for t in wg_units=0 wg_units=1 wg_units=0 wg_units=1; do
  timeout 900 python bench.py --workload lrs --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 1 --tune $t > gpurun_out/r4d/b.json 2> gpurun_out/r4d/b.err
  python - "$t" <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r4d/b.json").read().strip().splitlines()[-1]); print(t, d["ms_per_step"], "host", d.get("host_enqueue_ms"), "loss", d["final_loss"], {k: (v["ms_per_step"], v["tflops"]) for k, v in d["roofline"]["per_kernel"].items() if "wgrad" in k})
except Exception as e: print(t, "FAILED", e); print(open("gpurun_out/r4d/b.err").read()[-2000:])
PY
done
