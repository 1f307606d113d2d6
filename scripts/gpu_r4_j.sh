#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
for i in 1 2 3; do
  timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-lrs-leg --profile-steps 1 $@ > gpurun_out/r4j/b.json 2> gpurun_out/r4j/b.err
  python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4j/b.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], "host", d.get("host_enqueue_ms"), "loss", d["final_loss"], "roof", d["roofline"]["kernel"], d["roofline"]["frac"])
except Exception as e: print("FAILED", e); print(open("gpurun_out/r4j/b.err").read()[-1500:])
PY
done
