#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
for t in 0 2 4 0 2 4; do
  SVSR_W3_GROUP=$t timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-lrs-leg --profile-steps 1 > gpurun_out/r4g/b.json 2> gpurun_out/r4g/b.err
  python - "$t" <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r4g/b.json").read().strip().splitlines()[-1]); print("w3_group", t, d["ms_per_step"], "host", d.get("host_enqueue_ms"), "loss", d["final_loss"], {k: (v["ms_per_step"], v["launches"]) for k, v in d["roofline"]["per_kernel"].items() if "halo" in k})
except Exception as e: print(t, "FAILED", e); print(open("gpurun_out/r4g/b.err").read()[-1500:])
PY
done
SVSR_W3_GROUP=4 timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
