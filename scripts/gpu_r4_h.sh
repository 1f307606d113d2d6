#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
for t in 0 1 0 1; do
  SVSR_DEFER_REDUCTIONS=$t timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-lrs-leg --profile-steps 1 > gpurun_out/r4h/b.json 2> gpurun_out/r4h/b.err
  python - "$t" <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r4h/b.json").read().strip().splitlines()[-1]); print("defer", t, d["ms_per_step"], "host", d.get("host_enqueue_ms"), "loss", d["final_loss"])
except Exception as e: print(t, "FAILED", e); print(open("gpurun_out/r4h/b.err").read()[-1500:])
PY
done
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py tests/test_gpu_ddp_ranks.py -x -q 2>&1 | tail -3
