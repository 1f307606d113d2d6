#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
timeout 1800 python -m pytest tests/test_gpu_train.py tests/test_gpu_lrs_model.py tests/test_gpu_model.py tests/test_gpu_lrs_infer.py tests/test_gpu_ddp_ranks.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -4
for t in 0 1 0 1; do
  SVSR_SPLIT_OPTIMIZER=$t timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 1 --lrs-steps 10 > gpurun_out/r4l/b.json 2> gpurun_out/r4l/b.err
  python - "$t" <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r4l/b.json").read().strip().splitlines()[-1]); print("split", t, "LRW", d["ms_per_step"], "LRS", d["lrs"]["ms_per_step"], d["lrs"].get("host_enqueue_ms"))
except Exception as e: print(t, "FAILED", e); print(open("gpurun_out/r4l/b.err").read()[-2500:])
PY
done
