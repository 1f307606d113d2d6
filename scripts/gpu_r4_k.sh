#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_lrs_model.py tests/test_gpu_model.py -x -q 2>&1 | tail -4
for t in 0 1 0 1; do
  SVSR_DEFER_REDUCTIONS=$t timeout 900 python bench.py --workload lrs --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 1 > gpurun_out/r4k/b.json 2> gpurun_out/r4k/b.err
  python - "$t" <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r4k/b.json").read().strip().splitlines()[-1]); print("defer", t, d["ms_per_step"], "idle-queue host", d.get("host_enqueue_idle_queue_ms"), "launches", d.get("launches_per_step"))
except Exception as e: print(t, "FAILED", e); print(open("gpurun_out/r4k/b.err").read()[-2500:])
PY
done
