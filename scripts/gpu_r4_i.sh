#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i

for t in native eager native; do
  timeout 900 python bench.py --workload lrs --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 1 --enqueue $t > gpurun_out/r4i/b.json 2> gpurun_out/r4i/b.err
  python - "$t" <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r4i/b.json").read().strip().splitlines()[-1]); print(t, d["ms_per_step"], "host", d.get("host_enqueue_ms"), "idle-queue host", d.get("host_enqueue_idle_queue_ms"), "launches", d.get("launches_per_step"), "loss", d["final_loss"])
except Exception as e: print(t, "FAILED", e); print(open("gpurun_out/r4i/b.err").read()[-2500:])
PY
done
