#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
run() {  # tag, flags
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 1 $2 > gpurun_out/bench_$1.json 2> gpurun_out/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_$1.json").read().strip().splitlines()[-1])
    print("$1", d["value"], d["ms_per_step"], d.get("final_loss"))
except Exception as e:
    print("$1 failed", e); print(open("gpurun_out/bench_$1.err").read()[-800:])
PY
}
export SVSR_SIDE_GROUP=4 SVSR_GRAPH_SIDE=1
for q in 1 2 4 8; do DEBUG_HIP_FORCE_GRAPH_QUEUES=$q run gs_q$q "--graph"; done
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run gs_nocapture "--graph"
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=4 run gs_nocapture_q4 "--graph"
GPU_MAX_HW_QUEUES=8 run gs_hwq8 "--graph"
DEBUG_HIP_GRAPH_BATCH_SIZE=1 run gs_batch1 "--graph"
DEBUG_HIP_GRAPH_BATCH_SIZE=1000 run gs_batch1000 "--graph"
