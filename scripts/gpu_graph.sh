#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/train_tests.log 2>&1
tail -n 5 gpurun_out/train_tests.log
run() {  # tag, flags
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 1 $2 > gpurun_out/bench_$1.json 2> gpurun_out/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_$1.json").read().strip().splitlines()[-1])
    print("$1", d["value"], d["ms_per_step"], d.get("final_loss"))
except Exception as e:
    print("$1 failed", e); print(open("gpurun_out/bench_$1.err").read()[-1500:])
PY
}
for g in 1 4 8 100; do
  export SVSR_SIDE_GROUP=$g
  unset SVSR_GRAPH_SIDE; run eager_g$g ""
  export SVSR_GRAPH_SIDE=1; run graphside_g$g "--graph"
done
unset SVSR_GRAPH_SIDE SVSR_SIDE_GROUP
run graph "--graph"
python scripts/host_overhead.py 2>&1 | grep -E "enqueue"
