#!/bin/bash
# Runs on the GPU box (via gpurun): the whole -m gpu suite, then short bench runs (eager, graph); logs under gpurun_out/.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
rocminfo 2>/dev/null | grep -m1 -E "gfx9" > gpurun_out/gpu.txt
timeout ${1:-2400} python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/tests.log 2>&1
echo "tests exit $?" >> gpurun_out/tests.log
grep -E "passed|failed|error" gpurun_out/tests.log | tail -5
grep -E "^(FAILED|ERROR)" gpurun_out/tests.log | head -40
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_eager.json 2> gpurun_out/bench_eager.err
tail -c 1500 gpurun_out/bench_eager.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph --profile-steps 1 > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err
python - <<'PY'
import json
for f in ("bench_eager", "bench_graph"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("final_loss"))
    except Exception as e:
        print(f, "failed", e)
PY
