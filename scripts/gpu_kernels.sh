#!/bin/bash
# Runs on the GPU box: kernel-level tests + op micro-benchmarks + a short bench run.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_shapes.py tests/test_gpu_misc.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/ktests.log 2>&1
echo "ktests exit $?" >> gpurun_out/ktests.log
tail -n 25 gpurun_out/ktests.log
timeout 600 python scripts/op_bench.py ${1:-fwd dgrad wgrad linear} > gpurun_out/op_bench.log 2>&1
cat gpurun_out/op_bench.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_eager.json 2> gpurun_out/bench_eager.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_eager.json").read().strip().splitlines()[-1])
    print("bench", d["value"], d["ms_per_step"], d.get("final_loss"))
    for k, v in d["roofline"]["per_kernel"].items():
        print("   ", k, v)
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/bench_eager.err").read()[-2000:])
PY
