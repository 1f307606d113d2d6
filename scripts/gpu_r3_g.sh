#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_train.py tests/test_gpu_ddp_ranks.py -x -q -m gpu 2>&1 | grep -v "^ \|^{\|^}" | tail -8 | tee $OUT/tests.log
for mode in native; do
  timeout 300 python bench.py --no-cpu-baseline --no-lrs-leg --profile-steps 1 --steps 40 --warmup 5 > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3g/bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        pk = d["roofline"]["per_kernel"]
        print(os.path.basename(f), d["ms_per_step"], d["value"], "host", d.get("host_enqueue_ms"), "dom", d["roofline"]["kernel"], d["roofline"]["frac"])
        for k, v in pk.items(): print("   ", k, v["ms_per_step"], v["tflops"], v["launches"])
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-800:])
PY
