#!/bin/bash
# round 3, first GPU call: native step list — parity tests + eager vs native bench A/B
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3a
rm -rf $OUT; mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "native or checkpoint_resume_lrw or three_steps or rccl" 2>&1 | tail -25 | tee $OUT/tests.log
for mode in eager native eager native; do
  timeout 300 python bench.py --no-cpu-baseline --profile-steps 1 --steps 40 --warmup 5 --enqueue $mode > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  python - <<PY
import json
l=open("$OUT/bench_$mode.json").read().strip().splitlines()
try:
    d=json.loads(l[-1]); print("$mode", d["ms_per_step"], d["value"], "host", d.get("host_enqueue_ms"), "dom", d["roofline"]["kernel"], d["roofline"]["frac"])
except Exception as e:
    print("$mode FAILED", e); print(open("$OUT/bench_$mode.err").read()[-1500:])
PY
done
