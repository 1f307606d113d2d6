#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
for ph in 1 2; do
python - <<PY
import sys; sys.argv=["x"]
sys.path.insert(0, "scripts/probes")
import torch
from syncvsr_amd import ops
ops.tune("p8_ph", $ph); ops.tune("p8_min_items", 1)
import p8_check as P
print("PH=$ph")
P.run(928, 11, 128); P.run(928, 6, 256)
PY
done 2>&1 | grep -v "rel diff\|amdgpu" | tee $OUT/p8_check2.log
timeout 300 python scripts/probes/lin_bench.py default 2>&1 | grep -v amdgpu.ids | tee $OUT/lin.log
timeout 300 python bench.py --no-cpu-baseline --no-lrs-leg --profile-steps 1 --steps 40 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3b/bench.json")).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], "host", d.get("host_enqueue_ms"), "dom", d["roofline"]["kernel"], d["roofline"]["frac"])
for k, v in d["roofline"]["per_kernel"].items(): print("   ", k, v["ms_per_step"], v["tflops"], v["launches"], v.get("bn_epilogue_tflops"))
PY
