#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --profile-steps 1 --steps 40 --warmup 5 > $OUT/b_$tag.json 2> $OUT/b_$tag.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/b_$tag.json").read().strip().splitlines()[-1]); print("$tag", d["ms_per_step"], "host", d.get("host_enqueue_ms"))
except Exception as e: print("$tag FAILED", e)
PY
}
run base X=1
run no_conv_wgrad SVSR_ABLATE=conv_wgrad
run no_lin_wgrad SVSR_ABLATE=lin_wgrad
run no_wgrad SVSR_ABLATE=conv_wgrad,lin_wgrad
run no_side SVSR_SIDE_TRUNK=0
run side_all SVSR_SIDE_ENCODER=1
run base2 X=1
