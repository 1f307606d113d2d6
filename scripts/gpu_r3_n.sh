#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3n
for t in "" "p8_grid=4096" "p8_grid=384" "" "p8_grid=4096"; do
  python bench.py --no-cpu-baseline --no-lrs-leg --steps 100 --warmup 10 --tune "$t" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tune=[$t]', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r3n/ab5.log
done
