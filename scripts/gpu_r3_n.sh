#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3n
for t in "" "wg_blocks=448" "" "wg_blocks=448" "wg_blocks=480" "wg_blocks=416"; do
  python bench.py --no-cpu-baseline --no-lrs-leg --steps 100 --warmup 10 --tune "$t" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tune=[$t]', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r3n/ab4.log
done
for t in "" "w3_blocks=512" ""; do
  python bench.py --workload lrs --no-cpu-baseline --steps 10 --warmup 3 --tune "$t" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('LRS tune=[$t]', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r3n/ab4.log
done
