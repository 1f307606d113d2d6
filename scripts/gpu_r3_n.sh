#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3n
for t in "w3_blocks=320" "w3_blocks=320,wg_blocks=256" "w3_blocks=320,wg_blocks=448" "w3_blocks=320,wg_blocks=640" "w3_blocks=288" "w3_blocks=320" "w3_blocks=320,wg_blocks=448" "w3_blocks=224"; do
  python bench.py --no-cpu-baseline --no-lrs-leg --steps 60 --warmup 10 --tune "$t" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pk=d['roofline']['per_kernel']
print('tune=[$t]', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r3n/ab3.log
done
