#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3n
for t in "" "stem_keep_winners=0" "" "stem_keep_winners=0"; do
  python bench.py --no-cpu-baseline --no-lrs-leg --steps 100 --warmup 10 --tune "$t" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tune=[$t]', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r3n/ab6.log
done
for t in "" "stem_keep_winners=0"; do
  python bench.py --workload lrs --no-cpu-baseline --steps 10 --warmup 3 --tune "$t" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('LRS tune=[$t]', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r3n/ab6.log
done
bash scripts/gpu_kstats3.sh 2>&1 | grep "stem\|total"
