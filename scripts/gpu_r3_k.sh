#!/bin/bash
# p8 for layer4 (132 items < 256 CUs): A/B of the item threshold
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3k
python -m syncvsr_amd.build > /dev/null 2>&1
for t in "" "p8_min_items=100"; do
  for k in 1 2; do
    python bench.py --no-cpu-baseline --no-lrs-leg --steps 40 --warmup 10 --tune "$t" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tune=[$t]', d['ms_per_step'], d['value'], d.get('host_enqueue_ms'), d['roofline'].get('kernel'), d['roofline']['frac'])
for k,v in d['roofline']['per_kernel'].items():
    if 'p8' in k or '128,64,2' in k or '128,128' in k: print('   ', k, v)" | tee -a gpurun_out/r3k/ab.log
  done
done
