#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3n
python -m syncvsr_amd.build >/dev/null 2>&1
for t in "" "p8_bn64=0" "" "p8_bn64=0" "p8_bn64=0,p8=3" ""; do
  python bench.py --no-cpu-baseline --no-lrs-leg --steps 100 --warmup 10 --tune "$t" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pk=d['roofline']['per_kernel']
print('tune=[$t]', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], {k: (v['ms_per_step'], v['launches']) for k,v in pk.items() if 'p8' in k or '128,64,2' in k})"
done
