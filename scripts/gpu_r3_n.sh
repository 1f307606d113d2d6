#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3n
python -m syncvsr_amd.build >/dev/null 2>&1
for k in 1 2 3; do
  python bench.py --no-cpu-baseline --no-lrs-leg --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pk=d['roofline']['per_kernel']
print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], {k: v['ms_per_step'] for k,v in pk.items() if 'c64' in k})"
done
python bench.py --workload lrs --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LRS', d['ms_per_step'], d['value'])"
