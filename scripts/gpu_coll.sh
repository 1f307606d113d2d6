#!/bin/bash
mkdir -p gpurun_out
for mb in 16 32 64 128 1000; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --force-collective --bucket-mb $mb --profile-steps 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bucket_mb', $mb, d['value'], d['ms_per_step'], d['collective']['all_reduce_launches_per_step'])"
done 2>&1 | tee gpurun_out/coll_sweep.log
