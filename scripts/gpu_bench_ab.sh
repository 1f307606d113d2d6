#!/bin/bash
# whole-step A/B of tuning knobs: gpu_bench_ab.sh "" "igemm_ksplit=0" ...   (each argument is one --tune value; runs are interleaved twice)
mkdir -p gpurun_out
for rep in 1 2; do
for t in "$@"; do
  python bench.py --steps 60 --warmup 8 --no-cpu-baseline --profile-steps 1 --tune "$t" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tune[$t]', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/bench_ab.log
