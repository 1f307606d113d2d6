#!/bin/bash
# whole-step A/B of environment switches, interleaved: gpu_env_ab.sh "VAR=0" "VAR=1" ...
mkdir -p gpurun_out
for rep in 1 2 3; do
for e in "$@"; do
  env $e python bench.py --steps 60 --warmup 8 --no-cpu-baseline --profile-steps 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$e', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/env_ab.log
