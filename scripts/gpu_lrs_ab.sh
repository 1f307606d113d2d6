#!/bin/bash
# LRS whole-step A/B of tuning knobs (each argument one --tune value; interleaved twice)
for rep in 1 2; do
for t in "$@"; do
  python bench.py --workload lrs --steps 12 --warmup 3 --no-cpu-baseline --profile-steps 1 --tune "$t" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lrs tune[$t]', d['value'], d['ms_per_step'])"
done; done
