#!/bin/bash
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --force-collective --bucket-mb 64 --profile-steps 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
run A=1
run SVSR_DBG_NOAR=1
run SVSR_DBG_NOBCAST=1
run SVSR_DBG_NOAR=1 SVSR_DBG_NOBCAST=1
