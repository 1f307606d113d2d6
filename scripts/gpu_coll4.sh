#!/bin/bash
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
run SVSR_SIDE_TRUNK=0
run SVSR_DBG_PGONLY=1 SVSR_SIDE_TRUNK=0
run SVSR_DBG_PGONLY=1 SVSR_DBG_EARLYSIDE=1
run SVSR_DBG_DUMMY_STREAMS=1
run SVSR_DBG_DUMMY_STREAMS=2
run SVSR_DBG_DUMMY_STREAMS=3
run SVSR_DBG_DUMMY_STREAMS=4
run SVSR_DBG_PGONLY=1 GPU_MAX_HW_QUEUES=8
