#!/bin/bash
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 1 $EXTRA 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$* $EXTRA', d['value'], d['ms_per_step'])"; }
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=6
run GPU_MAX_HW_QUEUES=16
EXTRA="--force-collective --bucket-mb 16"
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=16
EXTRA="--force-collective --bucket-mb 64"
run GPU_MAX_HW_QUEUES=8
EXTRA="--graph"
run GPU_MAX_HW_QUEUES=8
