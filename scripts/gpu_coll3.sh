#!/bin/bash
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
run A=1
run SVSR_DBG_PGONLY=1
run SVSR_DBG_PGONLY=1 TORCH_NCCL_ENABLE_MONITORING=0 TORCH_NCCL_ASYNC_ERROR_HANDLING=0
