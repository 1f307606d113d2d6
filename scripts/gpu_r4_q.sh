#!/bin/bash
# dense layers on the persistent kernel (svsr_rows_plan_k): LRS step A/B
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_bench_shapes.py -x -q -k "wide_linears" 2>&1 | tail -2
B="python bench.py --workload lrs --no-cpu-baseline --profile-steps 0 --steps 12 --warmup 3"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d['ms_per_step'], d.get('final_loss'))" "$1"; }
$B 2>/dev/null | pick "p8 linears      "
$B --tune p8_lin_items=0 2>/dev/null | pick "4-wave linears  "
$B 2>/dev/null | pick "p8 linears      "
$B --tune p8_lin_items=0 2>/dev/null | pick "4-wave linears  "
