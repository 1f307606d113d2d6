#!/bin/bash
# round 5: side stream confined to a subset of the compute units, main stream unmasked ("reserve")
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r5k
B="python bench.py --no-cpu-baseline --no-lrs-leg --profile-steps 0 --steps 40 --warmup 8"
run() { echo "== $*"; $B "$@" 2>gpurun_out/r5k/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('ms/step', d['ms_per_step'], 'loss', d.get('final_loss'), d['config'].get('cu_split'))" || tail -5 gpurun_out/r5k/err.txt; }
run
run --cu-split 240:reserve
run --cu-split 224:reserve
run --cu-split 208:reserve
run --cu-split 192:reserve
run --cu-split 160:reserve
run
run --cu-split 224:reserve
echo "== LRS"
B="python bench.py --workload lrs --no-cpu-baseline --profile-steps 0 --steps 12 --warmup 3"
run
run --cu-split 224:reserve
run --cu-split 192:reserve
run
