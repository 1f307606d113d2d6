#!/bin/bash
# round 5, experiment A: main / side stream on disjoint compute units (hipExtStreamCreateWithCUMask)
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r5a
./scripts/probes/cumask_probe 2>&1 | tee gpurun_out/r5a/cumask_probe.txt
B="python bench.py --no-cpu-baseline --no-lrs-leg --profile-steps 1 --steps 40 --warmup 8"
run() { echo "== $*"; $B "$@" 2>gpurun_out/r5a/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('ms/step', d['ms_per_step'], 'host', d.get('host_enqueue_ms'), 'loss', d.get('final_loss'), d['config'].get('cu_split'))" || tail -5 gpurun_out/r5a/err.txt; }
run
run --tune w3_blocks=288
run --tune w3_blocks=512
run --cu-split 32:spread
run --cu-split 32:xcd
run --cu-split 64:spread
run --cu-split 64:xcd
run --cu-split 64:spread --tune wg_blocks=128,w3_blocks=128
run --cu-split 96:xcd --tune wg_blocks=192,w3_blocks=192
run --cu-split 128:xcd --tune wg_blocks=256,w3_blocks=256
run --cu-split 16:spread --tune wg_blocks=64,w3_blocks=64
run
