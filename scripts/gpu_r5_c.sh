#!/bin/bash
# round 5: streamed-key attention (mha_flash.h) — kernel tests, LRS model tests, LRS step A/B
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r5c
timeout 900 python -m pytest tests/test_gpu_lrs_kernels.py -x -q -k "mha or flash" 2>&1 | tail -25
timeout 1200 python -m pytest tests/test_gpu_lrs_model.py tests/test_gpu_train.py -x -q -k "lrs" 2>&1 | tail -8
B="python bench.py --workload lrs --no-cpu-baseline --profile-steps 1 --steps 12 --warmup 3"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1 lrs ms/step', d['ms_per_step'], 'loss', d.get('final_loss'), {k: (v['ms_per_step'], v['launches']) for k, v in d['roofline']['per_kernel'].items() if 'mha' in k})"; }
SVSR_MHA_FLASH=0 $B 2>gpurun_out/r5c/e0.txt | pick "per-tile" || tail -5 gpurun_out/r5c/e0.txt
SVSR_MHA_FLASH=1 $B 2>gpurun_out/r5c/e1.txt | pick "flash   " || tail -5 gpurun_out/r5c/e1.txt
SVSR_MHA_FLASH=0 $B 2>/dev/null | pick "per-tile"
SVSR_MHA_FLASH=1 $B 2>/dev/null | pick "flash   "
