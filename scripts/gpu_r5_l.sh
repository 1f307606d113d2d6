#!/bin/bash
# round 5: fused audio head — kernel test, model / training tests, bench
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_audio_head" 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_xt.py tests/test_gpu_train.py -x -q 2>&1 | tail -4
B="python bench.py --no-cpu-baseline --no-lrs-leg --profile-steps 1 --steps 40 --warmup 8"
for i in 1 2; do $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('LRW ms/step', d['ms_per_step'], 'launches', d['launches_per_step'], 'loss', d.get('final_loss'), {k: (v['ms_per_step'], v['tflops']) for k, v in d['roofline']['per_kernel'].items() if 'linear_ce' in k})"; done
