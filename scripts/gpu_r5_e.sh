#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_lrs_kernels.py -x -q -k "mha or flash" 2>&1 | tail -5
./scripts/gpu_r5_d.sh 2>&1 | grep -E "total kernel|mha|wgrad|igemm_fwd|add_ln|adamw"
cd $GRAFT_REPO_ROOT
python bench.py --workload lrs --no-cpu-baseline --profile-steps 1 --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('lrs ms/step', d['ms_per_step'], 'loss', d.get('final_loss'))"
