#!/bin/bash
# gradient buffer zeroed at step start on the side stream: A/B + the tests that read gradients after steps
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 8"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d['ms_per_step'], 'lrs', d.get('lrs',{}).get('ms_per_step'), d.get('final_loss'))" "$1"; }
$B 2>/dev/null | pick "early zero  "
SVSR_ZERO_GRADS_EARLY=0 $B 2>/dev/null | pick "zero in bwd "
$B 2>/dev/null | pick "early zero  "
SVSR_ZERO_GRADS_EARLY=0 $B 2>/dev/null | pick "zero in bwd "
