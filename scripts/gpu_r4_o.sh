#!/bin/bash
# encoder weight gradients behind the fused backward: how many layers per grouped launch
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
B="python bench.py --no-cpu-baseline --no-lrs-leg --profile-steps 0 --steps 60 --warmup 8"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d['ms_per_step'], 'launches', d.get('launches_per_step'), d.get('final_loss'))" "$1"; }
for g in 1 0 2 3 6; do SVSR_WG_GROUP_LAYERS=$g $B 2>/dev/null | pick "layers per grouped launch $g:"; done
SVSR_WG_GROUP_LAYERS=1 $B 2>/dev/null | pick "layers per grouped launch 1:"
