#!/bin/bash
# re-tune of the side-stream launch grids after the encoder fusion
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
B="python bench.py --no-cpu-baseline --no-lrs-leg --profile-steps 0 --steps 60 --warmup 8"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d['ms_per_step'], d.get('final_loss'))" "$1"; }
$B 2>/dev/null | pick "default          "
for v in 256 320 384 512; do $B --tune w3_blocks=$v 2>/dev/null | pick "w3_blocks=$v     "; done
for v in 256 384 512 768; do $B --tune wg_blocks=$v 2>/dev/null | pick "wg_blocks=$v     "; done
$B 2>/dev/null | pick "default          "
