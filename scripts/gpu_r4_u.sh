#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
B="python bench.py --workload lrs --no-cpu-baseline --profile-steps 0 --steps 16 --warmup 3"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d['ms_per_step'], d.get('final_loss'))" "$1"; }
for i in 1 2; do
$B 2>/dev/null | pick "embedding gradient on the side stream"
SVSR_EMBED_SIDE=0 $B 2>/dev/null | pick "embedding gradient on the main stream"
done
