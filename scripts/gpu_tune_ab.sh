#!/bin/bash
# same-box A/B of tuning knobs (bench.py --tune), 3 interleaved rounds of the LRW step and 2 of the LRS step: bash scripts/gpu_tune_ab.sh "w3_dense=0" "w3_dense=1"
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
for i in 1 2 3; do
for cfg in "$@"; do
  r=$(python bench.py --tune "$cfg" --no-cpu-baseline --no-lrs-leg --sustained-steps 0 --profile-steps 0 --steps 60 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d.get('final_loss'))")
  echo "LRW [$cfg] ms/step $r"
done
done
if [ -z "$NO_LRS" ]; then
for i in 1 2; do
for cfg in "$@"; do
  r=$(python bench.py --workload lrs --tune "$cfg" --no-cpu-baseline --profile-steps 0 --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d.get('final_loss'))")
  echo "LRS [$cfg] ms/step $r"
done
done
fi
