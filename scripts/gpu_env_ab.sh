#!/bin/bash
# same-box A/B of environment switches of the training step (3 interleaved rounds): [WORKLOAD=lrs] bash scripts/gpu_env_ab.sh "VAR=1" "VAR=0" ...
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
for i in 1 2 3; do
for cfg in "$@"; do
  r=$(env $cfg python bench.py --workload ${WORKLOAD:-lrw} --no-cpu-baseline --no-lrs-leg --sustained-steps 0 --profile-steps 0 --steps ${STEPS:-60} --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d.get('final_loss'))")
  echo "${WORKLOAD:-lrw} [$cfg] ms/step $r"
done
done
