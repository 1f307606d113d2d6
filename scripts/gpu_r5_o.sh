#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
for x in 1 5; do echo "== wg_xcd=$x"; python scripts/probes/trunk_times.py wg_xcd=$x 2>/dev/null | grep -E "layer3.conv|layer4.conv|layer3.0.conv1|layer4.0.conv1|layer2.0.conv1|downsample" | cut -c1-40,95-140; done
B="python bench.py --no-cpu-baseline --no-lrs-leg --profile-steps 0 --steps 40 --warmup 8"
pick2() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1 lrw ms/step', d['ms_per_step'], 'loss', d.get('final_loss'))"; }
for i in 1 2; do
for x in 1 5 2 6; do $B --tune wg_xcd=$x 2>/dev/null | pick2 "xcd=$x"; done
done
B="python bench.py --workload lrs --no-cpu-baseline --profile-steps 0 --steps 12 --warmup 3"
for i in 1 2; do
for x in 0 1 2 6; do $B --tune wg_xcd=$x 2>/dev/null | pick2 "LRS xcd=$x"; done
done
