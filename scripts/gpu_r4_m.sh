#!/bin/bash
# hand-over events without the system fence / batch resident in the step's input buffers: A/B on one box
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
B="python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 8"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d['ms_per_step'], 'lrs', d.get('lrs',{}).get('ms_per_step'), 'host', d.get('host_enqueue_ms'), d.get('final_loss'))" "$1"; }
$B 2>/dev/null | pick "default            "
SVSR_EVENT_SYSTEM_FENCE=1 $B 2>/dev/null | pick "system-fence events"
$B --staged-inputs 2>/dev/null | pick "staged inputs      "
SVSR_SIDE_GROUP=2 $B 2>/dev/null | pick "side group 2       "
$B 2>/dev/null | pick "default again      "
