#!/bin/bash
# fused encoder backward: A/B on one box + in-line kernel time of the launch
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
B="python bench.py --no-cpu-baseline --no-lrs-leg --steps 60 --warmup 8"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); pk=d.get('roofline',{}).get('per_kernel',{}); print(sys.argv[1], d['ms_per_step'], 'launches', d.get('launches_per_step'), 'enc_fwd', pk.get('k_enc_fwd',{}).get('ms_per_step'), 'enc_bwd', pk.get('k_enc_bwd',{}).get('ms_per_step'), d.get('final_loss'))" "$1"; }
$B 2>/dev/null | pick "fused bwd   "
SVSR_ENC_BWD_FUSED=0 $B 2>/dev/null | pick "chain bwd   "
$B 2>/dev/null | pick "fused bwd   "
SVSR_ENC_BWD_FUSED=0 $B 2>/dev/null | pick "chain bwd   "
