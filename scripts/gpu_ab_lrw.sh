#!/bin/bash
# same-box A/B of two builds of the library (syncvsr_amd/lib_old.bin, lib_new.bin): LRW step time (3 interleaved rounds) and the per-kernel event times
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
pick() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); pk=d.get('roofline',{}).get('per_kernel',{})
print('$1 ms/step', d['ms_per_step'], 'loss', d.get('final_loss'), ' '.join(f'{k}={v[\"ms_per_step\"]}' for k,v in pk.items() if any(s in k for s in '$2'.split(','))))"; }
for i in 1 2 3; do
for v in old new; do cp syncvsr_amd/lib_$v.bin syncvsr_amd/libsyncvsr_hip.so; python bench.py --no-cpu-baseline --no-lrs-leg --sustained-steps 0 --profile-steps 2 --steps 60 --warmup 8 2>/dev/null | pick "LRW $v" "${1:-stem}"; done
done
cp syncvsr_amd/lib_new.bin syncvsr_amd/libsyncvsr_hip.so
