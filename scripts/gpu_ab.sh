#!/bin/bash
# same-box A/B of two builds of the library: syncvsr_amd/lib_old.bin against syncvsr_amd/lib_new.bin (LRW step, LRS step, trunk launch times)
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
pick2() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1 ms/step', d['ms_per_step'], 'loss', d.get('final_loss'))"; }
for i in 1 2 3; do
for v in old new; do cp syncvsr_amd/lib_$v.bin syncvsr_amd/libsyncvsr_hip.so; python bench.py --no-cpu-baseline --no-lrs-leg --profile-steps 0 --steps 40 --warmup 8 2>/dev/null | pick2 "LRW $v"; done
done
for i in 1 2; do
for v in old new; do cp syncvsr_amd/lib_$v.bin syncvsr_amd/libsyncvsr_hip.so; python bench.py --workload lrs --no-cpu-baseline --profile-steps 0 --steps 12 --warmup 3 2>/dev/null | pick2 "LRS $v"; done
done
for v in old new; do cp syncvsr_amd/lib_$v.bin syncvsr_amd/libsyncvsr_hip.so; echo "== $v"; python scripts/probes/trunk_times.py 2>/dev/null | cut -c1-24,95-125; done
