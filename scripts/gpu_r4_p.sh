#!/bin/bash
# p8: epilogue operands touched ahead (p8_stagger bit 2) — A/B
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
cat > /tmp/ab.py <<'PY'
import json, subprocess, sys, os
from syncvsr_amd import ops
PY
B="python bench.py --no-cpu-baseline --no-lrs-leg --steps 60 --warmup 8"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; pk=r['per_kernel']; a=pk.get('k_igemm_p8<256,128,3>',{}); b=pk.get('k_igemm_p8<256,64,3>',{}); print(sys.argv[1], d['ms_per_step'], 'frac', r['frac'], r['kernel'], 'p8_128 ms', a.get('ms_per_step'), 'plain', a.get('plain_tflops'), 'bn', a.get('bn_epilogue_tflops'), '| p8_64 ms', b.get('ms_per_step'), 'bn', b.get('bn_epilogue_tflops'), d.get('final_loss'))" "$1"; }
$B --tune p8_stagger=7 2>/dev/null | pick "touch ahead "
$B --tune p8_stagger=3 2>/dev/null | pick "no touch    "
$B --tune p8_stagger=7 2>/dev/null | pick "touch ahead "
$B --tune p8_stagger=3 2>/dev/null | pick "no touch    "
