#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_enc_fused.py -x -q -k "ln or enc" 2>&1 | tail -3
pick() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1 lrs ms/step', d['ms_per_step'], 'loss', d.get('final_loss'), {k: (v['ms_per_step'], v['launches']) for k, v in d['roofline']['per_kernel'].items() if 'wgrad' in k})"; }
B="python bench.py --workload lrs --no-cpu-baseline --profile-steps 1 --steps 12 --warmup 3"
$B 2>/dev/null | pick "default        "
$B --tune wg_short_k=40 2>/dev/null | pick "wg_short_k=40  "
$B --tune wg_blocks=256 2>/dev/null | pick "wg_blocks=256  "
$B --tune wg_blocks=1024 2>/dev/null | pick "wg_blocks=1024 "
echo "== lin bench wg_short_k=40"; ROWS=2560 python scripts/probes/lrs_lin_bench.py wg_short_k=40 | cut -c1-30,120-200
echo "== lin bench default"; ROWS=2560 python scripts/probes/lrs_lin_bench.py | cut -c1-30,120-200
