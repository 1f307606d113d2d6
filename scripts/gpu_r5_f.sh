#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_lrs_kernels.py -x -q -k "mha or flash" 2>&1 | tail -3
ROWS=2560 python scripts/probes/lrs_lin_bench.py
python bench.py --workload lrs --no-cpu-baseline --profile-steps 1 --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('lrs ms/step', d['ms_per_step'], 'loss', d.get('final_loss'), {k: (v['ms_per_step'], v['launches']) for k, v in d['roofline']['per_kernel'].items()})"
echo "== igemm_ksplit128=12"; ROWS=2560 python scripts/probes/lrs_lin_bench.py igemm_ksplit128=12 | grep -E "attn_out|ffn2|pw1"
echo "== igemm_ksplit128=0"; ROWS=2560 python scripts/probes/lrs_lin_bench.py igemm_ksplit128=0 | grep -E "attn_out|ffn2|pw1"
python bench.py --workload lrs --no-cpu-baseline --profile-steps 1 --steps 12 --warmup 3 --tune igemm_ksplit128=12 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('ksplit128=12: lrs ms/step', d['ms_per_step'], 'loss', d.get('final_loss'))"
python bench.py --workload lrs --no-cpu-baseline --profile-steps 1 --steps 12 --warmup 3 --tune igemm_ksplit128=0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('ksplit128=0: lrs ms/step', d['ms_per_step'], 'loss', d.get('final_loss'))"
timeout 900 python -m pytest tests/test_gpu_lrs_model.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
