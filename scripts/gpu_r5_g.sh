#!/bin/bash
# round 5: c64 epilogue straight from the accumulators — parity + timing
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_shapes.py tests/test_gpu_blockwise.py -x -q 2>&1 | tail -6
B="python bench.py --no-cpu-baseline --no-lrs-leg --profile-steps 1 --steps 40 --warmup 8"
for i in 1 2; do $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('LRW ms/step', d['ms_per_step'], 'loss', d.get('final_loss'), {k: (v['ms_per_step'], v['tflops']) for k, v in d['roofline']['per_kernel'].items() if 'c64' in k or 'p8' in k})"; done
python bench.py --workload lrs --no-cpu-baseline --profile-steps 1 --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('lrs ms/step', d['ms_per_step'], 'loss', d.get('final_loss'), {k: (v['ms_per_step'], v['launches']) for k, v in d['roofline']['per_kernel'].items() if 'c64' in k})"
