#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_lrs_kernels.py -x -q 2>&1 | tail -2
B="python bench.py --workload lrs --no-cpu-baseline --profile-steps 1 --steps 12 --warmup 3"
python bench.py --workload lrs --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('lrs ms/step', d['ms_per_step'], 'host', d.get('host_enqueue_ms'), d.get('final_loss'))"
python bench.py --workload lrs --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('lrs ms/step', d['ms_per_step'], 'host', d.get('host_enqueue_ms'), d.get('final_loss'))"
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o l -- python $GRAFT_REPO_ROOT/bench.py --workload lrs --no-cpu-baseline --profile-steps 0 --steps 4 --warmup 2 > /dev/null 2>&1; grep "k_embed_pos_bwd\|k_ctc_lattice" /tmp/pp/*kernel_stats.csv | cut -c1-120
