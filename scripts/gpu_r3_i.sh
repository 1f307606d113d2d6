#!/bin/bash
# image-block weight-gradient enumeration: parity (kernel + bench-shape + blockwise tests), A/B of the step, per-kernel table
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3i
python -m syncvsr_amd.build > /dev/null 2>&1
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_shapes.py tests/test_gpu_blockwise.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | grep -v "^  \|^ \"" | tail -15 | tee gpurun_out/r3i/tests.log
for t in "" "wg_imgmajor=0"; do
  for k in 1 2; do
    python bench.py --no-cpu-baseline --no-lrs-leg --steps 40 --warmup 10 --tune "$t" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tune=[$t]', d['ms_per_step'], d['value'], d.get('host_enqueue_ms'), d['roofline']['kernel'] if 'kernel' in d['roofline'] else '', d['roofline']['frac'])" | tee -a gpurun_out/r3i/ab.log
  done
done
python bench.py --workload lrs --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | tail -1 | cut -c1-400 | tee gpurun_out/r3i/lrs.log
python bench.py --workload lrs --no-cpu-baseline --steps 8 --warmup 3 --tune wg_imgmajor=0 2>/dev/null | tail -1 | cut -c1-400 | tee -a gpurun_out/r3i/lrs.log
bash scripts/gpu_kstats3.sh 2>&1 | grep "wgrad\|total\|stats"
