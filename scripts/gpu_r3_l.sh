#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3l
python -m syncvsr_amd.build > /dev/null 2>&1
python scripts/probes/p8_check.py 2>&1 | grep -v amdgpu.ids | grep "N=\|ALL\|MISMATCH\|rel diff" | tee gpurun_out/r3l/p8.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_shapes.py tests/test_gpu_blockwise.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r3l/tests.log
for k in 1 2; do
python bench.py --no-cpu-baseline --no-lrs-leg --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('host_enqueue_ms'), d['roofline'].get('kernel'), d['roofline']['frac'])
for k,v in d['roofline']['per_kernel'].items():
    if 'p8' in k: print('   ', k, v)" | tee -a gpurun_out/r3l/bench.log
done
