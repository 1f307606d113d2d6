#!/bin/bash
# round 5 baseline at session start: default bench line (LRW + LRS leg), then in-line per-kernel stats of both steps
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r5base
python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>gpurun_out/r5base/err.txt | tail -1 > gpurun_out/r5base/bench.json
python -c "
import json; d=json.load(open('gpurun_out/r5base/bench.json')); print('LRW ms', d['ms_per_step'], 'launches', d['launches_per_step'], 'roof', d['roofline']['kernel'], d['roofline']['frac'], 'LRS ms', d.get('lrs',{}).get('ms_per_step'))"
./scripts/gpu_kstats.sh --no-lrs-leg 2>&1 | tail -60
cp gpurun_out/kstats/*kernel_stats.csv gpurun_out/r5base/lrw_kernel_stats.csv 2>/dev/null || cp $(find gpurun_out/kstats -name "*kernel_stats.csv" | head -1) gpurun_out/r5base/lrw_kernel_stats.csv
./scripts/gpu_r5_d.sh 2>&1 | tail -62
