#!/bin/bash
# quick GPU loop: parity tests + a short bench (no CPU baseline)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x -p no:cacheprovider > gpurun_out/tests.log 2>&1
echo "tests exit $?" >> gpurun_out/tests.log
tail -n 12 gpurun_out/tests.log
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
echo "bench exit $?"; tail -n 3 gpurun_out/bench_quick.err; python - <<'PY'
import json
try:
    r=json.load(open('gpurun_out/bench_quick.json'))
    print({k:r[k] for k in ('value','ms_per_step','step_mfma_frac','final_loss')})
    for k,v in r['roofline']['per_kernel'].items(): print(' ',k,v)
except Exception as e: print('no bench json', e)
PY
