#!/bin/bash
# round 5: the whole -m gpu suite + smoke + a bench line (advice fixes, cu-mask registry, loud encoder failure)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5b
mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -25 | tee $OUT/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>$OUT/err.txt | tail -1 > $OUT/bench.json; python -c "
import json; d=json.load(open('$OUT/bench.json')); print('LRW ms', d['ms_per_step'], 'roof', d['roofline']['kernel'], d['roofline']['frac'], 'LRS ms', d.get('lrs',{}).get('ms_per_step'))"
