#!/bin/bash
# the whole -m gpu suite + smoke, as the driver runs it at round end
OUT=$GRAFT_REPO_ROOT/gpurun_out/tests_all
mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 | tee $OUT/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
