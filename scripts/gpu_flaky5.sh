#!/bin/bash
# the whole GPU suite three times in a row (fresh processes): flakiness check before the round ends
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/flaky5
for i in 1 2 3; do
  timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/flaky5/run$i.log 2>&1
  echo "run $i rc=$? $(tail -1 gpurun_out/flaky5/run$i.log)" | tee -a gpurun_out/flaky5/summary.txt
  tail -c 3000 gpurun_out/flaky5/run$i.log > gpurun_out/flaky5/run$i.tail; rm gpurun_out/flaky5/run$i.log
done
