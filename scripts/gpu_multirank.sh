#!/bin/bash
# Control-flow check of the multi-rank bench on a ONE-GPU box: 2 ranks share GPU 0, gloo carries the collectives (RCCL refuses two ranks
# on one device).  Not a measurement — it proves that no rank waits for a collective the others never issue.
mkdir -p gpurun_out
export SVSR_BENCH_ONE_DEVICE=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 4 --warmup 2 --batch 8 --backend gloo --no-cpu-baseline > gpurun_out/multirank.log 2>&1
echo "exit $?"
tail -1 gpurun_out/multirank.log | cut -c1-700
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 2 --steps 3 --warmup 1 --batch 4 --backend gloo --no-cpu-baseline --workload lrs --frames 40 > gpurun_out/multirank_lrs.log 2>&1
echo "exit $?"
tail -1 gpurun_out/multirank_lrs.log | cut -c1-400
