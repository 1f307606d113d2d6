"""The multi-rank benchmark flow on ONE GPU (-m gpu): two ranks share cuda:0 and gloo carries the collectives (RCCL refuses two ranks
on one device; the 1-rank RCCL path is covered by tests/test_gpu_train.py).  This is the driver's own launch line for N > 1 — parameter
and buffer broadcast, gradient buckets all-reduced from inside the real backward, per-step BatchNorm broadcast, rank-mean metrics, the
rank-0-only profiling leg, the closing barrier: no rank may wait for a collective the others never issue."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [["--batch", "4"], ["--batch", "2", "--workload", "lrs", "--frames", "32"]])
def test_two_rank_bench_completes(extra):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    env = dict(os.environ, SVSR_BENCH_ONE_DEVICE="1", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--no-cpu-baseline", *extra]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["steps"] == 3
    assert d["collective"]["ranks"] == 2 and d["collective"]["all_reduce_launches_per_step"] >= 2
    assert d["value"] > 0 and d["final_loss"] == d["final_loss"]          # finite
    # what a first real 8-GPU run diagnoses itself with: the join every rank sat in, and every bucket's size / start / duration on rank 0
    c = d["collective"]
    assert len(c["exposed_join_ms_per_rank"]) == 2 and len(c["host_enqueue_ms_per_rank"]) == 2
    b = c["buckets_rank0"]
    assert len(b) == c["all_reduce_launches_per_step"] and all(x["mb"] > 0 and x["ms"] >= 0 and x["ready_ms"] >= 0 for x in b), b
    assert abs(sum(x["mb"] for x in b) - c["gradient_mb_per_step"]) < 0.1 * len(b) + 0.5, (b, c["gradient_mb_per_step"])
    assert b[0]["ready_ms"] == 0.0 and all(y["ready_ms"] >= x["ready_ms"] for x, y in zip(b, b[1:])), b
    assert "roofline" in d and d["roofline"]["per_kernel"]
