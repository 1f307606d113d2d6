"""Data-parallel semantics across RANKS (-m gpu; SURVEY.md section 4 iv): the 2-rank training trajectory against its definition, and an
all-reduce self-test with a rank pattern through the reducer's own bucket path.

Reference: Lightning strategy="ddp" (LRW/video/src/train.py:28; LRS/video/main.py:38) = torch DistributedDataParallel: every rank runs
forward/backward on its shard (BatchNorm statistics are per rank: there is no SyncBatchNorm in the reference), gradients are averaged,
every rank applies the same optimiser step, buffers follow rank 0.  The definition is evaluated in ONE process: two replicas, one per
shard, gradients averaged by hand as (g0 + g1) / 2 — one rounding, the same on both sides — so the 2-rank run must reproduce it BIT FOR
BIT (losses of both ranks at every step, final parameters).

With two or more GPUs the ranks run on one GPU each over RCCL (backend "nccl": bucketed all-reduce on the comm stream, overlapped with the
backward); on a single-GPU box both ranks share cuda:0 and gloo carries the collectives — same control flow, same assertions."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
STEPS = 3


def _case():
    from golden_cases import build_case

    cfg, sd, batch, training, gold = build_case("lrw_tiny")
    cfg.optim.scheduler.num_warmup_steps = 1
    cfg.optim.scheduler.num_training_steps = 10
    cfg.optim.optimizer.lr = 2e-4
    return cfg, sd, batch


def _shard(batch, rank, world):
    n = batch[0].shape[0] // world
    return [t[rank * n:(rank + 1) * n].contiguous() for t in batch]


def _worker():
    """One rank (spawned by the test): self-test of the bucket all-reduce, then STEPS training steps on this rank's shard."""
    import torch.distributed as dist

    sys.path.insert(0, HERE)
    from syncvsr_amd.engine import TrainStep
    from syncvsr_amd.model import Model

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ["SVSR_TEST_BACKEND"]
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, sd, batch = _case()
    model = Model(cfg, seed=100 + rank)               # different seeds: the reducer must bring rank 0's parameters everywhere
    if rank == 0:
        model.load_state_dict(sd)
    model.to(dev).train()
    comm = torch.bfloat16 if os.environ.get("SVSR_TEST_COMM", "fp32") == "bf16" else torch.float32
    ts = TrainStep(model, cfg, bucket_mb=0.25, grad_comm_dtype=comm)
    st = model.store()
    out = {"rank": rank}
    # -- all-reduce self-test: gradient buffer = (rank + 1) * (1 + i mod 7): after the reducer it must hold the rank mean of that
    idx = torch.arange(st.numel, device=dev, dtype=torch.float32) % 7 + 1
    st.grad.copy_((rank + 1) * idx)
    ts.dp.begin_step()
    ts.dp.on_ready(st.decay_end // 2)                 # a partial notification first, then the flush — as the backward issues them
    ts.dp.on_ready(0)
    ts.dp.finish()
    torch.cuda.synchronize()
    want = idx * (sum(range(1, world + 1)) / world)
    out["selftest_ok"] = bool(torch.equal(st.grad, want))
    out["buckets"] = len(ts.dp.launched)
    covered = sorted(ts.dp.launched)
    out["covered"] = covered[0][0] == 0 and covered[-1][1] == st.numel and all(a[1] <= b[0] for a, b in zip(covered, covered[1:]))
    if comm == torch.bfloat16:
        # random gradients through the bf16 wire format against the hand-averaged definition: each rank's values are rounded to bf16 once
        # (bf16 keeps 8 significant bits: unit roundoff 2^-8), their sum is rounded once more: |result - (g0 + g1) / 2| <= 2^-7 (|g0| + |g1|) / 2
        # element by element (stated bound; measured 0.98 of it)
        gs = [torch.randn(st.numel, generator=torch.Generator().manual_seed(4242 + r)).to(dev) for r in range(world)]
        st.grad.copy_(gs[rank])
        ts.dp.begin_step()
        ts.dp.on_ready(0)
        ts.dp.finish()
        torch.cuda.synchronize()
        mean = sum(gs) / world
        bound = sum(g.abs() for g in gs) / world * 2.0 ** -7 + 1e-30
        out["bf16_wire_max_err_over_bound"] = float(((st.grad - mean).abs() / bound).max())
    # -- training on this rank's shard
    mine = [t.to(dev) for t in _shard(batch, rank, world)]
    losses = []
    for _ in range(STEPS):
        o = ts.step(*mine)
        losses.append(float(o["loss_total"].item()))
    torch.cuda.synchronize()
    out["losses"] = losses
    torch.save({"flat": st.flat.detach().cpu(), "bufflat": st.bufflat.detach().cpu()}, os.environ["SVSR_TEST_OUT"] + f".rank{rank}.pt")
    json.dump(out, open(os.environ["SVSR_TEST_OUT"] + f".rank{rank}.json", "w"))
    dist.barrier()
    dist.destroy_process_group()


def _definition(dev):
    """Two replicas in one process, gradients averaged by hand: what 2-rank DistributedDataParallel training computes."""
    from syncvsr_amd import ops
    from syncvsr_amd.engine import TrainStep
    from syncvsr_amd.model import Model

    cfg, sd, batch = _case()
    reps = []
    for r in range(2):
        m = Model(cfg, seed=100)
        m.load_state_dict(sd)
        m.to(dev).train()
        reps.append((m, TrainStep(m, cfg, data_parallel=False), [t.to(dev) for t in _shard(batch, r, 2)]))
    losses = [[], []]
    for _ in range(STEPS):
        for r, (m, ts, b) in enumerate(reps):
            o = m(*b)
            o["loss_total"].backward()
            losses[r].append(float(o["loss_total"].item()))
        g = (reps[0][0].store().grad + reps[1][0].store().grad) / 2
        for m, ts, b in reps:
            st = m.store()
            st.grad.copy_(g)
            ops.grad_sumsq(st.grad, ts.opt_state)
            ops.adamw_step(st.flat, st.grad, ts.m, ts.v, st.w16, st.decay_end, ts.lr, ts.betas, ts.eps, ts.weight_decay, ts.max_norm, ts.warmup,
                           ts.total_steps, ts.opt_state)
            ops.transpose_shadows(st.flat, st.w16, st.w16t, st.table, st.n_entries)
            st.shadow_fresh = True
        reps[1][0].store().bufflat.copy_(reps[0][0].store().bufflat)          # broadcast_buffers: rank 0's running statistics
    torch.cuda.synchronize()
    st0 = reps[0][0].store()
    return losses, st0.flat.detach().cpu(), st0.bufflat.detach().cpu()


def _run_two_ranks(comm: str):
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    tmp = tempfile.mkdtemp(prefix="svsr_ddp_")
    base = os.path.join(tmp, "out")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   SVSR_TEST_BACKEND=backend, SVSR_TEST_OUT=base, SVSR_TEST_COMM=comm, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4",
                   PYTHONPATH=os.pathsep.join([ROOT, HERE, os.environ.get("PYTHONPATH", "")]))
        code = "import test_gpu_ddp_ranks as t; t._worker()"
        procs.append(subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=900)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-4000:]
    res = [json.load(open(base + f".rank{r}.json")) for r in range(2)]
    for r in res:
        assert r["selftest_ok"], "bucket all-reduce: rank-pattern self-test failed"
        assert r["buckets"] >= 3 and r["covered"], (r["buckets"], "the buckets must tile the gradient buffer")
    got = [torch.load(base + f".rank{r}.pt") for r in range(2)]
    return backend, res, got


def test_two_ranks_reproduce_the_data_parallel_definition():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    backend, res, got = _run_two_ranks("fp32")
    want_losses, want_flat, want_buf = _definition(torch.device("cuda:0"))
    print(backend, "ranks' losses", [r["losses"] for r in res], "definition", want_losses)
    for r in range(2):
        assert res[r]["losses"] == want_losses[r], (backend, r, res[r]["losses"], want_losses[r])
    assert torch.equal(got[0]["flat"], got[1]["flat"]), "the ranks' parameters diverged"
    assert torch.equal(got[0]["flat"], want_flat), f"{int((got[0]['flat'] != want_flat).sum())} parameters differ from the definition"
    assert torch.equal(got[0]["bufflat"], want_buf) and torch.equal(got[1]["bufflat"], want_buf), "running statistics do not follow rank 0"


def test_two_ranks_with_the_bf16_gradient_wire_format():
    """GradReducer(comm_dtype=torch.bfloat16) (engine.py; bench.py --grad-comm bf16): the buckets cross the links as bf16 and come back into
    the fp32 gradient buffer — an opt-in deviation from DDP's fp32 all-reduce.  The rank-pattern self-test uses
    values bf16 represents exactly, so it must still be EXACT; the 3-step trajectory is held to the fp32 definition with a stated tolerance:
    every gradient element is rounded to 8 mantissa bits once per step (relative 2^-9 = 2e-3 per element, uncorrelated), which AdamW's
    normalisation passes on to the update — losses within 5e-3 relative (measured 1.0e-3 at the third step of this ill-conditioned 10-frame
    batch, whose fp32-vs-bf16 trajectories the oracle tests hold to 1e-2); a random gradient through the reducer stays within the stated
    element-wise bound 2^-7 (|g0| + |g1|) / 2 of the hand average, and the
    two ranks still bit-identical to each other (they apply the same reduced gradient)."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    backend, res, got = _run_two_ranks("bf16")
    want_losses, want_flat, want_buf = _definition(torch.device("cuda:0"))
    cfg, sd, batch = _case()
    from syncvsr_amd.model import Model

    m0 = Model(cfg, seed=100)
    m0.load_state_dict(sd)
    flat0 = m0.to("cuda:0").store().flat.detach().cpu()
    print(backend, "bf16 wire: ranks' losses", [r["losses"] for r in res], "fp32 definition", want_losses)
    for r in range(2):
        for a, b in zip(res[r]["losses"], want_losses[r]):
            assert abs(a - b) <= 5e-3 * abs(b), (r, res[r]["losses"], want_losses[r])
    assert torch.equal(got[0]["flat"], got[1]["flat"]), "the ranks' parameters diverged"
    for r in res:
        assert r["bf16_wire_max_err_over_bound"] <= 1.0, r["bf16_wire_max_err_over_bound"]
    upd, want_upd = got[0]["flat"] - flat0, want_flat - flat0
    # (AdamW normalises every element's update to ~lr: elements whose two ranks' gradients nearly cancel change sign under the rounding, so the
    # update is reported, not bounded — measured 9 % after three steps of this 10-frame batch; the gradient itself is bounded above)
    print("relative L2 difference of the 3-step parameter update under the bf16 wire format:", float((upd - want_upd).norm() / want_upd.norm()))
    assert torch.equal(got[0]["bufflat"], got[1]["bufflat"]), "running statistics do not follow rank 0"
