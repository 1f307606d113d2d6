#!/usr/bin/env python
"""Headline benchmark: SyncVSR LRW training throughput (lip-clips/s) on MI355X, BASELINE.json configs[1]/[2].

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = forward + backward + (RCCL gradient all-reduce) + global-norm clip + AdamW on a synthetic batch of 32
clips of 29x88x88 per GPU; eager launches by default, `--graph` replays one HIP graph per step.  Rank 0 prints ONE JSON line.  Besides the
contract fields it carries:
  roofline      — dominant kernel of the step (by summed HIP-event time over profiled eager steps), its algorithmic
                  FLOPs / time against the dense bf16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md), plus the per-kernel table
  cpu_baseline  — oracle/lrw_oracle.py (torch fp32 port of the reference path) timed on this box's host cores, N=1 only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # before the HIP runtime starts: see syncvsr_amd/__init__.py

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAIN_FLOP_PER_CLIP = 56.9e9        # fwd + dgrad + wgrad, 29x88x88 clip, 6L-512d encoder
MFMA_PEAK_BF16 = 2.5e15             # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def physical_cores():
    """Physical cores of the host (lscpu: sockets x cores per socket), None when lscpu is not there."""
    try:
        import subprocess

        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        vals = {}
        for ln in txt.splitlines():
            k, _, v = ln.partition(":")
            vals[k.strip()] = v.strip()
        return int(vals["Socket(s)"]) * int(vals["Core(s) per socket"])
    except Exception:
        return None


def cpu_baseline(cfg, batch_size: int, budget_s: float = 20.0, min_timed: int = 5, threads: int = 32) -> dict:
    """Times the CPU port (oracle) of the same training step — forward, backward, clip, AdamW — on the host cores."""
    from oracle import lrw_oracle as O
    from syncvsr_amd.init import init_state_dict, synthetic_batch

    # cores this process may actually run on (the box reports 256 logical CPUs; oversubscribing them with one torch thread
    # each made a step take minutes); the reported value uses 32 threads, where the fp32 convolutions stop scaling, and a second
    # leg (`all_cores` in the line) uses every physical core as SURVEY section 8d asks
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    cores = max(1, min(threads, allowed))
    torch.set_num_threads(cores)
    sd = init_state_dict(cfg, seed=0)
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    for k in names:
        sd[k].requires_grad_(True)
    params = [sd[k] for k in names]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    batch = synthetic_batch(cfg, batch_size, seed=1234)
    opt = cfg.optim.optimizer
    times = []
    t_start = time.perf_counter()
    step = 0
    while True:
        t0 = time.perf_counter()
        for p in params:
            p.grad = None
        out = O.forward(sd, cfg, *batch, training=True)
        out["loss_total"].backward()
        with torch.no_grad():
            grads = [p.grad for p in params]
            O.clip_grad_norm(grads, float(cfg.train.gradient_clip_val))
            O.adamw_step(params, grads, m, v, step + 1, O.cosine_lr(step, float(opt.lr), 15000, 270000), tuple(opt.betas),
                         float(opt.eps), float(opt.weight_decay))
        times.append(time.perf_counter() - t0)
        step += 1
        # a median of at least `min_timed` steps after one untimed warm-up step; the time budget only stops it early
        # on a host so slow that the default run would not finish in minutes
        if step >= min_timed + 1 and (time.perf_counter() - t_start > budget_s or step >= 50):
            break
        if time.perf_counter() - t_start > 6 * budget_s and step >= 3:
            break
    steady = sorted(times[1:] or times)
    med = steady[len(steady) // 2]
    return {"value": batch_size / med, "unit": "clips/s", "cores": cores, "kind": "port", "host_physical_cores": physical_cores(),
            "host_logical_cpus": os.cpu_count(),
            "sample": f"median of {len(steady)} timed training steps after 1 warm-up (fwd+bwd+clip+AdamW, fp32) of oracle/lrw_oracle.py at batch "
                      f"{batch_size} x 29x88x88: {med * 1e3:.0f} ms per step, {cores} torch threads"}


def lrs_train_flops(B: int, T: int, L: int) -> float:
    """Algorithmic FLOPs of one LRS training step: per frame 0.4996 GMAC forward (front-end 0.3162, embed,
    Conformer 0.1772, audio + CTC heads), attention 36.9 k*T^2 MAC per clip, decoder 6*[(L+1)*8.26 M + T*1.18 M] + (L+1)*3.88 M MAC
    per clip; training = forward + data-gradient + weight-gradient (the stem, 30.4 MMAC/frame, has no data-gradient)."""
    fwd_mac = B * (T * 0.4996e9 + 36.9e3 * T * T + 6 * ((L + 1) * 8.26e6 + T * 1.18e6) + (L + 1) * 3.88e6)
    return 2.0 * (3.0 * fwd_mac - B * T * 30.4e6)


def cpu_baseline_lrs(lrs_args, odim: int, frames: int, budget_s: float = 20.0) -> dict:
    """CPU port (oracle/lrs_oracle.py) of the LRS training step on one clip of `frames` frames."""
    from oracle import lrs_oracle as OS
    from oracle import lrw_oracle as O
    from syncvsr_amd.lrs_init import lrs_init_state_dict, lrs_synthetic_batch

    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    cores = max(1, min(32, allowed))
    torch.set_num_threads(cores)
    sd = lrs_init_state_dict(lrs_args, odim, seed=0)
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    for k in names:
        sd[k].requires_grad_(True)
    params = [sd[k] for k in names]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    batch = lrs_synthetic_batch(lrs_args, 1, frames, odim=odim, seed=1234, label_len=(5, 20))
    times = []
    t_start = time.perf_counter()
    step = 0
    while True:
        t0 = time.perf_counter()
        for p in params:
            p.grad = None
        out = OS.forward(sd, lrs_args, *batch, training=True)
        out["loss"].backward()
        with torch.no_grad():
            grads = [p.grad for p in params]
            O.clip_grad_norm(grads, 5.0)
            O.adamw_step(params, grads, m, v, step + 1, O.cosine_lr(step, 1e-3, 25000, 500000), (0.9, 0.98), 1e-6, 0.03)
        times.append(time.perf_counter() - t0)
        step += 1
        if (time.perf_counter() - t_start > budget_s and step >= 2) or step >= 20 or time.perf_counter() - t_start > 4 * budget_s:
            break
    steady = sorted(times[1:] or times)
    med = steady[len(steady) // 2]
    return {"value": frames / med, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{step} training steps (fwd+bwd+clip+AdamW, fp32) of oracle/lrs_oracle.py on one {frames}-frame 88x88 clip, "
                      f"median step {med * 1e3:.0f} ms, {cores} torch threads"}


def pmc_traffic(kernel_label: str, lrs: bool = False):
    """HBM bytes per launch of `kernel_label` from the committed rocprofv3 PMC passes of the newest round (profiles/roundN_pmc_per_kernel.json,
    roundN_pmc_lrs_per_kernel.json for the sentence-level step; made by scripts/gpu_profiles_roundN.sh + scripts/collect_profiles.py at the
    commit recorded in their __meta__: FETCH_SIZE and WRITE_SIZE in KiB, separate --pmc runs).  gfx950 correction per MI355X_MICROARCH.md
    section HBM: FETCH_SIZE reports half of the bytes of wide coalesced reads, so it is doubled.  None when no PMC record exists for the
    kernel.  The bench label k_igemm_p8<256,128,3> covers the profiler's instantiations k_igemm_p8<0|1, PH, false> (plain / BatchNorm-backward
    epilogue): their launch-weighted mean; k_igemm_wgrad<BC,NS> covers k_igemm_wgrad and k_igemm_wgrad_units of that tile."""
    import glob
    import re

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_lrs_per_kernel.json" if lrs else "round*_pmc_per_kernel.json")),
                   key=lambda f: int(re.search(r"round(\d+)_", os.path.basename(f)).group(1)))
    if not files:
        return None
    path = files[-1]
    try:
        rec = json.load(open(path))
    except OSError:
        return None
    key = kernel_label.replace(",", ", ")
    keys = {key, key[:-1] + ", 1>"}          # k_igemm_fwd_glds carries a defaulted fourth template argument (K groups) in the profiler's name
    if kernel_label.startswith("k_igemm_wgrad<"):
        keys.add(key.replace("k_igemm_wgrad<", "k_igemm_wgrad_units<"))
    tot_b = tot_n = 0.0
    for name, v in rec.items():
        if name == "__meta__" or "FETCH_SIZE_avg_per_dispatch" not in v or "WRITE_SIZE_avg_per_dispatch" not in v:
            continue
        plain = name.replace("void ", "")
        if plain in keys or (kernel_label.startswith("k_igemm_p8<") and plain.startswith("k_igemm_p8<")):
            n = float(v.get("dispatches_FETCH_SIZE", 1))
            tot_b += n * (2.0 * v["FETCH_SIZE_avg_per_dispatch"] + v["WRITE_SIZE_avg_per_dispatch"]) * 1024.0
            tot_n += n
    if tot_n == 0:
        return None
    return {"bytes_per_launch": round(tot_b / tot_n),
            "source": f"profiles/{os.path.basename(path)} @ {rec.get('__meta__', {}).get('commit', '?')[:12]} (rocprofv3 --pmc, FETCH_SIZE doubled for gfx950)"}


def rocprof_avg_us(kernel_label: str, lrs: bool = False):
    """Average launch duration (us) of `kernel_label` in the committed rocprofv3 --kernel-trace --stats summary of the newest round
    (profiles/roundN_lrw_kernel_stats.csv / roundN_lrs_kernel_stats.csv, every launch in line), launch-weighted over the profiler's
    instantiations of the label (the label's <BM,BN,NS> are the tile, the profiler's template arguments are <EPI,PH,NJ>: k_igemm_p8<256,128,3> =
    k_igemm_p8<*, *, 2, *>, k_igemm_p8<256,64,3> = k_igemm_p8<*, *, 1, *>; k_igemm_wgrad<BC,NS> = k_igemm_wgrad + k_igemm_wgrad_units of that tile).
    None when the summary has no such kernel.  -> (us, source)"""
    import csv
    import glob
    import io
    import re

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_lrs_kernel_stats.csv" if lrs else "round*_lrw_kernel_stats.csv")),
                   key=lambda f: int(re.search(r"round(\d+)_", os.path.basename(f)).group(1)))
    if not files:
        return None
    path = files[-1]
    try:
        lines = open(path).read().splitlines()
    except OSError:
        return None
    head = next((ln for ln in lines if ln.startswith("#")), "")
    rows = list(csv.DictReader(io.StringIO("\n".join(ln for ln in lines if not ln.startswith("#")))))
    base = kernel_label.split("<")[0]
    targs = [a.strip() for a in kernel_label[len(base) + 1:-1].split(",")] if "<" in kernel_label else []

    def match(name: str) -> bool:
        plain = name.replace("void ", "").split("(")[0]
        b = plain.split("<")[0]
        a = [x.strip() for x in plain[len(b) + 1:-1].split(",")] if "<" in plain else []
        if base == "k_igemm_p8":
            return b == base and len(a) >= 3 and a[2] == ("2" if targs[1] == "128" else "1")
        if base == "k_igemm_wgrad":
            return b in ("k_igemm_wgrad", "k_igemm_wgrad_units") and a[: len(targs)] == targs
        if b != base:
            return False
        return not targs or a[: len(targs)] == targs

    tot_ns = tot_n = 0.0
    for r in rows:
        if match(r["Name"]):
            tot_ns += float(r["TotalDurationNs"])
            tot_n += float(r["Calls"])
    if tot_n == 0:
        return None
    m = re.search(r"commit ([0-9a-f]{7,40})", head)
    return tot_ns / tot_n / 1e3, f"profiles/{os.path.basename(path)} @ {(m.group(1)[:12] if m else '?')} (rocprofv3 --kernel-trace --stats, every launch in line)"


def build_lrs(args, dev, world: int, rank: int):
    """The LRS workload (BASELINE configs[3]/[4]): E2E at the shipped config, one length-bucketed batch of the longest bucket.
    -> (model, train config, device batch, lrs args, valid frames on this rank, label length); sets args.frames to the padded length."""
    from syncvsr_amd.engine import lrs_train_config
    from syncvsr_amd.lrs_data import LengthBucketBatchSampler, reference_length_histogram
    from syncvsr_amd.lrs_init import LRS_ODIM, default_lrs_args, lrs_synthetic_batch
    from syncvsr_amd.lrs_model import E2E

    lrs_args = default_lrs_args(dropout_rate=args.dropout, transformer_attn_dropout_rate=args.dropout)
    cfg = lrs_train_config()
    model = E2E(LRS_ODIM, lrs_args, seed=0).to(dev).train()
    model.reseed_dropout(1000 + rank)
    batch, n_frames, label_len = lrs_bucket_batch(args, dev, lrs_args, world, rank)
    return model, cfg, batch, lrs_args, n_frames, label_len


def lrs_bucket_batch(args, dev, lrs_args, world: int, rank: int):
    """One length-bucketed batch: the longest bucket of the reference's length histogram rescaled to --frames.  Sets args.frames to the padded
    length.  -> (device batch, valid frames on this rank, label length)"""
    from syncvsr_amd.lrs_data import LengthBucketBatchSampler, reference_length_histogram
    from syncvsr_amd.lrs_init import lrs_synthetic_batch

    # BASELINE configs[4]: length-bucketed batches — every rank draws its clips from the same length bucket, so all ranks pad
    # to the same number of frames in a step (syncvsr_amd/lrs_data.py; the hook is reference datamodule/data_module.py:66-74)
    pool = reference_length_histogram(4096, seed=7) * args.frames // 155            # the reference's length histogram, rescaled to --frames
    sampler = LengthBucketBatchSampler(pool, args.lrs_batch, world, rank, width=16, seed=11)
    step_idx = max(range(len(sampler)), key=lambda i: sampler.padded_frames()[i])   # time the longest bucket (the padded length --frames names)
    mine = list(sampler)[step_idx]
    args.frames = sampler.padded_frames()[step_idx]
    cpu_batch = lrs_synthetic_batch(lrs_args, args.lrs_batch, args.frames, seed=1234 + rank, lengths=pool[mine])
    batch = [t.to(dev) for t in cpu_batch]
    return batch, int(cpu_batch[1].sum()), cpu_batch[3].shape[-1]


def host_idle_queue_ms(trainer, batch, reps: int = 3) -> float:
    """Host time of one step's enqueue onto an EMPTY queue (median of `reps`): inside the timed loop the host runs ahead of the GPU until
    the launch queue is full and then waits for slots, so the in-loop figure of a step with thousands of launches measures the GPU."""
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trainer.step(*batch)
        ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    return sorted(ts)[len(ts) // 2]


def in_place_batch(trainer, batch, args):
    """The synthetic batch where a recorded step reads it: TrainStep.input_buffers() are the tensors a loader's host-to-device copy
    targets, and the warm-up steps have left this batch in them — so the timed steps start from inputs resident in HBM at the
    address the step consumes, without the device-to-device staging copy of a batch that lives somewhere else (--staged-inputs
    keeps that copy inside the step)."""
    bufs = trainer.input_buffers()
    if bufs is None or args.staged_inputs:
        return batch
    return tuple(b if b is not None else orig for b, orig in zip(bufs, batch))


def lrs_leg(args, dev, with_cpu: bool = False) -> dict:
    """A short, bounded LRS measurement attached to the default (LRW) line, so that BASELINE configs[3] gets a driver-timed number:
    warm-up, `args.lrs_steps` timed steps (barrier + synchronize on both sides), then one eager step with per-launch HIP events."""
    from syncvsr_amd import ops
    from syncvsr_amd.engine import TrainStep

    model, cfg, batch, lrs_args, n_frames, label_len = build_lrs(args, dev, 1, 0)
    native = args.enqueue == "native"
    trainer = TrainStep(model, cfg, native=native)
    for _ in range(3):
        trainer.step(*batch)
    batch = in_place_batch(trainer, batch, args)
    torch.cuda.synchronize()
    trainer.host_ms.clear()
    t0 = time.perf_counter()
    for _ in range(args.lrs_steps):
        out = trainer.step(*batch)
    host_ms = sorted(trainer.host_ms)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = elapsed / args.lrs_steps * 1e3
    host_idle = host_idle_queue_ms(trainer, batch)
    flops = lrs_train_flops(args.lrs_batch, args.frames, label_len)
    prof = TrainStep(model, cfg, data_parallel=False)          # eager in-line steps with per-launch HIP events
    model._side.enabled = model._side.enabled_small = False
    prof._step_impl(*batch)
    ops.start_event_timing()
    prof._step_impl(*batch)
    table = ops.stop_event_timing()
    rows = {}
    for k, v in table.items():          # "+bn" launches are counted with their kernel
        if v["flops"] > 0:
            r = rows.setdefault(k[:-3] if k.endswith("+bn") else k, dict(launches=0, ms=0.0, flops=0.0))
            for f in ("launches", "ms", "flops"):
                r[f] += v[f]
    dom = max(rows, key=lambda k: rows[k]["ms"])
    d = rows[dom]
    ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
    cpu = None
    if with_cpu:
        from syncvsr_amd.lrs_init import LRS_ODIM

        cpu = cpu_baseline_lrs(lrs_args, LRS_ODIM, 32, budget_s=12.0)
    rp = rocprof_avg_us(dom, lrs=True)
    n_launches = int(trainer._rec.size) if getattr(trainer, "_rec", None) is not None else None
    # BASELINE configs[3] says "<= 400 frames" and SURVEY section 8(d) asks for the length histogram rescaled to that: a second, shorter timing
    # of the SAME model on the longest bucket of the histogram rescaled to 400 frames (3 steps after 2 warm-up steps; no profile leg)
    lrs400 = None
    if not args.no_lrs400:
        try:
            del prof, trainer
            torch.cuda.empty_cache()
            a400 = argparse.Namespace(**vars(args))
            a400.frames = 400
            model._side.enabled = model._side.enabled_small = True
            batch400, n400, lab400 = lrs_bucket_batch(a400, dev, lrs_args, 1, 0)
            t400 = TrainStep(model, cfg, native=native)
            for _ in range(2):
                t400.step(*batch400)
            batch400 = in_place_batch(t400, batch400, a400)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                o400 = t400.step(*batch400)
            torch.cuda.synchronize()
            ms400 = (time.perf_counter() - t0) / 3 * 1e3
            lrs400 = {"ms_per_step": round(ms400, 3), "steps": 3, "per_gpu_batch": args.lrs_batch, "padded_frames": a400.frames, "valid_frames": n400,
                      "clips_per_s": round(args.lrs_batch * 1e3 / ms400, 2), "padded_frames_per_s": round(args.lrs_batch * a400.frames * 1e3 / ms400, 1),
                      "step_mfma_frac": round(lrs_train_flops(args.lrs_batch, a400.frames, lab400) / (ms400 * 1e-3) / MFMA_PEAK_BF16, 5),
                      "final_loss": round(float(o400[0].item()), 4),
                      "workload": f"the same model and step on the longest bucket of the reference's length histogram rescaled to <= 400 frames (padded to {a400.frames})"}
        except Exception as e:          # (the 160-frame figure must survive a failure of the longer one)
            lrs400 = {"error": f"{type(e).__name__}: {e}"}
    return {
        **({"lrs400": lrs400} if lrs400 is not None else {}),
        **({"cpu_baseline": cpu} if cpu is not None else {}),
        "metric": f"lip-clips/sec training (LRS, <= {args.frames}x88x88)", "value": round(args.lrs_batch * 1e3 / ms, 2), "unit": "clips/s",
        "ms_per_step": round(ms, 3), "steps": args.lrs_steps, "padded_frames_per_s": round(args.lrs_batch * args.frames * 1e3 / ms, 1),
        "host_enqueue_ms": round(host_idle, 3), "host_enqueue_in_loop_ms": round(host_ms[len(host_ms) // 2], 3) if host_ms else None,
        "launches_per_step": n_launches,
        "step_mfma_frac": round(flops / (ms * 1e-3) / MFMA_PEAK_BF16, 5), "final_loss": round(float(out[0].item()), 4),
        "config": {"workload": "LRS training step (fwd+bwd+clip+AdamW): Conv3d/ResNet18(Swish) front-end + 12-layer 768-d Conformer + CTC + 6-layer "
                               f"attention decoder + vq audio-token CE head (config/lrs3.yaml, 252 M parameters), random-init weights, N(0,1) clips, one "
                               f"length bucket padded to {args.frames} frames, dropout {args.dropout}",
                   "per_gpu_batch": args.lrs_batch, "valid_frames": n_frames, "enqueue": "native step list" if native else "eager (python)"},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": MFMA_PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                     "frac": round(ach * 1e12 / MFMA_PEAK_BF16, 5), "traffic": pmc_traffic(dom, lrs=True),
                     **({"frac_rocprof": round(d["flops"] / d["launches"] / (rp[0] * 1e-6) / MFMA_PEAK_BF16, 5), "rocprof_avg_launch_us": round(rp[0], 2),
                         "rocprof_source": rp[1]} if rp is not None else {"frac_rocprof": None}),
                     "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2), "launches_per_step": d["launches"],
                     "timing": "HIP events around each library call; a weight-gradient call is the contraction AND its fixed-order reduce launch "
                               "(rocprofv3 lists the two kernels separately: k_igemm_wgrad* + k_colsum / k_wgrad_unit_reduce)",
                     "per_kernel": {k: {"ms_per_step": round(v["ms"], 4), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), "launches": v["launches"]}
                                    for k, v in sorted(rows.items())}},
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--no-graph", action="store_true", help="(default) eager launches")
    ap.add_argument("--graph", action="store_true", help="capture the whole step into one HIP graph and replay it (measured slower than the "
                                                         "eager default, whose weight-gradient launches overlap on a side stream)")
    ap.add_argument("--enqueue", choices=("native", "eager"), default="native", help="native (default): the step's launch list is recorded once "
                    "and re-issued by one library call per step (csrc/steplist.hip) — the same eager launches on the same streams without the "
                    "per-launch Python cost; eager: every launch from Python (LRS and lrw-xt always run eager)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=32, help="batch of the CPU-baseline leg (default: the workload's own per-GPU batch; ~4 s per step on 32 threads)")
    ap.add_argument("--profile-steps", type=int, default=2, help="eager steps with per-launch HIP events for the roofline leg")
    ap.add_argument("--force-collective", action="store_true", help="run the RCCL path even with one rank")
    ap.add_argument("--staged-inputs", action="store_true", help="pass the batch as separate tensors: every step copies it into the recorded "
                    "step's input buffers first (default: the batch lives in TrainStep.input_buffers())")
    ap.add_argument("--bucket-mb", type=float, default=16.0, help="gradient all-reduce bucket size (MiB of fp32)")
    ap.add_argument("--grad-comm", choices=("fp32", "bf16"), default="fp32", help="wire format of the gradient buckets (fp32 = DDP's; bf16 halves the bytes)")
    ap.add_argument("--workload", choices=("lrw", "lrs", "lrw-xt"), default="lrw", help="lrw = BASELINE.json's headline metric (default); lrs = the "
                    "sentence-level E2E model (SURVEY §8 a13-a15, BASELINE configs[3]): --batch clips of up to --frames frames")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise the multi-rank control "
                    "flow on a box with one GPU, together with SVSR_BENCH_ONE_DEVICE=1)")
    ap.add_argument("--tune", default="", help="result-preserving tuning knobs for A/B runs, e.g. igemm_ksplit=0,wg_short_k=0 (syncvsr_amd.ops.tune)")
    ap.add_argument("--frames", type=int, default=150, help="LRS: padded clip length T (lengths are drawn in [0.3 T, T])")
    ap.add_argument("--lrs-steps", type=int, default=8, help="timed steps of the LRS leg attached to the default line")
    ap.add_argument("--no-lrs-leg", action="store_true", help="skip the LRS leg of the default (LRW, one GPU) run")
    ap.add_argument("--no-lrs400", action="store_true", help="skip the <= 400-frame timing inside the LRS leg")
    ap.add_argument("--sustained-steps", type=int, default=1600, help="back-to-back LRW steps of the `sustained` leg behind the headline region (0: skip); the default is ~8 s of GPU time, long enough for a 5-s utilisation sampler around the run to see it")
    ap.add_argument("--ablate", default="", help="TIMING EXPERIMENTS ONLY (gradients wrong, the line is marked invalid): comma list of conv_wgrad, lin_wgrad — "
                    "those launches are skipped (how much of the step do they cost?)")
    ap.add_argument("--dropout", type=float, default=0.1, help="LRS: dropout_rate = transformer_attn_dropout_rate (config/lrs3.yaml:20-21)")
    args = ap.parse_args()
    if args.workload == "lrs" and args.batch == 32 and "--batch" not in sys.argv:
        args.batch = 16
    args.lrs_batch = args.batch if args.workload == "lrs" else 16

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # one process per GPU, started by `python -m torch.distributed.run --nproc-per-node N` (the module docstring has the line): a bare
        # `python bench.py --gpus 4` would otherwise time ONE GPU and print a plausible-looking line
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', 'unset')} (= {world} rank(s)): launch the ranks with "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...`")
    if os.environ.get("SVSR_BENCH_ONE_DEVICE") == "1":       # test aid: every rank on GPU 0 (not a measurement)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_collective
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from syncvsr_amd import ops
    from syncvsr_amd.config import default_lrw_config
    from syncvsr_amd.engine import TrainStep
    from syncvsr_amd.init import synthetic_batch
    from syncvsr_amd.model import Model

    if args.ablate:
        import syncvsr_amd.model as _m

        _m._ABLATE = frozenset(filter(None, args.ablate.split(",")))
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        ops.tune(k, int(v))
    lrs = args.workload == "lrs"
    use_graph = args.graph and not args.no_graph     # default: eager launches + side-stream weight gradients (faster, see engine.TrainStep)
    if lrs:
        model, cfg, batch, lrs_args, n_frames, label_len = build_lrs(args, dev, world, rank)
    else:
        if args.workload == "lrw-xt":           # the encoder of the shipped yaml (bert-12l-512d_LRW_96_bf16_rrc_WB.yaml): x-transformers, 513 wide
            from syncvsr_amd.config import xtransformers_lrw_config

            cfg = xtransformers_lrw_config(True)
            if use_graph:
                raise SystemExit("layer_dropout changes the launch sequence from step to step: --graph is not available for lrw-xt")
        else:
            cfg = default_lrw_config()
        cfg.train.batch_size = args.batch
        model = Model(cfg, seed=0).to(dev).train()
        batch = [t.to(dev) for t in synthetic_batch(cfg, args.batch, seed=1234 + rank)]
    native = args.enqueue == "native" and args.workload in ("lrw", "lrs") and not use_graph
    trainer = TrainStep(model, cfg, use_graph=use_graph, always_reduce=args.force_collective, bucket_mb=args.bucket_mb,
                        grad_comm_dtype=torch.bfloat16 if args.grad_comm == "bf16" else torch.float32, native=native)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        out = trainer.step(*batch)
    batch = in_place_batch(trainer, batch, args)
    barrier()
    t0 = time.perf_counter()
    trainer.host_ms.clear()
    if trainer.dp is not None:
        trainer.dp.measure = True
    for _ in range(args.steps):
        out = trainer.step(*batch)
    host_ms = sorted(trainer.host_ms)
    barrier()
    elapsed = time.perf_counter() - t0
    host_idle = host_idle_queue_ms(trainer, batch) if not use_dist else None
    # ---- sustained leg (behind the headline region; the headline stays the driver's --steps): SURVEY section 8(d) asks for >= 200 timed steps,
    # the driver's 20 are 0.1 s of GPU time.  >= 400 back-to-back steps (>= 2 s), with the effective shader clock (svsr_clock_probe: s_memtime /
    # s_memrealtime over a block of MFMA work) read right before and right after: a figure that holds only while the chip is cold shows here.
    sustained = None
    if args.sustained_steps > 0 and not use_dist and args.workload == "lrw":
        clk0 = ops.shader_clock_mhz()
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for _ in range(args.sustained_steps):
            trainer.step(*batch)
        torch.cuda.synchronize()
        s_el = time.perf_counter() - ts0
        clk1 = ops.shader_clock_mhz()
        sustained = {"steps": args.sustained_steps, "seconds": round(s_el, 3), "ms_per_step": round(s_el / args.sustained_steps * 1e3, 4),
                     "clips_per_s": round(args.batch * args.sustained_steps / s_el, 2),
                     "shader_clock_mhz_before": round(clk0, 1), "shader_clock_mhz_after": round(clk1, 1),
                     "clock": "effective shader clock over a ~1 ms block of MFMA work on 256 workgroups (svsr_clock_probe: s_memtime / s_memrealtime), "
                              "read right before the first and right after the last of these steps"}
    if use_dist and world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    from syncvsr_amd.engine import reduce_metrics

    out = reduce_metrics(out)                        # sync_dist=True logging of the reference: rank-mean of the step's scalars
    loss = float((out[0] if lrs else out["loss_total"]).item())
    clips_per_s = args.batch * world * args.steps / elapsed

    result = {
        "metric": "lip-clips/sec training (29x88x88)",
        "value": round(clips_per_s, 2),
        "unit": "clips/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic", "inputs": "staged: copied into the step's input buffers every step" if args.staged_inputs else "resident in the step's input buffers (TrainStep.input_buffers())",
        "config": {"workload": "LRW training step (fwd+bwd+allreduce+clip+AdamW), ResNet18 + 6-layer 512-d encoder + vq audio-token CE "
                               "head, random-init weights, N(0,1) clips 29x88x88, uniform tokens/labels",
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                   "hip_graph": use_graph, "enqueue": "hip_graph" if use_graph else ("native step list" if native else "eager (python)")},
        # host time of one step's enqueue (median over the timed steps, this rank): below ms_per_step the GPU is the limit
        "host_enqueue_ms": round(host_ms[len(host_ms) // 2], 4) if host_ms else None,
        "host_enqueue_idle_queue_ms": round(host_idle, 4) if host_idle is not None else None,
        "launches_per_step": int(trainer._rec.size) if getattr(trainer, "_rec", None) is not None else None,
        "step_mfma_frac": round(clips_per_s / world * TRAIN_FLOP_PER_CLIP / MFMA_PEAK_BF16, 5),
        "final_loss": round(loss, 4),
    }
    if sustained is not None:
        result["sustained"] = sustained
    if args.ablate:
        result["INVALID_ablated"] = sorted(filter(None, args.ablate.split(",")))
    if args.workload == "lrw-xt":
        result["metric"] = "lip-clips/sec training (29x88x88, x-transformers encoder + word boundary)"
        result["config"]["workload"] = ("LRW training step with the shipped yaml's encoder: ResNet18 + 12-layer 513-d x-transformers encoder "
                                        "(RMSNorm, rotary, GEGLU, layer-drop 0.2, ff-dropout 0.3) + vq audio head; parity of that encoder is unpinned")
        result["step_mfma_frac"] = None
    if lrs:
        label_len, n_frames = label_len, n_frames
        step_flops = lrs_train_flops(args.batch, args.frames, label_len)
        result["metric"] = f"lip-clips/sec training (LRS, <= {args.frames}x88x88)"
        result["config"] = {"workload": "LRS training step (fwd+bwd+allreduce+clip+AdamW), Conv3d/ResNet18(Swish) front-end + 12-layer 768-d "
                                        "Conformer + CTC + 6-layer attention decoder + vq audio-token CE head (config/lrs3.yaml), random-init "
                                        f"weights, N(0,1) clips padded to {args.frames} frames, dropout {args.dropout}",
                            "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}", "hip_graph": use_graph,
                            "padded_frames_per_s": round(clips_per_s * args.frames, 1), "valid_frames_per_step_rank0": n_frames}
        result["step_mfma_frac"] = round(step_flops * args.steps / elapsed / MFMA_PEAK_BF16, 5)

    if use_dist and trainer.dp is not None:          # what the data-parallel path did in the last timed step
        st = model.store()
        result["collective"] = {
            "backend": (f"nccl (RCCL {'.'.join(map(str, torch.cuda.nccl.version()))})" if dist.get_backend() == "nccl" else dist.get_backend()),
            "ranks": world,
            "all_reduce_launches_per_step": len(trainer.dp.launched), "bucket_mb": args.bucket_mb, "wire_dtype": args.grad_comm,
            "gradient_mb_per_step": round(sum(hi - lo for lo, hi in trainer.dp.launched) * 4 / 2 ** 20, 1),
            "buffer_broadcast_mb_per_step": round(st.bufflat.numel() * 4 / 2 ** 20, 3), "overlapped_with_backward": True,
        }
        # what the step is exposed to: the time the main stream waits at the join after the backward (median, per rank), and every
        # rank's host enqueue time — a first multi-GPU run diagnoses itself (collective not hidden? a host-bound rank?)
        mine = torch.tensor([trainer.dp.exposed_ms() or 0.0, host_ms[len(host_ms) // 2] if host_ms else 0.0], device=dev, dtype=torch.float64)
        if world > 1:
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
        else:
            allr = [mine]
        result["collective"]["exposed_join_ms_per_rank"] = [round(float(t[0]), 4) for t in allr]
        result["collective"]["host_enqueue_ms_per_rank"] = [round(float(t[1]), 4) for t in allr]
        # every bucket of the step, in launch order (this rank): its size, when its all-reduce could START relative to the first bucket's
        # start (ready: its producers were done and the comm stream reached it) and how long the collective itself ran (ready -> done), medians
        # over the timed steps — a first 8-GPU run shows which bucket is exposed (the last one, by construction, is)
        result["collective"]["buckets_rank0"] = trainer.dp.bucket_times()

    if rank == 0:
        # ---- roofline leg: eager steps with HIP events around every contraction launch -----------------------------
        prof = TrainStep(model, cfg, use_graph=False, data_parallel=False)      # rank 0 alone: must not issue a collective
        model._side.enabled = model._side.enabled_small = False        # time every kernel alone, not overlapped with a side-stream neighbour
        if True:
            prof._step_impl(*batch)
            ops.start_event_timing()
            for _ in range(max(1, args.profile_steps)):
                prof._step_impl(*batch)
            table = ops.stop_event_timing()
            rows = {k: dict(v) for k, v in table.items() if v["flops"] > 0 and not k.endswith("+bn")}
            # launches of the same kernel whose epilogue also takes the first pass of a BatchNorm backward (label "+bn"): counted in
            # the kernel's totals (that is what rocprofv3 averages), and listed beside them
            for k, v in table.items():
                if k.endswith("+bn") and v["flops"] > 0:
                    base = rows.setdefault(k[:-3], dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
                    base["plain"] = dict(launches=base["launches"], ms=base["ms"], flops=base["flops"])
                    base["bn"] = dict(launches=v["launches"], ms=v["ms"], flops=v["flops"])
                    for f in ("launches", "ms", "flops", "bytes"):
                        base[f] += v[f]
            dom = max(rows, key=lambda k: rows[k]["ms"])
            d = rows[dom]
            achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
            rp = rocprof_avg_us(dom, lrs=lrs)
            result["roofline"] = {
                "bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": MFMA_PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                "frac": round(achieved * 1e12 / MFMA_PEAK_BF16, 5), "traffic": pmc_traffic(dom),
                # the same algorithmic FLOPs per launch over the average launch duration of the committed rocprofv3 summary (HIP events bracket the
                # library call and read ~8 % longer than the profiler's kernel time: the two figures are printed side by side, not reconciled by hand)
                **({"frac_rocprof": round(d["flops"] / d["launches"] / (rp[0] * 1e-6) / MFMA_PEAK_BF16, 5), "rocprof_avg_launch_us": round(rp[0], 2),
                    "rocprof_source": rp[1]} if rp is not None else {"frac_rocprof": None}),
                "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2), "launches_per_step": d["launches"] // max(1, args.profile_steps),
                "per_kernel": {k: {"ms_per_step": round(v["ms"] / max(1, args.profile_steps), 4),
                                   "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else None,
                                   "launches": v["launches"] // max(1, args.profile_steps),
                                   **({"plain_tflops": round(v["plain"]["flops"] / (v["plain"]["ms"] * 1e-3) / 1e12, 2) if v["plain"]["ms"] > 0 else None,
                                       "bn_epilogue_tflops": round(v["bn"]["flops"] / (v["bn"]["ms"] * 1e-3) / 1e12, 2),
                                       "bn_epilogue_launches": v["bn"]["launches"] // max(1, args.profile_steps)} if "bn" in v else {})}
                               for k, v in sorted(rows.items())},
            }
            if not args.no_cpu_baseline and world == 1:
                if lrs:
                    from syncvsr_amd.lrs_init import LRS_ODIM

                    result["cpu_baseline"] = cpu_baseline_lrs(lrs_args, LRS_ODIM, 32)
                else:       # the CPU port at the workload's own batch (the reported value) and at batch 2
                    result["cpu_baseline"] = cpu_baseline(cfg, args.cpu_batch, budget_s=14.0, min_timed=5)
                    phys = physical_cores()
                    if phys is not None and phys > result["cpu_baseline"]["cores"]:       # the same step on ALL physical cores
                        allc = cpu_baseline(cfg, args.cpu_batch, budget_s=8.0, min_timed=3, threads=phys)
                        result["cpu_baseline"]["all_cores"] = {"value": round(allc["value"], 3), "cores": allc["cores"], "sample": allc["sample"]}
                    if args.cpu_batch != 2:
                        small = cpu_baseline(cfg, 2, budget_s=4.0, min_timed=5)
                        result["cpu_baseline"]["batch2_value"] = round(small["value"], 3)
                        result["cpu_baseline"]["sample"] += "; batch 2: " + small["sample"]
        if args.workload == "lrw" and world == 1 and not args.no_lrs_leg and not use_dist:
            # BASELINE configs[3]: a bounded LRS measurement rides on the default line (the headline metric / value above stay LRW's)
            del prof, trainer, model
            torch.cuda.empty_cache()
            try:
                result["lrs"] = lrs_leg(args, dev, with_cpu=not args.no_cpu_baseline)
            except Exception as e:          # the headline line must survive a failure of the extra leg
                result["lrs"] = {"error": f"{type(e).__name__}: {e}"}
        try:        # RCCL prints its version banner through C stdio, which is flushed at exit: push it out first so the JSON line is last
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
