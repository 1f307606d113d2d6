"""Training-step runtime around the HIP model: what Lightning's Trainer does for the reference (DDP gradient all-reduce,
global-norm clip, AdamW, cosine schedule — LRW/video/src/train.py:23-40, lightning.py:216-223) restated
for one process per GPU:

  * gradients live in ONE flat fp32 buffer laid out in forward order; the hand-written backward (model.py) finalises it from
    its end to its start and reports progress, so contiguous buckets are all-reduced (RCCL, average) on a side HIP stream
    while the rest of the backward still runs;
  * clip + AdamW + schedule + bf16 shadow refresh are two kernels over the flat buffers, with the step counter, the gradient
    norm and the learning rate kept on the device — nothing in a step depends on host values, so
  * the whole step (forward, backward, collectives, optimiser) CAN be captured once into a HIP graph and replayed
    (`use_graph=True`); the default is eager launches with the weight-gradient kernels on a side stream, which measures faster
    (host enqueue time is about half the GPU time of a step).
"""
from __future__ import annotations

from typing import Optional

import warnings

import torch
import torch.distributed as dist

from . import ops
from .config import Config
from .model import TransformerLightningModule


import os as _os

ZERO_GRADS_EARLY = _os.environ.get("SVSR_ZERO_GRADS_EARLY", "1") != "0"       # gradient buffer zeroed at the start of the step, on the side stream (TrainStep._zero_grads_early)
SPLIT_OPTIMIZER = _os.environ.get("SVSR_SPLIT_OPTIMIZER", "1") != "0"      # AdamW of everything behind the front-end on the side stream, beside the next forward

_ROCTX = _os.environ.get("SVSR_ROCTX", "0") == "1"


def lrs_train_config(**kw) -> Config:
    """Optimiser / schedule / clip values of LRS/video/config/lrs3.yaml:66-77,97 in the layout TrainStep reads."""
    cfg = Config(optimizer=Config(lr=1e-3, betas=[0.9, 0.98], eps=1e-6, weight_decay=0.03),
                 scheduler=Config(name="cosine", num_warmup_steps=25000, num_training_steps=500000),
                 trainer=Config(gradient_clip_val=5.0))
    for k, v in kw.items():
        cfg.set_path(k.replace("__", "."), v)
    return cfg


def _rng_state_to_tensor(state) -> torch.Tensor:
    """random.Random.getstate() = (version, 625 words, gauss_next or None) as an int64 vector
    [version, 625 words, has_gauss, gauss bits]: plain numbers, nothing a checkpoint reader has to unpickle."""
    import struct

    version, words, gauss = state
    if len(words) != 625:
        raise ValueError("unexpected random.Random state layout")
    bits = 0 if gauss is None else struct.unpack("<q", struct.pack("<d", float(gauss)))[0]
    return torch.tensor([int(version), *[int(w) for w in words], 0 if gauss is None else 1, bits], dtype=torch.int64)


def _rng_state_from_tensor(t: torch.Tensor):
    import struct

    if t.dtype != torch.int64 or t.numel() != 628:
        raise ValueError("layer_rng in this checkpoint is not the int64[628] generator state TrainStep.state_dict writes "
                         "(pickled states of older checkpoints are refused)")
    v = [int(x) for x in t.cpu().tolist()]
    gauss = struct.unpack("<d", struct.pack("<q", v[627]))[0] if v[626] else None
    return (v[0], tuple(v[1:626]), gauss)


class TrainStep:
    """forward + backward + (all-reduce) + clip + AdamW for the LRW model (`TransformerLightningModule`; its step takes
    (videos, audio_tokens, labels, word_mask)) or the LRS model (`lrs_model.E2E`; (x, lengths, audio_tokens, label));
    optionally one HIP graph per step.

    After step() returns, the tail of the optimiser step (AdamW of everything behind the front-end, its shadow transposes) may still be
    running on the model's side stream, beside whatever the main stream does next.  The model's own entry points join it where they need
    the parameters (forward before the encoder, forward_videos, state_dict, load_state_dict, the LRS scorers, refresh_shadows); code that
    touches parameters DIRECTLY (p.data, p.cpu(), an EMA update) calls synchronize() first."""

    def __init__(self, model, config: Optional[Config] = None, process_group=None,
                 use_graph: bool = False, bucket_mb: float = 32.0, always_reduce: bool = False, data_parallel: bool = True,
                 grad_comm_dtype: torch.dtype = torch.float32, native: bool = False):
        """native=True: the launch sequence of the first step is recorded into a native step list (csrc/steplist.hip) and every
        later step re-issues it with one library call per segment — eager launches on the same streams (the weight-gradient side
        stream keeps overlapping, which a captured HIP graph loses) without the per-launch host cost of the Python loop.  Batch
        shapes are fixed by the first call, as with use_graph."""
        self.model = model
        self.is_lrw = isinstance(model, TransformerLightningModule)
        if self.is_lrw:            # LRW/video/config/*.yaml: optim.optimizer / optim.scheduler / train.gradient_clip_val
            cfg = config or model.config
            opt = cfg.optim.optimizer
            sch = cfg.optim.get("scheduler", {}) or {}
            clip = cfg.train.get("gradient_clip_val", 0.0)
        else:                      # LRS/video/config/lrs3.yaml: optimizer / scheduler / trainer.gradient_clip_val
            cfg = config or lrs_train_config()
            opt = cfg.optimizer
            sch = cfg.get("scheduler", {}) or {}
            clip = (cfg.get("trainer", {}) or {}).get("gradient_clip_val", 0.0)
        accum = (cfg.get("train", {}) or {}).get("accumulate_grad_batches", 1) if self.is_lrw else (cfg.get("trainer", {}) or {}).get("accumulate_grad_batches", 1)
        if int(accum or 1) != 1:
            raise NotImplementedError("accumulate_grad_batches != 1 is not supported by TrainStep (one optimiser step per batch)")
        self.lr = float(opt.lr)
        self.betas = (float(opt.betas[0]), float(opt.betas[1]))
        self.eps = float(opt.eps)
        self.weight_decay = float(opt.weight_decay)
        self.max_norm = float(clip or 0.0)
        self.warmup = int(sch.get("num_warmup_steps", 0) or 0)
        self.total_steps = int(sch.get("num_training_steps", 0) or 0)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # always_reduce: run the collective path even for a 1-rank group (exercises RCCL + graph capture on one GPU)
        # data_parallel=False: a purely local step even inside an initialised process group (no collective is ever issued — e.g. the
        # single-rank profiling leg of bench.py, which the other ranks do not take part in)
        self.dp = GradReducer(model, process_group, bucket_mb, always_reduce, grad_comm_dtype) if data_parallel and (self.world > 1 or always_reduce) else None
        if not data_parallel:
            model.grad_ready_hook = None
        self.use_graph = use_graph
        self.native = bool(native)
        if self.native and use_graph:
            raise ValueError("native=True and use_graph=True are two ways of replaying a step: pick one")
        if self.native and not self.is_lrw and getattr(model, "length_norm", False):
            raise NotImplementedError("native=True: transformer_length_normalized_loss needs a torch kernel inside the step (use native=False)")
        if self.native and getattr(model, "layer_drop_p", 0.0) > 0.0:
            raise NotImplementedError("layer_dropout changes the launch sequence from step to step: it cannot be replayed from a recorded list")
        self._rec: Optional[ops.StepRecorder] = None
        self.fused_encoder_fallbacks = 0        # times _watch_fused_encoder switched the encoder to the launch chain
        self.host_ms: list[float] = []          # host time of the last steps' enqueue (bench.py reports the median)
        if use_graph and getattr(model, "layer_drop_p", 0.0) > 0.0:
            raise NotImplementedError("layer_dropout skips whole encoder blocks at random: the launch sequence differs from step to "
                                      "step and cannot be replayed from one captured HIP graph (use use_graph=False)")
        # Weight-gradient launches only feed the flat gradient buffer, so in EAGER mode they run on a side stream next to the
        # data-gradient chain and fill its tails: LRW 8.06 -> 7.39 ms (trunk convs), LRS 31.9 -> 30.4 ms (trunk convs + the
        # 20-40 us linear GEMMs that fill about half the chip each).  Under HIP-graph replay the forked branches cost more than
        # they hide (LRW 7.85 -> 8.5 ms, LRS 33.2 ms), so a captured step keeps everything in line.
        import os
        graph_side = os.environ.get("SVSR_GRAPH_SIDE", "0") == "1"
        model._side.enabled = (not use_graph or graph_side) and os.environ.get("SVSR_SIDE_TRUNK", "1") != "0"
        if not self.is_lrw:
            model._side.enabled_small = (not use_graph or graph_side) and os.environ.get("SVSR_SIDE_ENCODER", "1") != "0"
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._static: Optional[list[torch.Tensor]] = None
        self._out: Optional[dict[str, torch.Tensor]] = None
        st = model.store()
        dev = st.flat.device
        self.m = torch.zeros_like(st.flat)
        self.v = torch.zeros_like(st.flat)
        # device state of the optimiser kernels: {step, sumsq, lr_last, gnorm_last, 1024 partial sums of squares}
        self.opt_state = torch.zeros(4 + 1024, dtype=torch.int32, device=dev)

    # -- one eager step -----------------------------------------------------------------------------
    def _step_impl(self, *batch):
        model = self.model
        st = model.store()
        trace = _ROCTX                                   # SVSR_ROCTX=1: roctx ranges (rocprofv3 --marker-trace) around the phases
        if self.dp is not None:
            self.dp.begin_step()
        if trace:
            torch.cuda.nvtx.range_push("svsr.forward")
        self._zero_grads_early(st)
        out = model(*batch)
        if trace:
            torch.cuda.nvtx.range_pop()
            torch.cuda.nvtx.range_push("svsr.backward")
        (out["loss_total"] if self.is_lrw else out[0]).backward()
        if trace:
            torch.cuda.nvtx.range_pop()
            torch.cuda.nvtx.range_push("svsr.allreduce_join+optimizer")
        if self.dp is not None:
            self.dp.finish()
        self._optimizer(st)
        if self.use_graph:
            model._side.join()       # (SVSR_GRAPH_SIDE=1: work forked onto the side stream must be joined before the capture ends)
        if trace:
            torch.cuda.nvtx.range_pop()
        if self.is_lrw:
            return {k: v.detach() for k, v in out.items()}
        return tuple(v.detach() for v in out)

    def _zero_grads_early(self, st) -> None:
        """The flat gradient buffer is zeroed when the step BEGINS, on the side stream (behind the previous step's optimiser, which is the
        last reader; ahead of this step's weight gradients, the first writers there; the model joins the side stream before its encoder
        runs, long before the backward's main-stream writers): the 128 MB (LRW) / 1 GB (LRS) fill leaves the main stream, where it sat in front
        of the backward (17 / 140 us).  The backward's own zero_grad() then finds `grad_clean` set and does nothing."""
        model = self.model
        # (a step that aborted between its backward and its optimiser — an exception in the training loop — must not leave its handshake
        # behind: the flag is only valid for the step that set it)
        st.__dict__.pop("sumsq_tail_done", None)
        # no collective between the backward and the clip: the model may sum the squares of every gradient but the last while that one is
        # computed (set per step: two TrainSteps may drive one model, each with its own optimiser state)
        model._early_sumsq = self.opt_state if (self.dp is None and ops.EARLY_SUMSQ) else None
        if getattr(model, "accumulate_grads", False) or not ZERO_GRADS_EARLY:
            return
        model._side.run(lambda: ops.memset(st.grad, 0))
        model._side.flush()
        st.grad_clean = True

    def _optimizer(self, st) -> None:
        """Global-norm clip + AdamW + bf16 shadows.  With the model's side stream in use the update is SPLIT: the visual front-end's weights
        (and every 1-D tensor: BatchNorm / LayerNorm parameters, biases), which the next forward needs first, on the main stream; everything
        behind the front-end (two thirds of the word-level model, 95 % of the sentence-level one) on the side stream, where this
        HBM-bound pass runs beside the next step's stem / trunk forward.  The model joins the side stream before its encoder runs and
        before state_dict(); the step counter advances behind the last range."""
        model = self.model
        model._early_sumsq = None        # (a backward outside a step must not write this optimiser's state)
        if st.sumsq_head:
            # two ranges, always the same two (one association whatever ran early): [head, n) -> partial sums 0..1022 — already summed on the
            # side stream beside the stem's weight gradient when the model could (model._early_sumsq) — and the stem weight -> partial 1023
            if not st.__dict__.pop("sumsq_tail_done", False):
                ops.grad_sumsq_parts(st.grad, st.sumsq_head, st.numel - st.sumsq_head, self.opt_state, 0, ops.SUMSQ_PARTS - 1)
            ops.grad_sumsq_parts(st.grad, 0, st.sumsq_head, self.opt_state, ops.SUMSQ_PARTS - 1, 1)
        else:
            ops.grad_sumsq(st.grad, self.opt_state)
        side = model._side
        hp = (self.lr, self.betas, self.eps, self.weight_decay, self.max_norm, self.warmup, self.total_steps, self.opt_state)
        if not (SPLIT_OPTIMIZER and side.enabled and st.front_end < st.decay_end):
            ops.adamw_step(st.flat, st.grad, self.m, self.v, st.w16, st.decay_end, *hp)
            ops.transpose_shadows(st.flat, st.w16, st.w16t, st.table, st.n_entries)
        else:
            bufs = (st.flat, st.grad, self.m, self.v, st.w16)
            ops.adamw_range(*bufs, 0, st.front_end, st.decay_end, *hp, advance=False)
            ops.adamw_range(*bufs, st.decay_end, st.numel, st.decay_end, *hp, advance=False)
            nf = st.n_entries_front
            side.run(lambda: (ops.adamw_range(*bufs, st.front_end, st.decay_end, st.decay_end, *hp, advance=True),
                              ops.transpose_shadows_range(st.w16, st.w16t, st.table, nf, st.n_entries - nf)))
            side.flush()
            ops.transpose_shadows_range(st.w16, st.w16t, st.table, 0, nf)
        st.shadow_fresh = True
        st.generation += 1

    def synchronize(self) -> None:
        """Joins whatever the last step left on the model's side stream (the tail of its optimiser step)."""
        self.model._side.join()

    # -- native step list ----------------------------------------------------------------------------
    def _direct_impl(self, *batch):
        """_step_impl without autograd or torch kernels: every device operation is a library call (recordable)."""
        model = self.model
        st = model.store()
        self._zero_grads_early(st)
        out = model.train_step_direct(*batch)
        if self.dp is not None:
            ops.host_callback(self.dp.finish)
        self._optimizer(st)
        return out

    def _native_step(self, *batch):
        model = self.model
        if self._rec is None:
            if not model.training:
                raise RuntimeError("TrainStep(native=True) records a TRAINING step: call model.train() first")
            prepped = model.prepare_batch(*batch)
            if self._static is None:
                self._static = [t.clone() if torch.is_tensor(t) else t for t in prepped]
            else:                        # recorded again (_watch_fused_encoder): the input buffers a loader may be writing into stay the same
                for dst, src in zip(self._static, prepped):
                    if torch.is_tensor(dst) and dst.data_ptr() != src.data_ptr():
                        dst.copy_(src)
            st = model.store()
            if not st.shadow_fresh:
                st.refresh_shadows()
                st.shadow_fresh = True
            if self.dp is not None:
                self.dp.begin_step()
            if model._side.stream is None:
                model._side.stream = torch.cuda.Stream()
            model.direct_constants(self._static[0].device)
            if getattr(model, "_drop_word", None) is None and (model.drop_p > 0.0 or model.attn_drop_p > 0.0 or getattr(model, "emb_drop_p", 0.0) > 0.0):
                model._advance_dropout(self._static[0].device)      # creates the seed word outside the recorded region ...
                ops.word_add(model._drop_word, -1)                   # ... and leaves its value where the first forward expects it
            rec = ops.StepRecorder()
            with ops.recording(rec):
                out = self._direct_impl(*self._static)
            self._rec = rec
            self._out = {k: v.detach() for k, v in out.items()} if isinstance(out, dict) else tuple(v.detach() for v in out)
            self._main_stream = rec.main_stream
            return self._out
        if not self.is_lrw:          # LRS: the conversions (and the decoder / CTC targets) are redone per batch, then copied into the static inputs
            batch = model.prepare_batch(*batch)
        for dst, src in zip(self._static, batch):
            if not torch.is_tensor(dst):
                continue
            if dst.shape[0] != src.shape[0] or dst.shape[2:] != src.shape[2:] or (not self.is_lrw and dst.shape != src.shape):
                raise ValueError("a recorded TrainStep needs fixed batch shapes (pad to the recorded size or use native=False)")
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src if src.dim() != 2 or src.shape[1] == dst.shape[1] else src[:, : dst.shape[1]], non_blocking=True)
        if ops._stream() != self._main_stream:
            raise RuntimeError("a recorded TrainStep must be replayed on the stream it was recorded on")
        if self.dp is not None:
            self.dp.begin_step()
        self._rec.run()
        model._store.generation += 1
        return self._out

    def input_buffers(self) -> Optional[tuple]:
        """The device tensors a recorded (native / captured) step reads its batch from, in the order of step()'s arguments — None
        before the first step, for an eager TrainStep, and in the slots whose content step() derives per batch (LRS: lengths, labels).
        A loader that writes the next batch INTO these (its host-to-device copy lands where the step reads) and passes them back to
        step() saves the device-to-device copy of the clip (LRW B = 32: 29 MB, 111 us; LRS 16 x 160 frames: 79 MB, ~300 us):
        step() skips every argument that already is its static buffer."""
        if self._static is None:
            return None
        if self.is_lrw or self.use_graph:
            return tuple(self._static)
        return (self._static[0], None, self._static[2], None)

    def step(self, *batch):
        """One optimisation step; returns the model's outputs (LRW: the dict of five scalars; LRS: the 5-tuple).
        With use_graph / native the batch shapes are fixed by the first call (later batches are copied into the static buffers)."""
        self._watch_fused_encoder()
        try:
            return self._step(*batch)
        except BaseException:
            # an aborted step must not leave its early-sum-of-squares handshake pointing at this optimiser's state (a later stand-alone
            # backward would write partial sums into it and the next optimiser step would trust them)
            self.model._early_sumsq = None
            self.model.store().__dict__.pop("sumsq_tail_done", None)
            raise

    def _watch_fused_encoder(self) -> None:
        """Automatic fall-back of the fused encoder (csrc/enc_fused.hip).  Its launches need the 8 workgroups of a sequence resident together;
        when a bounded cluster wait gives up (a co-tenant kernel holding LDS or compute units — e.g. a peer-waiting collective kernel), the
        launch poisons its output with NaN and sets a flag that is also stored into pinned host memory.  The poisoned step updates nothing
        (svsr_adamw_step skips a step whose gradient norm is not finite); here, before the next step is enqueued, the flag is read WITHOUT a
        synchronisation and the encoder is re-routed to the per-layer launch chain for the rest of the run: the recorded list / captured
        graph is dropped and re-made on that path.  At most the steps already enqueued when the wait gave up are lost (skipped)."""
        if not self.is_lrw or not ops.enc_gave_up_peek():
            return
        torch.cuda.synchronize()
        ops.check_enc_clusters(reset=True)
        ops.disable_enc_fused("a cluster wait of svsr_enc_fwd / svsr_enc_bwd gave up (its workgroups were not resident together)")
        self.fused_encoder_fallbacks += 1
        self._rec = None                 # native: record the step again (now on the chain)
        self._graph = None               # graph: capture again

    def _step(self, *batch):
        if self.native:
            import time as _time

            t0 = _time.perf_counter()
            out = self._native_step(*batch)
            self.host_ms.append((_time.perf_counter() - t0) * 1e3)
            if len(self.host_ms) > 256:
                del self.host_ms[:128]
            return out
        if not self.use_graph:
            import time as _time

            t0 = _time.perf_counter()
            out = self._step_impl(*batch)
            self.host_ms.append((_time.perf_counter() - t0) * 1e3)
            if len(self.host_ms) > 256:
                del self.host_ms[:128]
            return out
        if self._graph is None:
            self._capture(*batch)
        else:
            for dst, src in zip(self._static, batch):
                if dst.shape != src.shape:
                    raise ValueError("a captured TrainStep needs fixed batch shapes (pad to the captured size or use use_graph=False)")
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
        self._graph.replay()
        self.model._store.generation += 1
        return self._out

    def _capture(self, *batch) -> None:
        ops.PLAN_CACHE_PINNED = True       # the captured launches reference the plans' device words
        if self._static is None:
            self._static = [t.clone() for t in batch]
        else:                            # captured again (_watch_fused_encoder): same input buffers
            for dst, src in zip(self._static, batch):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
        # warm-up on a side stream (allocator pools, hipFuncSetAttribute, lazy module loads), state restored afterwards
        st = self.model.store()
        snap = (st.flat.clone(), self.m.clone(), self.v.clone(), self.opt_state.clone(),
                {k: b.clone() for k, b in st.buffers.items()})
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._step_impl(*self._static)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        st.flat.copy_(snap[0]); self.m.copy_(snap[1]); self.v.copy_(snap[2]); self.opt_state.copy_(snap[3])
        for k, b in st.buffers.items():
            b.copy_(snap[4][k])
        st.refresh_shadows()          # shadows follow the restored weights; inside the graph the optimiser keeps them fresh
        st.shadow_fresh = True
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: with a process group alive, the collective backend's watchdog thread polls HIP events at any
        # moment, which the default (global) mode turns into "operation not permitted when stream is capturing" and an abort
        with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
            self._out = self._step_impl(*self._static)

    # -- checkpoint / resume -------------------------------------------------------------------------
    def state_dict(self) -> dict[str, torch.Tensor]:
        """Optimiser state for checkpointing (what Lightning stores next to the model's state_dict): AdamW moments as flat
        fp32 vectors in the parameter store's order, and the 16-byte device state {step, -, lr, grad-norm}."""
        self.synchronize()
        sd = {"exp_avg": self.m.detach().clone(), "exp_avg_sq": self.v.detach().clone(), "opt_state": self.opt_state.detach().clone(),
              # what the order of the additions inside a weight gradient / a BatchNorm statistic depends on (ops.REDUCTION_KNOBS; the
              # compute-unit count the splits are planned for is one of them and is a fixed 256, not the device's): a resumed run is
              # bit-identical when these agree
              "reduction_plan": torch.tensor(ops.reduction_plan_params(), dtype=torch.int64)}
        if hasattr(self.model, "rng_state"):        # dropout seed word + layer-drop generator: a resumed run draws the same masks / skips
            rs = self.model.rng_state()
            sd["dropout_word"] = torch.tensor([rs["dropout_word"]], dtype=torch.int64)
            if "layer_rng" in rs:
                sd["layer_rng"] = _rng_state_to_tensor(rs["layer_rng"])
        return sd

    def load_state_dict(self, sd: dict[str, torch.Tensor]) -> None:
        self.synchronize()
        if sd["exp_avg"].numel() != self.m.numel():
            raise ValueError("optimiser state belongs to a different parameter layout")
        self.m.copy_(sd["exp_avg"])
        self.v.copy_(sd["exp_avg_sq"])
        self.opt_state[:4].copy_(sd["opt_state"][:4])
        if "reduction_plan" in sd:
            then, now = [int(x) for x in sd["reduction_plan"].tolist()], ops.reduction_plan_params()
            if then != now:
                diff = {k: (a, b) for k, a, b in zip(ops.REDUCTION_KNOBS, then, now) if a != b}
                warnings.warn(f"this checkpoint was written with other reduction-split knobs {diff} (then, now): the resumed run is numerically "
                              f"equivalent but not bit-identical to the original (ops.tune(key, value) restores them)")
        self.opt_state[1] = 0                    # (word 1 counts the skipped steps of THIS run; older checkpoints kept a float there)
        if "dropout_word" in sd and hasattr(self.model, "load_rng_state"):
            rs = {"dropout_word": int(sd["dropout_word"].reshape(-1)[0])}
            if "layer_rng" in sd:
                try:
                    rs["layer_rng"] = _rng_state_from_tensor(sd["layer_rng"])
                except ValueError as e:       # checkpoints written before the int64[628] format (pickled uint8 payload): nothing here unpickles
                    warnings.warn(f"{e}; the layer-drop generator is NOT restored (it continues from this process's seed), everything else is")
            self.model.load_rng_state(rs)

    # -- introspection ------------------------------------------------------------------------------
    def state(self) -> dict[str, float]:
        """{step, lr, grad_norm, skipped_steps} of the last optimiser step (a host synchronisation).  skipped_steps counts steps whose gradient
        norm was not finite: they updated nothing and did not advance `step` (svsr_adamw_step).  A fused-encoder launch whose cluster wait
        gave up (csrc/enc_fused.hip) produces such a step; it is picked up here as well as before every step (_watch_fused_encoder): the
        encoder continues on the per-layer launch chain, with a warning."""
        self.synchronize()
        raw = self.opt_state.cpu()
        if self.is_lrw and ops.check_enc_clusters(reset=False):
            self._watch_fused_encoder()
        if not self.is_lrw and hasattr(self.model, "check_targets"):
            self.model.check_targets()          # a label outside [1, odim) reached svsr_lrs_targets: raises with the cause
        f = raw.view(torch.float32)
        return {"step": int(raw[0]), "lr": float(f[2]), "grad_norm": float(f[3]), "skipped_steps": int(raw[1])}


def reduce_metrics(metrics, process_group=None):
    """`self.log(..., sync_dist=True)` of the reference (LRW/video/src/lightning.py:208,214; LRS/video/lightning.py:143-216): the
    logged scalars are averaged over the ranks.  `metrics` is the dict (LRW) or tuple (LRS) of 0-d tensors a step returns; they are
    packed into one vector so a step's logging costs ONE collective.  With no process group (or one rank) the input is returned."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return metrics
    keys = list(metrics.keys()) if isinstance(metrics, dict) else list(range(len(metrics)))
    vec = torch.stack([metrics[k].detach().float().reshape(()) for k in keys])
    if dist.get_backend(process_group) == "nccl":
        dist.all_reduce(vec, op=dist.ReduceOp.AVG, group=process_group)
    else:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=process_group)
        vec = vec / dist.get_world_size(process_group)
    out = {k: vec[i] for i, k in enumerate(keys)}
    return out if isinstance(metrics, dict) else tuple(out[k] for k in keys)


class GradReducer:
    """Bucketed gradient all-reduce over the flat gradient buffer, overlapped with the backward pass.

    The flat buffer is [decayed tensors in forward order | 1-D tensors].  The backward calls `on_ready(lo)` meaning
    "decayed offsets >= lo are final"; whole buckets are peeled off the top of the decayed region as soon as they are
    complete and reduced on a side stream.  `on_ready(0)` (end of backward) flushes the rest plus the 1-D tail.
    """

    def __init__(self, model, process_group=None, bucket_mb: float = 32.0, always: bool = False, comm_dtype: torch.dtype = torch.float32):
        self.model = model
        # comm_dtype=torch.bfloat16: buckets cross the links as bf16 (half the bytes) and come back into the fp32 gradient
        # buffer — an opt-in deviation from DDP's fp32 all-reduce, for when the step is short enough for the collective to show
        if comm_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("comm_dtype must be torch.float32 or torch.bfloat16")
        self.comm_dtype = comm_dtype
        self._comm_buf: Optional[torch.Tensor] = None
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.always = always and dist.is_initialized()
        self.bucket_elems = max(1, int(bucket_mb * (1 << 20) / 4))
        # the bucket at the bottom of the buffer (stem, layer1: final only when the backward ends) is the one collective nothing can
        # hide: bucket edges are anchored so that it holds at most 4 MB, whatever is left over goes to the FIRST bucket (top of the
        # buffer, ready earliest)
        self.last_elems = min(self.bucket_elems, 1 << 20)
        self._st = None
        self.measure = False                  # bench.py: time the join in finish() with HIP events (the collective time the step is exposed to)
        self._join_events: list = []
        self._bucket_events: list = []        # measure=True: per step a list of (elements, ready event, done event), one per bucket in launch order
        self._backend = dist.get_backend(process_group) if dist.is_initialized() else None
        self.comm_stream: Optional[torch.cuda.Stream] = None
        self.top = 0
        self.launched: list[tuple[int, int]] = []
        model.grad_ready_hook = self.on_ready
        # DDP construction semantics (torch DistributedDataParallel behind Lightning's strategy="ddp", reference
        # LRW/video/src/train.py:28): every rank starts from rank 0's parameters AND buffers, whatever its own seed or a
        # per-rank load_state_dict left behind.
        if dist.is_initialized() and (self.world > 1 or self.always):
            st = model.store()
            if st.flat.is_cuda and dist.get_backend(process_group) == "nccl":
                pg = process_group if process_group is not None else dist.distributed_c10d._get_default_group()
                if getattr(pg, "bound_device_id", None) is None:
                    # measured on one MI355X with a 1-rank group: init_process_group("nccl", ..., device_id=dev) 6.18 ms per LRW step,
                    # the lazily initialised communicator (no device_id) 14.3 ms — with or without a collective inside the step
                    warnings.warn("the RCCL process group is not bound to a device: pass device_id=torch.device('cuda', local_rank) to "
                                  "init_process_group (a lazily initialised communicator measured 2.3x slower training steps)")
            dist.broadcast(st.flat, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0, group=process_group)
            self._broadcast_buffers(st)
            for b in st.buffers.values():
                if not b.is_floating_point():
                    dist.broadcast(b, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0, group=process_group)
            st.shadow_fresh = False          # the bf16 shadows follow the broadcast weights at the next forward

    def exposed_ms(self) -> Optional[float]:
        """Median time the main stream sat at the join of finish() (measure=True): what the step pays for collectives the backward
        did not hide.  Synchronises the device."""
        if not self._join_events:
            return None
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in self._join_events)
        self._join_events.clear()
        return t[len(t) // 2]

    def bucket_times(self) -> list:
        """measure=True: per bucket of a step (launch order) {mb, ready_ms: start of its all-reduce after the first bucket's start, ms: ready -> done
        on the comm stream}, medians over the measured steps.  Synchronises the device."""
        steps = [s for s in self._bucket_events if s]
        if not steps:
            return []
        torch.cuda.synchronize()
        n = min(len(s) for s in steps)
        out = []
        for i in range(n):
            ready = sorted(s[0][1].elapsed_time(s[i][1]) for s in steps)
            dur = sorted(s[i][1].elapsed_time(s[i][2]) for s in steps)
            out.append({"mb": round(steps[0][i][0] * 4 / 2 ** 20, 2), "ready_ms": round(ready[len(ready) // 2], 4), "ms": round(dur[len(dur) // 2], 4)})
        self._bucket_events.clear()
        return out

    def _broadcast_buffers(self, st) -> None:
        """broadcast_buffers=True of DDP: the BatchNorm running statistics of every rank follow rank 0's — one collective over
        the flat buffer vector (model._ParamStore.bufflat).  num_batches_tracked advances identically on every rank."""
        if st.bufflat.numel():
            src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            dist.broadcast(st.bufflat, src=src, group=self.group)

    def begin_step(self) -> None:
        st = self._st = self.model.store()         # (store() re-validates ~300 tensors: once per step, not once per hook call)
        self.top = st.decay_end
        self.launched = []
        if self.measure:
            self._bucket_events.append([])
            if len(self._bucket_events) > 64:
                del self._bucket_events[:32]
        if self.comm_stream is None and st.flat.is_cuda:
            self.comm_stream = torch.cuda.Stream(device=st.flat.device)

    def _reduce(self, lo: int, hi: int, fence: bool = True) -> None:
        """fence=False: the comm stream already waits for this segment's producers (an earlier bucket of the same on_ready call)."""
        if hi <= lo:
            return
        self.launched.append((lo, hi))
        st = self._st if self._st is not None else self.model.store()
        seg = st.grad[lo:hi]
        if self.world == 1 and not self.always:
            return
        backend = self._backend
        if seg.is_cuda:
            if fence:
                self.comm_stream.wait_stream(torch.cuda.current_stream())     # the segment's producers are enqueued there ...
                side = getattr(self.model, "_side", None)
                if side is not None and side.stream is not None and (side.enabled or side.enabled_small):
                    self.comm_stream.wait_stream(side.stream)                 # ... and, for weight gradients, on the model's side stream
            with torch.cuda.stream(self.comm_stream):
                if self.measure and self._bucket_events:
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()              # on the comm stream, behind its waits: the bucket's producers are done
                if self.comm_dtype == torch.bfloat16:
                    if self._comm_buf is None or self._comm_buf.numel() < seg.numel():      # on the comm stream, used only there
                        self._comm_buf = torch.empty(max(seg.numel(), self.bucket_elems), dtype=torch.bfloat16, device=seg.device)
                    wire = self._comm_buf[: seg.numel()]
                    wire.copy_(seg)
                else:
                    wire = seg
                if backend == "nccl":
                    dist.all_reduce(wire, op=dist.ReduceOp.AVG, group=self.group)
                else:
                    dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group)
                    wire.div_(self.world)
                if wire is not seg:
                    seg.copy_(wire)
                if self.measure and self._bucket_events:
                    ev1.record()
                    self._bucket_events[-1].append((hi - lo, ev0, ev1))
        else:
            dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group)
            seg.div_(self.world)

    def on_ready(self, lo: int) -> None:
        st = self._st if self._st is not None else self.model.store()
        # one fence per call: every bucket launched here waits for the same producers (all enqueued by now), the later ones follow
        # the first in comm-stream order
        if lo == 0:
            self._reduce(0, self.top)
            fenced = self.top > 0
            self.top = 0
            self._reduce(st.decay_end, st.numel, fence=not fenced)
            return
        fence = True
        while self.top > self.last_elems:
            edge = self.last_elems + (self.top - 1 - self.last_elems) // self.bucket_elems * self.bucket_elems     # lower edge of the top bucket
            if edge < lo:
                break
            self._reduce(edge, self.top, fence)
            self.top, fence = edge, False

    def finish(self) -> None:
        """Joins the bucket all-reduces.  DDP re-broadcasts the buffers before every forward; here rank 0's BatchNorm running
        statistics (a few KB in one flat vector) follow the buckets on the comm stream and the ONE join below covers both, so
        nothing is left un-joined when the step ends: a captured HIP graph has no dangling branch, and an eval forward or a
        state_dict() read right after the step sees the broadcast values."""
        if (self.world > 1 or self.always) and self.comm_stream is not None:
            st = self._st if self._st is not None else self.model.store()
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self._broadcast_buffers(st)
            if self.measure:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            torch.cuda.current_stream().wait_stream(self.comm_stream)
            if self.measure:
                e1.record()
                self._join_events.append((e0, e1))
        elif self.world > 1 or self.always:
            self._broadcast_buffers(self._st if self._st is not None else self.model.store())
