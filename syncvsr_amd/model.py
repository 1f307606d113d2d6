"""Drop-in module for the reference's ``TransformerLightningModule`` (LRW/video/src/lightning.py:36-223), HIP-native.

Same constructor (``Model(config)``), same ``forward(videos, audio_tokens, labels, word_mask) -> dict`` with the five
scalar keys, same ``forward_videos`` / ``training_step`` / ``configure_optimizers`` helpers and the same state-dict key
names (``stem3d.0.weight`` … ``encoder.encoder.layer.N.attention.self.query.weight`` … ``audio_projection.weight``).
Everything between the inputs and the two losses runs in libsyncvsr_hip.so; this file is orchestration only:
it owns the parameters (one flat fp32 buffer + flat fp32 gradient buffer + bf16 shadows for the MFMA kernels),
records the forward "tape" and replays it backwards by hand — a single autograd node stands for the whole model,
so ``loss_total.backward()`` works as in the reference loops while no per-op autograd graph is built.
"""
from __future__ import annotations

import math
import os
from typing import Any, Optional

import torch
import torch.nn as nn

from . import ops
from .config import Config, audio_codec_dims
from .init import buffer_specs, hidden_dim, init_state_dict, pad64, param_specs, phys_shape, resnet_block_specs, xt_dims

BF16 = torch.bfloat16


class _SideStream:
    """Weight-gradient launches go to a second HIP stream: they only feed the flat gradient buffer, so they can fill the
    tails of the data-gradient / BatchNorm kernels of the main stream (every kernel here ends with a partially filled
    last round of workgroups).  Tensors handed to the side stream are kept alive until join()."""

    def __init__(self) -> None:
        self.stream: Optional[torch.cuda.Stream] = None
        self.keep: list = []
        self.pending: list = []
        self._flushing = False
        self.enabled = False     # set by engine.TrainStep
        # linear-layer weight gradients: neutral for the LRW encoder, a gain for the LRS linears (engine.TrainStep turns it on there)
        self.enabled_small = False
        # launches are handed to the side stream in groups: every hand-over is one cross-stream dependency (an event record +
        # wait: host time in eager mode, a cross-branch edge in a captured graph), and a weight gradient is in no hurry
        self.group = int(os.environ.get("SVSR_SIDE_GROUP", "1"))     # measured (LRW, B = 32): eager 6.62 / 6.72 / 7.72 ms at 1 / 4 / all; graph 7.56 / 7.20 / 7.41

    def run(self, fn, *keep, small: bool = False) -> None:
        """fn must not depend on variables that are rebound before flush() (bind them as arguments); the tensors it reads
        must not be written in place before join()."""
        if not (self.enabled or (small and self.enabled_small)):
            fn()
            return
        if self._flushing:       # called from inside a function that is being issued on the side stream: issue it there, now, in order
            self.keep.extend(keep)
            fn()
            return
        self.pending.append(fn)
        self.keep.extend(keep)
        if len(self.pending) >= self.group:
            self.flush()

    def flush(self) -> None:
        if not self.pending or self._flushing:
            return
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        ops.stream_wait(self.stream, torch.cuda.current_stream())
        # launches go to the side stream through ops.STREAM_OVERRIDE (they allocate nothing on the device), which is
        # much cheaper on the host than entering a torch.cuda.stream() context per weight-gradient launch
        ops.STREAM_OVERRIDE = self.stream.cuda_stream
        self._flushing = True
        try:
            for fn in self.pending:
                fn()
        finally:
            self._flushing = False
            ops.STREAM_OVERRIDE = None
            self.pending.clear()

    def join(self) -> None:
        self.flush()
        if self.stream is not None and (self.enabled or self.enabled_small):
            ops.stream_wait(torch.cuda.current_stream(), self.stream)
        self.keep.clear()


class _Holder(nn.Module):
    """Name-space node: exists only so parameters get the reference's state-dict keys."""


def _attach(root: nn.Module, name: str, tensor: torch.Tensor, is_param: bool) -> None:
    parts = name.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Holder())
        mod = mod._modules[p]
    if is_param:
        mod.register_parameter(parts[-1], nn.Parameter(tensor))
    else:
        mod.register_buffer(parts[-1], tensor)


def _get(root: nn.Module, name: str) -> torch.Tensor:
    obj: Any = root
    for p in name.split("."):
        obj = obj._modules[p] if p in getattr(obj, "_modules", {}) else getattr(obj, p)
    return obj


class TransformerLightningModule(nn.Module):
    """HIP-native twin of the reference LightningModule (word-level LRW model with the SyncVSR audio head)."""

    def __init__(self, config: Config, seed: Optional[int] = None):
        super().__init__()
        if not isinstance(config, Config):
            config = Config(config)
        self.config = config
        bert = config.model.bert
        self.encoder_type = str(bert.type)
        if self.encoder_type not in ("huggingface", "x-transformers"):
            raise NotImplementedError(f"model.bert.type {bert.type!r}: the reference knows `huggingface` and `x-transformers` (lightning.py:90-105)")
        self.use_wb = bool(config.data.use_word_boundary)
        if self.use_wb and self.encoder_type == "huggingface":
            raise NotImplementedError(
                "use_word_boundary needs a 513-wide encoder, which the reference's huggingface branch cannot build either "
                "(BertConfig.hidden_size stays 512, lightning.py:92,145); use `type: x-transformers` as the shipped yaml does")
        self.is_train = False
        self.word_labels = int(bert.num_labels)
        self.lambda_audio = float(config.optim.lambda_audio)
        # audio head as one contraction with the cross-entropy in its epilogue (csrc/audio_head.hip) whenever the shape is taken; keep_audio_logits
        # additionally materialises logits_audio in `_last` (one extra projection launch, for inspection and the parity tests)
        self.fused_audio_head = True
        self.keep_audio_logits = False
        self.label_smoothing = float(config.train.label_smoothing)
        self.codec, self.audio_alignment, self.vq_groups, self.audio_vocab_size = audio_codec_dims(config.model.wav2vec.path)
        self.dim = hidden_dim(config)
        self.dim_p = pad64(self.dim)            # row pitch of the encoder's activations (pad columns are zero)
        self.dropout_seed = 0 if seed is None else int(seed)
        self._drop_word: Optional[torch.Tensor] = None
        # Dropout masks are counter based (csrc/common.h): element i of site s is kept iff hash(seed, s, i) >= p * 2^32; the seed is a
        # device word advanced once per training forward.
        from .dropout import lrw_sites, lrw_xt_sites

        self.emb_drop_p = float(bert.get("emb_dropout", 0.0))          # read directly by the module (lightning.py:45)
        if self.encoder_type == "huggingface":
            self.heads = int(bert.num_attention_heads)
            self.layers = int(bert.num_hidden_layers)
            self.inter = int(bert.intermediate_size)
            self.ln_eps = float(bert.get("layer_norm_eps", 1e-12))
            # missing keys take BertConfig(**config.model.bert)'s defaults, 0.1 each (lightning.py:92)
            self.drop_p = float(bert.get("hidden_dropout_prob", 0.1))
            self.attn_drop_p = float(bert.get("attention_probs_dropout_prob", 0.1))
            self.layer_drop_p = 0.0
            self._sites = lrw_sites(self.layers)
            if self.dim % 512 or self.dim // self.heads != 64:
                raise NotImplementedError("encoder width must be a multiple of 512 with 64-wide heads")
        else:
            # x_transformers.Encoder(dim, depth, heads, attn_dropout, layer_dropout, ff_dropout, use_rmsnorm, ff_glu, rotary_pos_emb)
            # (lightning.py:95-105).  The shipped combination (RMSNorm + GEGLU + rotary) is the one built here.
            if not (bool(bert.get("use_rmsnorm", False)) and bool(bert.get("ff_glu", False)) and bool(bert.get("rotary_pos_emb", False))):
                raise NotImplementedError("the x-transformers encoder is implemented for use_rmsnorm = ff_glu = rotary_pos_emb = true "
                                          "(the shipped yamls, bert-12l-512d_LRW_96_bf16_rrc_*.yaml:27-29)")
            _, self.attn_inner, self.inter, self.layers = xt_dims(config)
            self.heads = int(bert.heads)
            self.inter_p, self.glu_p = pad64(self.inter), pad64(2 * self.inter)
            self.drop_p = float(bert.get("ff_dropout", 0.0))            # FeedForward's dropout, after the gate
            self.attn_drop_p = float(bert.get("attn_dropout", 0.0))
            self.layer_drop_p = float(bert.get("layer_dropout", 0.0))
            self.final_norm = bool(bert.get("final_norm", False))
            self.rotate_value = bool(bert.get("rotate_value", True))   # x-transformers 1.x rotates q, k AND v
            self.rms_eps = 1e-8
            self._sites = lrw_xt_sites(self.layers)
            import random

            self._layer_rng = random.Random(self.dropout_seed)
            self.layer_skip_override: Optional[set[int]] = None     # tests: the set of `encoder.layers.{n}` indices to skip
            self._rot_tab: dict[tuple[int, str], torch.Tensor] = {}

        self._specs = param_specs(config)
        self._bspecs = buffer_specs(config)
        self._phys = {n: phys_shape(config, n, shp) for n, shp, _ in self._specs}
        sd = init_state_dict(config, seed=0 if seed is None else seed)
        for name, shape, kind in self._specs:
            t = sd[name]
            if kind == "conv" and len(shape) == 4:
                t = t.contiguous(memory_format=torch.channels_last)      # physical [Co][kh][kw][Ci]
            _attach(self, name, t, True)
        for name, shape, kind in self._bspecs:
            _attach(self, name, sd[name], False)
        from .augment import CutMix

        self.cutmix = CutMix(self.word_labels).eval()          # lightning.py:85-88
        self._store: Optional[_ParamStore] = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.mark_params_dirty())
        # (PRE hook: the copies of load_state_dict into the flat buffer must not race with an AdamW range still running on the side stream)
        self.register_load_state_dict_pre_hook(lambda module, *a, **k: module._side.join())
        self.stem_name, self.trunk_name = "stem3d", "resnet"
        self.stem_act, self.trunk_act = ops.ACT_GELU, ops.ACT_RELU          # lightning.py:52, timm BasicBlock
        self.use_tr = True          # ds_read_b64_tr_b16 fragments in the weight-gradient kernels
        self._side = _SideStream()  # second stream for the trunk's weight-gradient launches (._side.enabled = False serialises)
        self.grad_ready_hook = None  # called as hook(lo, hi) when flat gradient range [lo, hi) is final (DDP buckets)

    # ------------------------------------------------------------------------------------------------
    # reference-compatible helpers
    # ------------------------------------------------------------------------------------------------
    def configure_optimizers(self):
        """lightning.py:216-223 — returns the two parameter groups; the fused HIP optimiser lives in engine.TrainStep."""
        do_decay = [p for p in self.parameters() if p.requires_grad and p.ndim >= 2]
        no_decay = [p for p in self.parameters() if p.requires_grad and p.ndim < 2]
        return [{"params": do_decay}, {"params": no_decay, "weight_decay": 0.0}]

    def training_step(self, batch, idx: int = 0) -> torch.Tensor:
        """lightning.py:194-202: (CutMix) -> forward -> loss_total."""
        self.is_train = True
        if self.config.train.use_cutmix:
            batch = self.cutmix(*batch)
        return self(*batch)["loss_total"]

    def _advance_dropout(self, dev: torch.device) -> None:
        if self._drop_word is None or self._drop_word.device != dev:
            self._drop_word = torch.tensor([self.dropout_seed], dtype=torch.int32, device=dev)
        ops.word_add(self._drop_word, 1)          # a device op: graph / step-list replays keep drawing fresh masks

    def reseed_dropout(self, seed: int) -> None:
        self.dropout_seed = int(seed)
        if self._drop_word is not None:          # in place: a captured graph / recorded step list holds this word's address
            self._drop_word.fill_(self.dropout_seed)
        if hasattr(self, "_layer_rng"):          # the layer-drop draws follow the seed too (per-rank / per-run reseeding)
            self._layer_rng.seed(self.dropout_seed)

    def rng_state(self) -> dict:
        """Everything random about the next training forward, for bit-exact resume: the dropout seed word (the device counter the
        masks are hashed from) and the python generator behind layer_dropout.  engine.TrainStep.state_dict stores it."""
        word = self.dropout_seed if self._drop_word is None else int(self._drop_word.item())
        out = {"dropout_word": word}
        if hasattr(self, "_layer_rng"):
            out["layer_rng"] = self._layer_rng.getstate()
        return out

    def load_rng_state(self, state: dict) -> None:
        self.dropout_seed = int(state["dropout_word"])
        if self._drop_word is not None:
            self._drop_word.fill_(self.dropout_seed)
        if "layer_rng" in state and hasattr(self, "_layer_rng"):
            self._layer_rng.setstate(state["layer_rng"])

    def _d(self, site: str, kind: str = "hidden"):
        """(seed word, site id, p) for ops.*(drop=...) or None when dropout is off (eval mode / p = 0)."""
        p = {"hidden": self.drop_p, "attn": self.attn_drop_p, "emb": self.emb_drop_p}[kind]
        if not self.training or p <= 0.0:
            return None
        return (self._drop_word, self._sites[site], p)

    def state_dict(self, *args, **kwargs):
        self._side.join()             # (a TrainStep may have left the tail of its optimiser step on the side stream)
        return super().state_dict(*args, **kwargs)

    def mark_params_dirty(self) -> None:
        """Call after changing parameters outside engine.TrainStep (load_state_dict does it): the bf16 shadows are re-cast."""
        if self._store is not None:
            self._store.shadow_fresh = False

    @staticmethod
    def _fwd_rank(name: str) -> int:
        if name.startswith("stem3d"):
            return 0
        if name.startswith("resnet"):
            return 1
        if name == "cls_token" or name.startswith("encoder.embeddings"):
            return 2
        if name.startswith(("encoder.encoder", "encoder.layers", "encoder.final_norm")):
            return 3
        return 4

    def _transposed_entries(self, offsets) -> list[tuple[str, int, int, int]]:
        """(key, source offset, out features, in features) — storage sizes — of every linear weight whose data-gradient GEMM runs."""
        out = []
        if self.encoder_type == "x-transformers":
            Dp, E = self.dim_p, self.attn_inner
            for i in range(self.layers):
                a, f = f"encoder.layers.{2 * i}.1", f"encoder.layers.{2 * i + 1}.1.ff"
                q, k, v = (offsets[f"{a}.to_{x}.weight"][0] for x in "qkv")
                assert k == q + E * Dp and v == k + E * Dp, "to_q / to_k / to_v must be adjacent in the flat buffer"
                out.append((f"{a}.qkv", q, 3 * E, Dp))
                out.append((f"{a}.to_out.weight", offsets[f"{a}.to_out.weight"][0], Dp, E))
                out.append((f"{f}.0.proj.weight", offsets[f"{f}.0.proj.weight"][0], self.glu_p, Dp))
                out.append((f"{f}.3.weight", offsets[f"{f}.3.weight"][0], Dp, self.inter_p))
        else:
            D = self.dim
            for i in range(self.layers):
                p = f"encoder.encoder.layer.{i}"
                out.append((f"{p}.qkv", offsets[f"{p}.attention.self.query.weight"][0], 3 * D, D))
                out.append((f"{p}.attention.output.dense.weight", offsets[f"{p}.attention.output.dense.weight"][0], D, D))
                out.append((f"{p}.intermediate.dense.weight", offsets[f"{p}.intermediate.dense.weight"][0], self.inter, D))
                out.append((f"{p}.output.dense.weight", offsets[f"{p}.output.dense.weight"][0], D, self.inter))
                q, k, v = (offsets[f"{p}.attention.self.{x}.weight"][0] for x in ("query", "key", "value"))
                assert k == q + D * D and v == k + D * D, "q/k/v weights must be adjacent in the flat buffer"
                qb, kb, vb = (offsets[f"{p}.attention.self.{x}.bias"][0] for x in ("query", "key", "value"))
                assert kb == qb + D and vb == kb + D
        for n in ("audio_projection.weight", "category_classifier.weight"):
            out.append((n, offsets[n][0], offsets[n][2][0], self.dim_p))
        return out

    def store(self) -> "_ParamStore":
        dev = self.cls_token.device
        if self._store is None or self._store.device != dev or not self._store.owns(self):
            self._store = _ParamStore(self, dev)
        return self._store

    # ------------------------------------------------------------------------------------------------
    def forward_videos(self, videos: torch.Tensor) -> torch.Tensor:
        """lightning.py:112-119: [B,1,T,H,W] -> fp32 [B,T,512] (no autograd; use forward() for training)."""
        st = self.store()
        st.refresh_shadows()
        tape: dict[str, Any] = {}
        with torch.no_grad():
            feats = _frontend_forward(self, st, tape, videos.float().contiguous(), self.training)
        return feats.float().view(videos.size(0), -1, 512)

    def forward(self, videos: torch.Tensor, audio_tokens: torch.Tensor, labels: torch.Tensor, word_mask: torch.Tensor) -> dict[str, torch.Tensor]:
        if videos.device.type != "cuda":
            raise RuntimeError("syncvsr_amd runs on an MI355X HIP device only; there is no CPU fallback (use oracle/ for checking)")
        st = self.store()
        T = videos.size(2)
        A = self.audio_alignment
        audio_tokens = audio_tokens[:, : T * A].contiguous()
        if audio_tokens.size(1) != T * A:
            raise ValueError(f"audio_tokens has {audio_tokens.size(1)} steps, need >= {T * A}")
        wm = None
        if self.use_wb:           # lightning.py:145: the word-boundary indicator becomes feature 513 of every frame
            if word_mask.shape != (videos.size(0), T):
                raise ValueError(f"word_mask must be [batch, frames] = {(videos.size(0), T)}, got {tuple(word_mask.shape)}")
            wm = word_mask.to(device=videos.device, dtype=torch.float32).contiguous()
        outs = _LrwFunction.apply(self.cls_token, self, st, videos.float().contiguous(), audio_tokens, labels.contiguous(),
                                   torch.is_grad_enabled(), wm)
        loss_category, loss_audio, acc = outs
        loss_total = loss_category + loss_audio * self.lambda_audio
        return {
            "loss_total": loss_total,
            "loss_category": loss_category,
            "loss_audio": loss_audio,
            "accuracy_top1": acc[0],
            "accuracy_top5": acc[1],
        }


    # ------------------------------------------------------------------------------------------------
    def prepare_batch(self, videos, audio_tokens, labels, word_mask):
        """The input conversions of forward(), done ahead of it: what engine.TrainStep keeps as the static inputs of a recorded step
        (inside the recorded region every one of them must be a no-op, so that no torch kernel is needed at replay)."""
        T = videos.size(2)
        A = self.audio_alignment
        if audio_tokens.size(1) < T * A:
            raise ValueError(f"audio_tokens has {audio_tokens.size(1)} steps, need >= {T * A}")
        hard = labels.dtype in (torch.int64, torch.int32)
        wm = word_mask
        if self.use_wb:
            if word_mask.shape != (videos.size(0), T):
                raise ValueError(f"word_mask must be [batch, frames] = {(videos.size(0), T)}, got {tuple(word_mask.shape)}")
            wm = word_mask.to(device=videos.device, dtype=torch.float32).contiguous()
        return (videos.float().contiguous(), audio_tokens[:, : T * A].contiguous(), (labels.long() if hard else labels.float()).contiguous(), wm)

    def direct_constants(self, dev) -> None:
        """The two loss weights d loss_total / d loss_{category, audio} as device scalars (made once, outside any recorded region)."""
        if getattr(self, "_g_one", None) is None or self._g_one.device != dev:
            self._g_one = torch.ones((), dtype=torch.float32, device=dev)
            self._g_lam = torch.full((), self.lambda_audio, dtype=torch.float32, device=dev)

    def train_step_direct(self, videos, audio_tokens, labels, word_mask) -> dict[str, torch.Tensor]:
        """forward + backward of loss_total WITHOUT autograd: the same tape functions `forward()` + `loss_total.backward()` run,
        called directly, with loss_total and the two loss weights formed on the device.  Inputs as prepare_batch returns them.
        Every device operation in here is a library call, so the whole step can be recorded into a native step list."""
        if videos.device.type != "cuda":
            raise RuntimeError("syncvsr_amd runs on an MI355X HIP device only; there is no CPU fallback (use oracle/ for checking)")
        st = self.store()
        self.direct_constants(videos.device)

        class _Ctx:
            def mark_non_differentiable(self, *a):
                pass

        ctx = _Ctx()
        self._metrics_on_side = True          # what only the caller reads (accuracy, loss_total) is computed on the side stream: joined when the backward ends
        try:
            loss_category, loss_audio, acc = _LrwFunction.forward(ctx, self.cls_token, self, st, videos, audio_tokens, labels, True,
                                                                  word_mask if self.use_wb else None)
        finally:
            self._metrics_on_side = False
        if self._side.enabled and METRICS_ON_SIDE:
            box: dict = {}
            self._side.run(lambda: box.__setitem__("t", ops.lincomb2(loss_category, loss_audio, self.lambda_audio)), loss_category, loss_audio)
            self._side.flush()
            loss_total = box["t"]
        else:
            loss_total = ops.lincomb2(loss_category, loss_audio, self.lambda_audio)
        _LrwFunction.backward(ctx, self._g_one, self._g_lam, None)
        return {"loss_total": loss_total, "loss_category": loss_category, "loss_audio": loss_audio, "accuracy_top1": acc[0],
                "accuracy_top5": acc[1]}


Model = TransformerLightningModule


# ----------------------------------------------------------------------------------------------------
# parameter store: flat fp32 master / gradient buffers + bf16 shadows
# ----------------------------------------------------------------------------------------------------
class _ParamStore:
    """Flat fp32 master / gradient buffers + bf16 shadows for one model.  The model supplies `_specs`, `_bspecs`,
    `_fwd_rank(name)` (position of a tensor in the forward pass) and `_transposed_entries(offsets)` (which linear weights
    need a transposed bf16 shadow for their data-gradient GEMM, and which tensors must be adjacent)."""

    def __init__(self, model, device: torch.device):
        self.device = device
        specs = model._specs
        # flat order: [decayed (ndim >= 2)] then [not decayed]; q/k/v of a layer adjacent so one GEMM covers them
        # decayed tensors are laid out in FORWARD order (stem, trunk, embeddings, encoder layers, heads) so the backward
        # pass finalises the buffer from its end towards its start — contiguous all-reduce buckets (engine.DataParallel).
        fwd_rank = model._fwd_rank
        decay = sorted([(n, s) for n, s, k in specs if len(s) >= 2], key=lambda e: fwd_rank(e[0]))   # stable
        nodecay = [(n, s) for n, s, k in specs if len(s) < 2]
        self.offsets: dict[str, tuple[int, int, tuple[int, ...]]] = {}
        self._side = getattr(model, "_side", None)      # a TrainStep may leave the tail of its optimiser step there: whoever reads or rewrites flat / w16 from the main stream joins it first
        self._vc: dict = {}        # cached views of the flat buffers (p32 / g32 / s16 / t16)
        off = 0
        phys = getattr(model, "_phys", {})
        self.phys: dict[str, tuple[int, ...]] = {n: tuple(phys.get(n, s)) for n, s, _ in specs}
        for n, s in decay + nodecay:
            numel = math.prod(self.phys[n])          # storage size: rows / columns padded to 64 where the model asks for it
            off = (off + 3) // 4 * 4                    # 16-byte alignment of every tensor
            self.offsets[n] = (off, numel, tuple(s))
            off += numel
            if (n, s) == decay[-1]:
                off = (off + 3) // 4 * 4
                self.decay_end = off
        self.numel = (off + 3) // 4 * 4
        # [0, sumsq_head): the stem convolution's weight, the LAST gradient of a backward pass: the global norm's sum of squares over everything
        # behind it can run while that gradient is still being computed (engine.TrainStep._optimizer; 0: the buffer does not start with it)
        first = (decay + nodecay)[0][0]
        self.sumsq_head = self.offsets[first][1] if first == getattr(model, "stem_name", "?") + ".0.weight" and self.offsets[first][1] % 4 == 0 else 0
        # [0, front_end): the visual front-end's weights (forward ranks 0 and 1), what the next forward needs first (engine.TrainStep updates the
        # rest on the side stream beside that forward)
        later = [self.offsets[n][0] for n, s in decay if fwd_rank(n) >= 2]
        self.front_end = min(later) if later else self.decay_end
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.w16 = torch.zeros(self.numel, dtype=BF16, device=device)
        self._params: dict[str, nn.Parameter] = {}
        for n, s, kind in specs:
            p = _get(model, n)
            o, numel, shape = self.offsets[n]
            view = self._view(self.flat, n)
            view.copy_(p.data.to(device))
            p.data = view
            p.grad = self._view(self.grad, n)
            self._params[n] = p
        self.buffers = {n: _get(model, n) for n, _, _ in model._bspecs}
        for n, b in self.buffers.items():
            if b.device != device:
                raise RuntimeError("move the module to the GPU before the first forward (model.to('cuda'))")
        # the floating-point buffers (BatchNorm running statistics) are views of ONE flat vector, so data-parallel training can
        # broadcast them from rank 0 in a single collective (DDP's broadcast_buffers, engine.GradReducer)
        fbufs = [(n, b) for n, b in self.buffers.items() if b.is_floating_point()]
        self.bufflat = torch.empty(sum(b.numel() for _, b in fbufs), dtype=torch.float32, device=device)
        o = 0
        for n, b in fbufs:
            view = self.bufflat[o : o + b.numel()].view(b.shape)
            view.copy_(b)
            b.data = view
            o += b.numel()
        # transposed shadows for the data-gradient contractions
        entries = []
        toff = 0
        self.t_offsets: dict[str, tuple[int, tuple[int, int, int]]] = {}

        def add_t(key: str, src_off: int, A: int, T: int, Bd: int):
            nonlocal toff
            Apad = (A + 63) // 64 * 64
            self.t_offsets[key] = (toff, (Bd, T, Apad))
            entries.append((src_off, toff, A, T, Bd, Apad))
            toff += Bd * T * Apad

        for n, s, kind in specs:
            if kind == "conv" and len(s) == 4:
                add_t(n, self.offsets[n][0], s[0], s[2] * s[3], s[1])
        self.n_entries_front = sum(1 for e in entries if e[0] < self.front_end)       # the front-end's entries come first
        assert all(e[0] < self.front_end for e in entries[: self.n_entries_front]) and all(e[0] >= self.front_end for e in entries[self.n_entries_front:])
        for key, src_off, A, Bd in model._transposed_entries(self.offsets):
            add_t(key, src_off, A, 1, Bd)
        self.w16t = torch.zeros(max(toff, 1), dtype=BF16, device=device)
        import numpy as np

        tab = np.zeros(len(entries), dtype=np.dtype([("src", "<i8"), ("dst", "<i8"), ("A", "<i4"), ("T", "<i4"), ("Bd", "<i4"), ("Apad", "<i4")]))
        for i, e in enumerate(entries):
            tab[i] = e
        self.table = torch.from_numpy(tab.view(np.uint8).copy()).to(device)
        self.n_entries = len(entries)
        # per-BatchNorm workspaces
        self.bn: dict[str, dict[str, torch.Tensor]] = {}
        for n, s, kind in specs:
            if kind == "norm_w" and (n + "").replace(".weight", ".running_mean") in self.buffers:
                C = s[0]
                base = n[: -len(".weight")]
                self.bn[base] = dict(
                    coef=torch.empty(3 * C, dtype=torch.float32, device=device),
                    mean=torch.empty(C, dtype=torch.float32, device=device),
                    rstd=torch.empty(C, dtype=torch.float32, device=device),
                )
        self.shadow_fresh = False     # True only while an optimiser that writes the shadows itself owns the step loop
        self.generation = 0           # counts weight updates (optimiser steps, shadow refreshes): caches derived from the weights key on it

    def owns(self, model) -> bool:
        lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + self.flat.numel() * 4
        anchor = model._specs[0][0]
        blo, bhi = self.bufflat.data_ptr(), self.bufflat.data_ptr() + self.bufflat.numel() * 4
        bufs_ok = all(blo <= b.data_ptr() < bhi for b in self.buffers.values() if b.is_floating_point() and b.numel())
        return bufs_ok and all(lo <= p.data_ptr() < hi for p in self._params.values()) and _get(model, anchor) is self._params[anchor]

    def _view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        o, numel, shape = self.offsets[name]
        seg = flat[o : o + numel]
        if len(shape) == 4:   # conv weights are stored [Co][kh][kw][Ci]; the tensor keeps the logical [Co,Ci,kh,kw] shape
            return seg.view(shape[0], shape[2], shape[3], shape[1]).permute(0, 3, 1, 2)
        ph = self.phys[name]
        if ph != shape:       # padded storage: the tensor is the [:logical] corner of it
            return seg.view(ph)[tuple(slice(0, d) for d in shape)]
        return seg.view(shape)

    # -- accessors used by the engine -------------------------------------------------------------
    # (the views are cached: the flat buffers are never reallocated, and building ~400 tensor views per step was 0.3 ms of host time)
    def p32(self, name: str) -> torch.Tensor:
        v = self._vc.get(("p", name))
        if v is None:
            o, n, _ = self.offsets[name]
            v = self._vc[("p", name)] = self.flat[o : o + n]
        return v

    def g32(self, name: str) -> torch.Tensor:
        v = self._vc.get(("g", name))
        if v is None:
            o, n, _ = self.offsets[name]
            v = self._vc[("g", name)] = self.grad[o : o + n]
        return v

    def s16(self, name: str, numel: Optional[int] = None) -> torch.Tensor:
        v = self._vc.get(("s", name, numel))
        if v is None:
            o, n, _ = self.offsets[name]
            v = self._vc[("s", name, numel)] = self.w16[o : o + (numel or n)]
        return v

    def t16(self, key: str) -> torch.Tensor:
        v = self._vc.get(("t", key))
        if v is None:
            o, (Bd, T, Apad) = self.t_offsets[key]
            v = self._vc[("t", key)] = self.w16t[o : o + Bd * T * Apad].view(Bd, T, Apad)
        return v

    def span(self, name: str, numel: Optional[int] = None) -> tuple[int, int]:
        o, n, _ = self.offsets[name]
        return o, o + (numel or n)

    def refresh_shadows(self) -> None:
        if self._side is not None:
            self._side.join()         # the side stream's AdamW range may still be writing flat / w16 (engine.TrainStep._optimizer)
        self.generation += 1
        ops.cast_bf16(self.flat, self.w16)
        ops.transpose_shadows(self.flat, self.w16, self.w16t, self.table, self.n_entries)

    def zero_grad(self) -> None:
        if self.__dict__.get("grad_clean", False):       # engine.TrainStep zeroed the buffer on the side stream when the step began
            self.grad_clean = False
            return
        ops.memset(self.grad, 0)

    def rebind_grads(self) -> None:
        for n, p in self._params.items():
            g = p.grad
            if g is None or g.data_ptr() != self.grad.data_ptr() + self.offsets[n][0] * 4:
                p.grad = self._view(self.grad, n)


# ----------------------------------------------------------------------------------------------------
# forward / backward tape
# ----------------------------------------------------------------------------------------------------
def _bn_stats(st: _ParamStore, base: str, training: bool, count: int, stats):
    """stats: the partial sums (buffer, rows) the producing convolution just wrote on this stream (training mode)."""
    ws = st.bn[base]
    if training:
        ops.bn_finalize(stats, ws["mean"].numel(), count, ws["mean"], ws["rstd"], st.buffers[f"{base}.running_mean"],
                        st.buffers[f"{base}.running_var"], st.buffers[f"{base}.num_batches_tracked"])
    else:
        ops.bn_eval_prepare(st.buffers[f"{base}.running_mean"], st.buffers[f"{base}.running_var"], ws["mean"], ws["rstd"])
    return ws["mean"], ws["rstd"]      # per-layer buffers: valid until this layer's next forward


def _conv_bn(st: _ParamStore, tape: dict, x: torch.Tensor, conv: str, bn: str, k: int, stride: int, pad: int, training: bool,
             res: Optional[torch.Tensor], act: int) -> torch.Tensor:
    o, n, shape = st.offsets[f"{conv}.weight"]
    w16 = st.w16[o : o + n].view(shape[0], k, k, shape[1])
    c, stats = ops.conv2d_fwd(x, w16, k, stride, pad, want_stats=training)
    mean, rstd = _bn_stats(st, bn, training, c.numel() // c.shape[-1], stats)
    y = ops.bn_act_fwd(c, res, mean, rstd, st.p32(f"{bn}.weight"), st.p32(f"{bn}.bias"), act)
    tape[conv] = dict(x=x, c=c, y=y, mean=mean, rstd=rstd, k=k, stride=stride, pad=pad, act=act, bn=bn, res=res)
    return y


def _trunk_blocks(model):
    """(prefix, inplanes, planes, stride, has_downsample) with the model's own trunk name (`resnet` for LRW,
    `encoder.frontend.trunk` for LRS — same BasicBlock topology, different activation)."""
    for prefix, inp, planes, stride, down in resnet_block_specs():
        yield model.trunk_name + prefix[len("resnet"):], inp, planes, stride, down


def _frontend_forward(model, st: "_ParamStore", tape: dict, videos: torch.Tensor, training: bool) -> torch.Tensor:
    """videos: fp32 [B,1,T,H,W] (LRW) — the LRS layout [B,T,1,H,W] is the same memory.  -> [B*T, 512] bf16."""
    B, _, T, H, W = videos.shape
    N = B * T
    sc, sb, act = model.stem_name + ".0", model.stem_name + ".1", model.trunk_act
    c, stats = ops.stem_conv_fwd(videos, st.p32(f"{sc}.weight"), want_stats=training)
    mean, rstd = _bn_stats(st, sb, training, c.numel() // 64, stats)
    if training and ops.STEM_KEEP_WINNERS:
        x, amax, xwin = ops.stem_bn_gelu_pool_fwd(c, mean, rstd, st.p32(f"{sb}.weight"), st.p32(f"{sb}.bias"), model.stem_act, want_win=True)
    else:
        (x, amax), xwin = ops.stem_bn_gelu_pool_fwd(c, mean, rstd, st.p32(f"{sb}.weight"), st.p32(f"{sb}.bias"), model.stem_act), None
    tape["stem"] = dict(videos=videos, c=c, amax=amax, xwin=xwin, mean=mean, rstd=rstd, pooled_shape=tuple(x.shape))
    for prefix, inp, planes, stride, down in _trunk_blocks(model):
        xin = x
        if down and training and model._side.enabled and DOWN_ON_SIDE:
            # (round 6) the downsample branch (1x1 convolution, its statistics and apply pass) meets the main branch only at the residual sum: on the
            # side stream beside conv1 / bn1 (3 blocks x 3 launches off the main chain: -0.035 ms LRW same box, LRS unchanged; the backward
            # twin measured SLOWER — its join waits for the weight gradients queued on that stream — and is not here)
            box: dict = {}
            model._side.run(lambda xin=xin, prefix=prefix, stride=stride: box.__setitem__(
                "idt", _conv_bn(st, tape, xin, f"{prefix}.downsample.0", f"{prefix}.downsample.1", 1, stride, 0, training, None, 0)), xin)
            model._side.flush()
            o1 = _conv_bn(st, tape, xin, f"{prefix}.conv1", f"{prefix}.bn1", 3, stride, 1, training, None, act)
            model._side.join()
            idt = box["idt"]
        else:
            o1 = _conv_bn(st, tape, xin, f"{prefix}.conv1", f"{prefix}.bn1", 3, stride, 1, training, None, act)
            idt = _conv_bn(st, tape, xin, f"{prefix}.downsample.0", f"{prefix}.downsample.1", 1, stride, 0, training, None, 0) if down else xin
        x = _conv_bn(st, tape, o1, f"{prefix}.conv2", f"{prefix}.bn2", 3, 1, 1, training, idt, act)
    tape["trunk_out_shape"] = tuple(x.shape)
    return ops.avgpool_fwd(x)            # [N, 512] bf16


def _frontend_backward(model, st: "_ParamStore", tape: dict, dfeats: torch.Tensor) -> None:
    use_tr = model.use_tr
    act = model.trunk_act
    dx = ops.avgpool_bwd(dfeats, tape["trunk_out_shape"])
    # ReLU trunk: the launch that produces the gradient of a BatchNorm+ReLU output also masks it and takes the first pass of that
    # BatchNorm's backward in its epilogue (ops.conv2d_dgrad_bn); dx_stats != None means dx already is that masked gradient.
    fused = act in (1, 2) and ops.BN_BWD_FUSED       # 1 ReLU (LRW), 2 Swish (LRS): the epilogue masks / multiplies by swish'
    dx_stats = None
    blocks = list(_trunk_blocks(model))
    rec = tape.get("_record_grads")       # tests: {block prefix: (gradient entering the block from above, is it already masked by relu')}
    for bi in range(len(blocks) - 1, -1, -1):
        prefix, inp, planes, stride, down = blocks[bi]
        if rec is not None:
            rec[prefix] = (dx.clone(), dx_stats is not None)
        t2 = tape[f"{prefix}.conv2"]
        ws2 = st.bn[t2["bn"]]
        if dx_stats is not None:
            dres = dx
            dc2 = ops.bn_bwd_from_stats(dx, t2["c"], t2["mean"], t2["rstd"], st.p32(f"{t2['bn']}.weight"), dx_stats, ws2["coef"],
                                        st.g32(f"{t2['bn']}.weight"), st.g32(f"{t2['bn']}.bias"))
        else:
            dc2, dres = ops.bn_act_bwd(dx, t2["y"], t2["c"], t2["mean"], t2["rstd"], st.p32(f"{t2['bn']}.weight"), ws2["coef"],
                                       st.g32(f"{t2['bn']}.weight"), st.g32(f"{t2['bn']}.bias"), act, True,
                                       beta=st.p32(f"{t2['bn']}.bias"), res=t2["res"])
        _conv_wgrad(model, st, f"{prefix}.conv2", t2, dc2, use_tr)
        t1 = tape[f"{prefix}.conv1"]
        ws1 = st.bn[t1["bn"]]
        w2t = st.t16(f"{prefix}.conv2.weight").view(planes, 3, 3, planes)
        if fused:
            # bn1 has no residual branch: the mask / pre-activation is recomputed from its input (y is not read)
            g1, st1 = ops.conv2d_dgrad_bn(dc2, w2t, 3, 1, 1, t2["x"].shape[1:3], None, None, t1["c"], t1["mean"], t1["rstd"],
                                          st.p32(f"{t1['bn']}.weight"), st.p32(f"{t1['bn']}.bias"), act)
            dc1 = ops.bn_bwd_from_stats(g1, t1["c"], t1["mean"], t1["rstd"], st.p32(f"{t1['bn']}.weight"), st1, ws1["coef"],
                                        st.g32(f"{t1['bn']}.weight"), st.g32(f"{t1['bn']}.bias"))
        else:
            do1 = ops.conv2d_dgrad(dc2, w2t, 3, 1, 1, t2["x"].shape[1:3])
            dc1, _ = ops.bn_act_bwd(do1, t1["y"], t1["c"], t1["mean"], t1["rstd"], st.p32(f"{t1['bn']}.weight"), ws1["coef"],
                                    st.g32(f"{t1['bn']}.weight"), st.g32(f"{t1['bn']}.bias"), act, False, beta=st.p32(f"{t1['bn']}.bias"))
        _conv_wgrad(model, st, f"{prefix}.conv1", t1, dc1, use_tr)
        in_hw = t1["x"].shape[1:3]
        w1t = st.t16(f"{prefix}.conv1.weight").view(inp, 3, 3, planes)
        tp = tape[f"{blocks[bi - 1][0]}.conv2"] if fused and bi > 0 else None       # the block below: its output is this block's input
        if tp is not None:      # what the epilogue needs of that output: ReLU its value (the mask), Swish its residual input
            tp_aux = tp["y"] if act == 1 else tp["res"]
            tp_gb = (st.p32(f"{tp['bn']}.weight"), st.p32(f"{tp['bn']}.bias"), act)
        if down:
            td = tape[f"{prefix}.downsample.0"]
            wsd = st.bn[td["bn"]]
            dcd, _ = ops.bn_act_bwd(dres, None, td["c"], td["mean"], td["rstd"], st.p32(f"{td['bn']}.weight"), wsd["coef"],
                                    st.g32(f"{td['bn']}.weight"), st.g32(f"{td['bn']}.bias"), 0, False)
            _conv_wgrad(model, st, f"{prefix}.downsample.0", td, dcd, use_tr)
            wdt = st.t16(f"{prefix}.downsample.0.weight").view(inp, 1, 1, planes)
            # the 1x1 branch first (pixels its stride skips are written as zeros), the 3x3 branch adds to it: the launch that completes
            # the block's input gradient is the one with the taps that reach every pixel
            dxd = ops.conv2d_dgrad(dcd, wdt, 1, stride, 0, in_hw)
            if tp is not None:
                dx, dx_stats = ops.conv2d_dgrad_bn(dc1, w1t, 3, stride, 1, in_hw, dxd, tp_aux, tp["c"], tp["mean"], tp["rstd"], *tp_gb)
            else:
                dx, dx_stats = ops.conv2d_dgrad(dc1, w1t, 3, stride, 1, in_hw, addend=dxd), None
        elif tp is not None:
            dx, dx_stats = ops.conv2d_dgrad_bn(dc1, w1t, 3, stride, 1, in_hw, dres, tp_aux, tp["c"], tp["mean"], tp["rstd"], *tp_gb)
        else:
            dx, dx_stats = ops.conv2d_dgrad(dc1, w1t, 3, stride, 1, in_hw, addend=dres), None
        _ready(model, st, f"{prefix}.conv1.weight")
    ts = tape["stem"]
    if rec is not None:
        rec["stem"] = (dx.clone(), False)
    sc, sb = model.stem_name + ".0", model.stem_name + ".1"
    ws = st.bn[sb]
    fused = ts.get("xwin") is not None and use_tr and ops.stem_bwd_wgrad_ok(ts["videos"])
    dconv = ops.stem_bn_gelu_pool_bwd(dx, ts["amax"], ts["c"], ts["mean"], ts["rstd"], st.p32(f"{sb}.weight"), st.p32(f"{sb}.bias"),
                                      ws["coef"], st.g32(f"{sb}.weight"), st.g32(f"{sb}.bias"), model.stem_act, xwin=ts.get("xwin"),
                                      want_dx=not fused)
    if fused:
        # The gradient of the convolution output is made tile by tile inside the weight-gradient pass, never written.  The side stream is
        # joined FIRST: the pass is a persistent grid of two 248-register workgroups per CU, and beside the last trunk weight gradients
        # (8-wave workgroups that leave no CU room for them) half its workgroups started late and the step LOST 0.02 / 0.4 ms (LRW / LRS);
        # alone it is 223 us against 105 + 134 (928 frames) and the sentence-level step gains 0.2 ms.
        _flush_deferred(model)
        model._side.join()
        early = getattr(model, "_early_sumsq", None)      # the optimiser's device state, set by engine.TrainStep when no collective follows
        if early is not None and st.sumsq_head:
            # every gradient but the stem convolution's is final: their sum of squares (the global-norm clip's) runs beside that last pass
            model._side.run(lambda: ops.grad_sumsq_parts(st.grad, st.sumsq_head, st.numel - st.sumsq_head, early, 0, ops.SUMSQ_PARTS - 1))
            model._side.flush()
            st.sumsq_tail_done = True
        ops.stem_bwd_wgrad(ts["videos"], dconv, ts["amax"], ts["c"], ts["mean"], ts["rstd"], ws["coef"], st.g32(f"{sc}.weight"))
    else:
        ops.stem_conv_wgrad(ts["videos"], dconv, st.g32(f"{sc}.weight"), use_tr)
    _flush_deferred(model)
    model._side.join()
    _ready(model, st, None)


# Timing experiments only (results WRONG): a probe script may set this to {"conv_wgrad"} / {"lin_wgrad"} to skip those launches.
# Deliberately not an environment switch: nothing outside an explicit assignment in a probe can turn gradients off.
_ABLATE: frozenset = frozenset()


def _conv_wgrad(model, st: "_ParamStore", conv: str, t: dict, dc: torch.Tensor, use_tr: bool) -> None:
    if "conv_wgrad" in _ABLATE:
        return
    model._side.run(lambda: ops.conv2d_wgrad(t["x"], dc, st.g32(f"{conv}.weight"), t["k"], t["stride"], t["pad"], use_tr), dc)


def _lin_wgrad(model, x, dy, gw, gb, rows: int, K: int, N: int, x_pitch: int, dy_pitch: int) -> None:
    """Weight (+ bias) gradient of an encoder linear layer, handed to the side stream (a function call, so that the deferred
    launch sees THIS layer's tensors and not the loop variables of a later one)."""
    if "lin_wgrad" in _ABLATE:
        return
    group = getattr(model, "_wg_group", None)
    if group is not None:          # collected: one grouped launch at the end of the encoder's backward (_flush_lin_wgrads)
        group.append(dict(x=x, dy=dy, dw=gw, db=gb, rows=rows, K=K, N=N, x_pitch=x_pitch, dy_pitch=dy_pitch))
        return
    model._side.run(lambda: ops.linear_wgrad(x, dy, gw, rows=rows, K=K, N=N, x_pitch=x_pitch, dy_pitch=dy_pitch, db=gb), x, dy, small=True)


HEADS_CONCURRENT = os.environ.get("SVSR_HEADS_CONCURRENT", "1") != "0"      # train_step_direct: the word head (forward and backward) on the side stream beside the audio head
DOWN_ON_SIDE = os.environ.get("SVSR_DOWN_ON_SIDE", "1") != "0"      # _frontend_forward: a block's downsample branch (1x1 convolution + its BatchNorm) on the side stream beside conv1 / bn1
METRICS_ON_SIDE = os.environ.get("SVSR_METRICS_ON_SIDE", "1") != "0"      # train_step_direct: accuracy and loss_total (read by the caller only) on the side stream
DEFER_REDUCTIONS = os.environ.get("SVSR_DEFER_REDUCTIONS", "1") != "0"     # encoder backward: parameter-gradient reductions on the side stream


WG_GROUP_LAYERS = int(os.environ.get("SVSR_WG_GROUP_LAYERS", "1"))      # encoder layers per grouped weight-gradient launch (0: one launch for the whole encoder)


def _flush_lin_wgrads(model) -> None:
    """The collected linear weight gradients of this backward pass as one launch (ops.linear_wgrad_group), on the side stream when
    the trunk's weight gradients go there: nothing downstream reads them before the optimiser."""
    group, model._wg_group = model._wg_group, None
    if not group:
        return
    keep = [t for q in group for t in (q["x"], q["dy"])]
    if model._side.enabled:
        model._side.run(lambda: ops.linear_wgrad_group(group), *keep)
    else:
        ops.linear_wgrad_group(group)


def _defer_list(model) -> Optional[list]:
    """The list ops.add_ln_bwd / ops.bias_act_bwd append their postponed reductions to (None: reduce in line)."""
    if not DEFER_REDUCTIONS or not (model._side.enabled or model._side.enabled_small):
        return None
    d = model.__dict__.get("_deferred")
    if d is None:
        d = model.__dict__["_deferred"] = []
    return d


def _flush_deferred(model) -> None:
    d = model.__dict__.get("_deferred")
    if d:
        fns, keep = [f for f, _ in d], [k for _, k in d]
        d.clear()
        model._side.run(lambda: ops.run_deferred(fns), *keep, small=True)


def _ready(model, st: "_ParamStore", name: Optional[str]) -> None:
    """Gradient-ready notification for bucketed all-reduce: everything at or above `name`'s offset in the decayed
    region is final (backward walks the flat buffer from its end to its start); None = all gradients final."""
    _flush_deferred(model)                    # postponed parameter-gradient reductions: to the side stream, once per layer
    if model.grad_ready_hook is not None:     # (the reducer's comm stream waits for the side stream itself: engine.GradReducer._reduce)
        model._side.flush()
        hook, lo = model.grad_ready_hook, (0 if name is None else st.offsets[name][0])
        ops.host_callback(lambda: hook(lo))   # a host-side step (collective): a segment boundary of a recorded step list


def _encoder_forward(model: TransformerLightningModule, st: _ParamStore, tape: dict, feats: torch.Tensor, B: int, T: int) -> torch.Tensor:
    D, S, H = model.dim, T + 1, model.heads
    R = B * S
    pos = st.p32("encoder.embeddings.position_embeddings.weight")
    type0 = st.p32("encoder.embeddings.token_type_embeddings.weight")
    d_in, d_out = model._d("emb.in", "emb"), model._d("emb.out")
    s0, x, mean0, rstd0 = ops.embed_ln_fwd(feats, st.p32("cls_token"), pos, type0, st.p32("encoder.embeddings.LayerNorm.weight"),
                                           st.p32("encoder.embeddings.LayerNorm.bias"), B, S, D, model.ln_eps, drop_in=d_in, drop_out=d_out)
    tape["emb"] = dict(sum=s0, mean=mean0, rstd=rstd0, d_in=d_in, d_out=d_out)
    if ops.enc_fused_ok(D, H, model.inter, S):
        return _encoder_forward_fused(model, st, tape, x, B, S)
    for i in range(model.layers):
        p = f"encoder.encoder.layer.{i}"
        wqkv = st.s16(f"{p}.attention.self.query.weight", 3 * D * D)
        bqkv = st.flat[st.offsets[f"{p}.attention.self.query.bias"][0] :][: 3 * D]
        qkv, _ = ops.linear_fwd(x, wqkv, bqkv, rows=R, K=D, N=3 * D, x_pitch=D)
        dpr, dao, dfo = model._d(f"enc.{i}.attn.probs", "attn"), model._d(f"enc.{i}.attn.out"), model._d(f"enc.{i}.ff.out")
        ctx, probs = ops.mha_fwd(qkv, 3 * D, qkv[:, D:], qkv[:, 2 * D:], 3 * D, B=B, H=H, Lq=S, Lk=S, drop=dpr)      # csrc/mha.hip
        ao, _ = ops.linear_fwd(ctx, st.s16(f"{p}.attention.output.dense.weight"), st.p32(f"{p}.attention.output.dense.bias"),
                               rows=R, K=D, N=D, x_pitch=D, drop=dao)                      # BertSelfOutput: dropout(dense(ctx))
        x1, m1, r1 = ops.add_ln_fwd(ao, x, st.p32(f"{p}.attention.output.LayerNorm.weight"), st.p32(f"{p}.attention.output.LayerNorm.bias"), model.ln_eps)
        hg, z = ops.linear_fwd(x1, st.s16(f"{p}.intermediate.dense.weight"), st.p32(f"{p}.intermediate.dense.bias"),
                               rows=R, K=D, N=model.inter, x_pitch=D, gelu=True)
        f, _ = ops.linear_fwd(hg, st.s16(f"{p}.output.dense.weight"), st.p32(f"{p}.output.dense.bias"), rows=R, K=model.inter, N=D,
                              x_pitch=model.inter, drop=dfo)                               # BertOutput: dropout(dense(h))
        x2, m2, r2 = ops.add_ln_fwd(f, x1, st.p32(f"{p}.output.LayerNorm.weight"), st.p32(f"{p}.output.LayerNorm.bias"), model.ln_eps)
        tape[p] = dict(x=x, qkv=qkv, ctx=ctx, probs=probs, ao=ao, x1=x1, m1=m1, r1=r1, z=z, hg=hg, f=f, m2=m2, r2=r2, dpr=dpr, dao=dao, dfo=dfo)
        x = x2
    return x


def _encoder_forward_fused(model: TransformerLightningModule, st: _ParamStore, tape: dict, x: torch.Tensor, B: int, S: int) -> torch.Tensor:
    """All encoder layers in one launch (ops.enc_fwd, csrc/enc_fused.hip); fills the same tape entries as the per-layer path."""
    D, H, I, Lc = model.dim, model.heads, model.inter, model.layers
    R = B * S
    dev = x.device
    ldp = ops.probs_pitch(S)
    # one allocation per tensor kind for all layers
    qkv = torch.empty((Lc, R, 3 * D), dtype=BF16, device=dev)
    probs = torch.empty((Lc, B * H, S, ldp), dtype=BF16, device=dev)
    act = torch.empty((5, Lc, R, D), dtype=BF16, device=dev)               # ctx, ao, x1, f, xout
    zz = torch.empty((2, Lc, R, I), dtype=BF16, device=dev)                # z, hg
    stat = torch.empty((4, Lc, R), dtype=torch.float32, device=dev)        # m1, r1, m2, r2
    training = model.training
    recs = []
    xin = x
    for i in range(Lc):
        p = f"encoder.encoder.layer.{i}"
        dpr, dao, dfo = model._d(f"enc.{i}.attn.probs", "attn"), model._d(f"enc.{i}.attn.out"), model._d(f"enc.{i}.ff.out")
        rec = dict(
            wqkv=st.s16(f"{p}.attention.self.query.weight", 3 * D * D), wo=st.s16(f"{p}.attention.output.dense.weight"),
            w1=st.s16(f"{p}.intermediate.dense.weight"), w2=st.s16(f"{p}.output.dense.weight"),
            bqkv=st.flat[st.offsets[f"{p}.attention.self.query.bias"][0]:][: 3 * D], bo=st.p32(f"{p}.attention.output.dense.bias"),
            b1=st.p32(f"{p}.intermediate.dense.bias"), b2=st.p32(f"{p}.output.dense.bias"),
            g1=st.p32(f"{p}.attention.output.LayerNorm.weight"), be1=st.p32(f"{p}.attention.output.LayerNorm.bias"),
            g2=st.p32(f"{p}.output.LayerNorm.weight"), be2=st.p32(f"{p}.output.LayerNorm.bias"),
            qkv=qkv[i], probs=probs[i], ctx=act[0, i], ao=act[1, i], x1=act[2, i], z=zz[0, i], hg=zz[1, i], f=act[3, i], xout=act[4, i],
            m1=stat[0, i], r1=stat[1, i], m2=stat[2, i], r2=stat[3, i],
            site_probs=model._sites[f"enc.{i}.attn.probs"], site_ao=model._sites[f"enc.{i}.attn.out"], site_fo=model._sites[f"enc.{i}.ff.out"])
        recs.append(rec)
        tape[p] = dict(x=xin, qkv=rec["qkv"], ctx=rec["ctx"], probs=rec["probs"], ao=rec["ao"], x1=rec["x1"], m1=rec["m1"], r1=rec["r1"], z=rec["z"],
                       hg=rec["hg"], f=rec["f"], m2=rec["m2"], r2=rec["r2"], dpr=dpr, dao=dao, dfo=dfo)
        xin = rec["xout"]
    on = training and (model.drop_p > 0.0 or model.attn_drop_p > 0.0)
    ops.enc_fwd(x, recs, B, S, model.ln_eps, model._drop_word if on else None, model.drop_p if training else 0.0,
                model.attn_drop_p if training else 0.0)
    return xin


def _encoder_layers_backward_fused(model: TransformerLightningModule, st: _ParamStore, tape: dict, dh: torch.Tensor, B: int, S: int,
                                   defer: Optional[list]) -> torch.Tensor:
    """The backward of all encoder layers in one launch (ops.enc_bwd, csrc/enc_fused.hip k_enc_bwd), then the parameter gradients it leaves
    to the weight-gradient launches — the same calls, in the same order, as the per-layer path below makes.  Returns the gradient of the
    first layer's input."""
    D, H, I, Lc = model.dim, model.heads, model.inter, model.layers
    R = B * S
    dev = dh.device
    t0 = tape["encoder.encoder.layer.0"]
    hidden_drop = t0["dfo"] is not None or t0["dao"] is not None
    act = torch.empty((6 if hidden_drop else 4, Lc, R, D), dtype=BF16, device=dev)      # ds2, dx1, ds1, dx (, df, dao)
    dz = torch.empty((Lc, R, I), dtype=BF16, device=dev)
    dqkv = torch.empty((Lc, R, 3 * D), dtype=BF16, device=dev)
    part = torch.empty((Lc, 2, B, 2 * D), dtype=torch.float32, device=dev)              # [layer][LayerNorm 1 | 2][sequence][dy*xhat | dy]
    recs = []
    for i in range(Lc):
        p = f"encoder.encoder.layer.{i}"
        t = tape[p]
        recs.append(dict(
            w2t=st.t16(f"{p}.output.dense.weight"), w1t=st.t16(f"{p}.intermediate.dense.weight"),
            wot=st.t16(f"{p}.attention.output.dense.weight"), wqkvt=st.t16(f"{p}.qkv"),
            g1=st.p32(f"{p}.attention.output.LayerNorm.weight"), g2=st.p32(f"{p}.output.LayerNorm.weight"),
            f=t["f"], x1=t["x1"], ao=t["ao"], xin=t["x"], z=t["z"], qkv=t["qkv"], probs=t["probs"], m1=t["m1"], r1=t["r1"], m2=t["m2"], r2=t["r2"],
            ds2=act[0, i], dx1=act[1, i], ds1=act[2, i], dx=act[3, i], df=act[4, i] if hidden_drop else act[0, i],
            dao=act[5, i] if hidden_drop else act[2, i], dz=dz[i], dqkv=dqkv[i], part1=part[i, 0], part2=part[i, 1],
            site_probs=model._sites[f"enc.{i}.attn.probs"], site_ao=model._sites[f"enc.{i}.attn.out"], site_fo=model._sites[f"enc.{i}.ff.out"]))
    on = hidden_drop or t0["dpr"] is not None
    ops.enc_bwd(dh, recs, B, S, model._drop_word if on else None, model.drop_p if hidden_drop else 0.0,
                model.attn_drop_p if t0["dpr"] is not None else 0.0)

    def ln_grads(rows, name):
        gw, gb = st.g32(f"{name}.weight"), st.g32(f"{name}.bias")
        if defer is None:
            ops.colsum_rows(rows, B, 2 * D, gw, D, gb, D)
        else:
            defer.append((lambda: ops.colsum_rows(rows, B, 2 * D, gw, D, gb, D), rows))

    for i in reversed(range(Lc)):
        p = f"encoder.encoder.layer.{i}"
        t, q = tape[p], recs[i]
        ln_grads(q["part2"], f"{p}.output.LayerNorm")
        _lin_wgrad(model, t["hg"], q["df"], st.g32(f"{p}.output.dense.weight"), st.g32(f"{p}.output.dense.bias"), R, I, D, I, D)
        # (the intermediate bias's gradient: the column sums of dz, taken by the weight-gradient launch)
        _lin_wgrad(model, t["x1"], q["dz"], st.g32(f"{p}.intermediate.dense.weight"), st.g32(f"{p}.intermediate.dense.bias"), R, D, I, D, I)
        ln_grads(q["part1"], f"{p}.attention.output.LayerNorm")
        _lin_wgrad(model, t["ctx"], q["dao"], st.g32(f"{p}.attention.output.dense.weight"), st.g32(f"{p}.attention.output.dense.bias"), R, D, D, D, D)
        gq = st.grad[st.offsets[f"{p}.attention.self.query.weight"][0]:][: 3 * D * D]
        gqb = st.grad[st.offsets[f"{p}.attention.self.query.bias"][0]:][: 3 * D]
        _lin_wgrad(model, t["x"], q["dqkv"], gq, gqb, R, D, 3 * D, D, 3 * D)
        _flush_deferred(model)
        if getattr(model, "_wg_group", None) is None:
            _ready(model, st, f"{p}.attention.self.query.weight")
        elif WG_GROUP_LAYERS > 0 and (Lc - i) % WG_GROUP_LAYERS == 0 and i > 0:
            _flush_lin_wgrads(model)
            model._wg_group = []
            _ready(model, st, f"{p}.attention.self.query.weight")
    return recs[0]["dx"]


def _encoder_backward(model: TransformerLightningModule, st: _ParamStore, tape: dict, dh: torch.Tensor, B: int, T: int) -> torch.Tensor:
    D, S, H, I = model.dim, T + 1, model.heads, model.inter
    R = B * S
    use_tr = model.use_tr
    dx = dh
    # parameter-gradient reductions of the row passes (LayerNorm gamma / beta, the intermediate bias): nothing downstream waits for them, so
    # they leave the chain of dependent launches and run on the side stream, one hand-over per layer (_flush_deferred, from _ready)
    defer = _defer_list(model)

    fused = ops.ENC_BWD_FUSED and ops.enc_fused_ok(D, H, I, S) and model.layers <= 8
    if fused:
        dx = _encoder_layers_backward_fused(model, st, tape, dh, B, S, defer)
    for i in reversed(range(0 if fused else model.layers)):
        p = f"encoder.encoder.layer.{i}"
        t = tape[p]
        ds2 = ops.add_ln_bwd(dx, t["f"], t["x1"], st.p32(f"{p}.output.LayerNorm.weight"), t["m2"], t["r2"],
                             st.g32(f"{p}.output.LayerNorm.weight"), st.g32(f"{p}.output.LayerNorm.bias"), defer=defer)
        # ds2 is the gradient of (dropout(f) + x1): the dense layer sees it through the regenerated mask, the skip path as it is
        df = ds2 if t["dfo"] is None else ops.scale_bf16(ds2, 1.0, drop=t["dfo"])
        _lin_wgrad(model, t["hg"], df, st.g32(f"{p}.output.dense.weight"), st.g32(f"{p}.output.dense.bias"), R, I, D, I, D)
        dhg = ops.linear_dgrad(df, st.t16(f"{p}.output.dense.weight"), rows=R, N=D, K=I, dy_pitch=D)
        dz = ops.bias_act_bwd(dhg, t["z"], st.g32(f"{p}.intermediate.dense.bias"), R=R, N=I, n_valid=I, ld=I, defer=defer)
        _lin_wgrad(model, t["x1"], dz, st.g32(f"{p}.intermediate.dense.weight"), None, R, D, I, D, I)
        # (not in place: the side stream may still be reading ds2 / ds1 for the weight gradients)
        dx1 = ops.linear_dgrad(dz, st.t16(f"{p}.intermediate.dense.weight"), rows=R, N=I, K=D, dy_pitch=I, addend=ds2)
        ds1 = ops.add_ln_bwd(dx1, t["ao"], t["x"], st.p32(f"{p}.attention.output.LayerNorm.weight"), t["m1"], t["r1"],
                             st.g32(f"{p}.attention.output.LayerNorm.weight"), st.g32(f"{p}.attention.output.LayerNorm.bias"), defer=defer)
        dao_ = ds1 if t["dao"] is None else ops.scale_bf16(ds1, 1.0, drop=t["dao"])
        _lin_wgrad(model, t["ctx"], dao_, st.g32(f"{p}.attention.output.dense.weight"), st.g32(f"{p}.attention.output.dense.bias"), R, D, D, D, D)
        dctx = ops.linear_dgrad(dao_, st.t16(f"{p}.attention.output.dense.weight"), rows=R, N=D, K=D, dy_pitch=D)
        qkv = t["qkv"]
        dqkv = torch.empty_like(qkv)
        ops.mha_bwd(dctx, qkv, 3 * D, qkv[:, D:], qkv[:, 2 * D:], 3 * D, t["probs"], B=B, H=H, Lq=S, Lk=S, dq=dqkv, dq_pitch=3 * D,
                    dk=dqkv[:, D:], dv=dqkv[:, 2 * D:], dkv_pitch=3 * D, drop=t["dpr"])
        gq = st.grad[st.offsets[f"{p}.attention.self.query.weight"][0] :][: 3 * D * D]
        gqb = st.grad[st.offsets[f"{p}.attention.self.query.bias"][0] :][: 3 * D]
        _lin_wgrad(model, t["x"], dqkv, gq, gqb, R, D, 3 * D, D, 3 * D)
        dx = ops.linear_dgrad(dqkv, st.t16(f"{p}.qkv"), rows=R, N=3 * D, K=D, dy_pitch=3 * D, addend=ds1)
        _flush_deferred(model)
        if getattr(model, "_wg_group", None) is None:
            _ready(model, st, f"{p}.attention.self.query.weight")
        elif WG_GROUP_LAYERS > 0 and (model.layers - i) % WG_GROUP_LAYERS == 0 and i > 0:
            # the weight gradients collected so far as one launch on the side stream NOW: the encoder's backward is a chain of small
            # launches that leaves most of the chip idle, while a single launch at its end competes with the trunk's backward
            _flush_lin_wgrads(model)
            model._wg_group = []
            _ready(model, st, f"{p}.attention.self.query.weight")
    if getattr(model, "_wg_group", None) is not None:
        # the remaining linear weight gradients of the encoder (and the heads) in one launch; their flat-buffer range is final from here
        _flush_lin_wgrads(model)
        _ready(model, st, "encoder.encoder.layer.0.attention.self.query.weight")
    te = tape["emb"]
    if te["d_out"] is not None:
        dx = ops.scale_bf16(dx, 1.0, drop=te["d_out"])
    ds0 = ops.add_ln_bwd(dx, te["sum"], None, st.p32("encoder.embeddings.LayerNorm.weight"), te["mean"], te["rstd"],
                         st.g32("encoder.embeddings.LayerNorm.weight"), st.g32("encoder.embeddings.LayerNorm.bias"), defer=defer)
    dfeats = ops.embed_bwd_scatter(ds0, st.g32("cls_token"), st.g32("encoder.embeddings.position_embeddings.weight"),
                                   st.g32("encoder.embeddings.token_type_embeddings.weight"), B, S, D, drop_in=te["d_in"])
    _ready(model, st, "cls_token")
    return dfeats


# ----------------------------------------------------------------------------------------------------
# `type: x-transformers` encoder (lightning.py:93-105,157-158): pre-norm residual blocks
#   x += to_out(attention(rotary(to_q|to_k|to_v(RMSNorm(x)))));   x += W2 . dropout(value * gelu(gate)) (+ biases), [value|gate] = W1 . RMSNorm(x)
# with whole blocks skipped at random in training (layer_dropout).  Rows are `dim_p` wide (513 -> 576), pads zero.
# ----------------------------------------------------------------------------------------------------
def _xt_skips(model) -> set:
    """Indices n of `encoder.layers.{n}` skipped this step: AttentionLayers.forward draws python's random() once per layer, in
    order, and skips the layer when it falls below layer_dropout (training only)."""
    if not model.training or model.layer_drop_p <= 0.0:
        return set()
    if model.layer_skip_override is not None:
        return set(model.layer_skip_override)
    return {n for n in range(2 * model.layers) if model._layer_rng.random() < model.layer_drop_p}


def _xt_encoder_forward(model, st: _ParamStore, tape: dict, feats: torch.Tensor, wm: Optional[torch.Tensor], B: int, T: int) -> torch.Tensor:
    D, Dp, S, H, E = model.dim, model.dim_p, T + 1, model.heads, model.attn_inner
    I, Ip, Up = model.inter, model.inter_p, model.glu_p
    R = B * S
    d_in = model._d("emb.in", "emb")
    x = ops.xt_embed_fwd(feats, wm, st.p32("cls_token"), B, S, feats.shape[-1], D, Dp, drop=d_in)
    key = (S, str(feats.device))
    if key not in model._rot_tab:
        model._rot_tab[key] = ops.rotary_table(S, feats.device)
    tab = model._rot_tab[key]
    skip = _xt_skips(model)
    nrot = (3 if model.rotate_value else 2) * H
    tape["xt"] = dict(d_in=d_in, tab=tab, nrot=nrot, F=feats.shape[-1])
    for i in range(model.layers):
        a, f = f"encoder.layers.{2 * i}", f"encoder.layers.{2 * i + 1}"
        if 2 * i not in skip:
            h, inv = ops.rmsnorm_fwd(x, st.p32(f"{a}.0.0.g"), D, model.rms_eps)
            qkv, _ = ops.linear_fwd(h, st.s16(f"{a}.1.to_q.weight", 3 * E * Dp), None, rows=R, K=Dp, N=3 * E, x_pitch=Dp)
            ops.rotary_(qkv, tab, S, nrot, 1)
            dpr = model._d(f"enc.{i}.attn.probs", "attn")
            ctx, probs = ops.mha_fwd(qkv, 3 * E, qkv[:, E:], qkv[:, 2 * E:], 3 * E, B=B, H=H, Lq=S, Lk=S, drop=dpr)
            x1, _ = ops.linear_fwd(ctx, st.s16(f"{a}.1.to_out.weight"), None, rows=R, K=E, N=Dp, x_pitch=E, addend=x)
            tape[a] = dict(x=x, h=h, inv=inv, qkv=qkv, ctx=ctx, probs=probs, dpr=dpr)
            x = x1
        if 2 * i + 1 not in skip:
            h, inv = ops.rmsnorm_fwd(x, st.p32(f"{f}.0.0.g"), D, model.rms_eps)
            u, _ = ops.linear_fwd(h, st.s16(f"{f}.1.ff.0.proj.weight"), st.p32(f"{f}.1.ff.0.proj.bias"), rows=R, K=Dp, N=Up, x_pitch=Dp)
            dff = model._d(f"enc.{i}.ff.hidden")
            y = ops.geglu_fwd(u, I, Ip, drop=dff)
            x1, _ = ops.linear_fwd(y, st.s16(f"{f}.1.ff.3.weight"), st.p32(f"{f}.1.ff.3.bias"), rows=R, K=Ip, N=Dp, x_pitch=Ip, addend=x)
            tape[f] = dict(x=x, h=h, inv=inv, u=u, y=y, dff=dff)
            x = x1
    if model.final_norm:
        h, inv = ops.rmsnorm_fwd(x, st.p32("encoder.final_norm.g"), D, model.rms_eps)
        tape["xt"]["final"] = dict(x=x, inv=inv)
        x = h
    return x


def _xt_encoder_backward(model, st: _ParamStore, tape: dict, dh: torch.Tensor, B: int, T: int) -> torch.Tensor:
    D, Dp, S, H, E = model.dim, model.dim_p, T + 1, model.heads, model.attn_inner
    I, Ip, Up = model.inter, model.inter_p, model.glu_p
    R = B * S
    tx = tape["xt"]
    dx = dh
    if "final" in tx:
        dx = ops.rmsnorm_bwd(dx, tx["final"]["x"], st.p32("encoder.final_norm.g"), tx["final"]["inv"], st.g32("encoder.final_norm.g"), D)
    for i in reversed(range(model.layers)):
        a, f = f"encoder.layers.{2 * i}", f"encoder.layers.{2 * i + 1}"
        if f in tape:
            t = tape[f]
            _lin_wgrad(model, t["y"], dx, st.g32(f"{f}.1.ff.3.weight"), st.g32(f"{f}.1.ff.3.bias"), R, Ip, Dp, Ip, Dp)
            dy = ops.linear_dgrad(dx, st.t16(f"{f}.1.ff.3.weight"), rows=R, N=Dp, K=Ip, dy_pitch=Dp)
            du = ops.geglu_bwd(dy, t["u"], I, drop=t["dff"])
            _lin_wgrad(model, t["h"], du, st.g32(f"{f}.1.ff.0.proj.weight"), st.g32(f"{f}.1.ff.0.proj.bias"), R, Dp, Up, Dp, Up)
            dhn = ops.linear_dgrad(du, st.t16(f"{f}.1.ff.0.proj.weight"), rows=R, N=Up, K=Dp, dy_pitch=Up)
            dx = ops.rmsnorm_bwd(dhn, t["x"], st.p32(f"{f}.0.0.g"), t["inv"], st.g32(f"{f}.0.0.g"), D, addend=dx)
        if a in tape:
            t = tape[a]
            _lin_wgrad(model, t["ctx"], dx, st.g32(f"{a}.1.to_out.weight"), None, R, E, Dp, E, Dp)
            dctx = ops.linear_dgrad(dx, st.t16(f"{a}.1.to_out.weight"), rows=R, N=Dp, K=E, dy_pitch=Dp)
            qkv = t["qkv"]
            dqkv = torch.empty_like(qkv)
            ops.mha_bwd(dctx, qkv, 3 * E, qkv[:, E:], qkv[:, 2 * E:], 3 * E, t["probs"], B=B, H=H, Lq=S, Lk=S, dq=dqkv, dq_pitch=3 * E,
                        dk=dqkv[:, E:], dv=dqkv[:, 2 * E:], dkv_pitch=3 * E, drop=t["dpr"])
            ops.rotary_(dqkv, tx["tab"], S, tx["nrot"], -1)           # the rotation is orthogonal: its transpose is the inverse rotation
            gq = st.grad[st.offsets[f"{a}.1.to_q.weight"][0]:][: 3 * E * Dp]
            _lin_wgrad(model, t["h"], dqkv, gq, None, R, Dp, 3 * E, Dp, 3 * E)
            dhn = ops.linear_dgrad(dqkv, st.t16(f"{a}.1.qkv"), rows=R, N=3 * E, K=Dp, dy_pitch=3 * E)
            dx = ops.rmsnorm_bwd(dhn, t["x"], st.p32(f"{a}.0.0.g"), t["inv"], st.g32(f"{a}.0.0.g"), D, addend=dx)
        _ready(model, st, f"{a}.1.to_q.weight")
    dfeats = ops.xt_embed_bwd(dx, st.g32("cls_token"), B, S, tx["F"], D, drop=tx["d_in"])
    _ready(model, st, "cls_token")
    return dfeats


class _LrwFunction(torch.autograd.Function):
    """One autograd node for the whole model: forward records a tape, backward replays it by hand and writes the
    parameter gradients straight into the flat gradient buffer (the returned input gradients are all None)."""

    @staticmethod
    def forward(ctx, _anchor, model: TransformerLightningModule, st: _ParamStore, videos, audio_tokens, labels, need_grad: bool, word_mask=None):
        training = model.training
        B, _, T, H, W = videos.shape
        D, S = model.dim_p, T + 1            # storage width of the encoder rows (= dim unless the word boundary makes it 513 -> 576)
        A, G, V = model.audio_alignment, model.vq_groups, model.audio_vocab_size
        if not st.shadow_fresh:          # engine.TrainStep keeps the bf16 shadows current from its optimiser kernel
            st.refresh_shadows()
        if training and (model.drop_p > 0.0 or model.attn_drop_p > 0.0 or model.emb_drop_p > 0.0):
            model._advance_dropout(videos.device)
        tape: dict[str, Any] = {}
        feats = _frontend_forward(model, st, tape, videos, training)
        model._side.join()            # the previous step's optimiser may still be updating the encoder / heads on the side stream (engine.TrainStep)
        if model.encoder_type == "x-transformers":
            h = _xt_encoder_forward(model, st, tape, feats, word_mask, B, T)
        else:
            h = _encoder_forward(model, st, tape, feats, B, T)          # [B*S, D] bf16
        # word head: rows s = 0
        C = model.word_labels
        hard = labels.dtype in (torch.int64, torch.int32)
        lab_idx = labels.long() if hard else None
        lab_prob = None if hard else labels.float().contiguous()
        # (round 6) inside TrainStep the word head (classifier, its loss, the metric — and in the backward its loss gradient and data gradient) is a
        # branch of its own between the two encoder launches, of 1-8 workgroups per launch: it runs on the side stream beside the audio head
        side_heads = bool(getattr(model, "_metrics_on_side", False) and model._side.enabled and HEADS_CONCURRENT and need_grad
                          and ops.WGRAD_GROUP and model.encoder_type == "huggingface")       # (the heads' weight gradients then follow on the side stream too)
        hb: dict = {}

        def word_head():
            hb["logits_c"], _ = ops.linear_fwd(h, st.s16("category_classifier.weight"), st.p32("category_classifier.bias"), rows=B, K=D, N=C,
                                               x_pitch=D, out_f32=True, seq=(S, 0, 1))
            hb["loss_c"], hb["lse_c"] = ops.ce_fwd(hb["logits_c"], C, lab_idx, lab_prob, B, C, model.label_smoothing)
            if side_heads:
                hb["acc"] = ops.topk_acc(hb["logits_c"], lab_idx, lab_prob)

        if side_heads:
            model._side.run(word_head, h)
            model._side.flush()
        else:
            word_head()
        logits_c, loss_c, lse_c = hb["logits_c"], hb["loss_c"], hb["lse_c"]
        # audio head: rows s = 1..T, logits [B*T, A*G*V] == [B*T*A*G, V]
        NA = A * G * V
        tok = audio_tokens.reshape(-1)
        fused_head = model.fused_audio_head and ops.linear_ce_ok(B * T, D, A * G, V)
        logits_a = None
        if not fused_head or model.keep_audio_logits:     # (keep_audio_logits: the logits tensor for inspection / the parity tests; the loss below does not read it)
            logits_a, _ = ops.linear_fwd(h, st.s16("audio_projection.weight"), st.p32("audio_projection.bias"), rows=B * T, K=D, N=NA,
                                         x_pitch=D, seq=(S, 1, T))
        if fused_head:
            # projection + per-frame cross-entropy in one contraction, logits never stored (csrc/audio_head.hip)
            loss_a, lse_a = ops.linear_ce_fwd(h, st.s16("audio_projection.weight"), st.p32("audio_projection.bias"), tok, B * T, D, A * G, V, seq=(S, 1, T))
        else:
            loss_a, lse_a = ops.ce_fwd(logits_a, V, tok, None, B * T * A * G, V, 0.0)
        if side_heads:
            acc = hb["acc"]
        elif getattr(model, "_metrics_on_side", False) and model._side.enabled and METRICS_ON_SIDE:
            # (round 6) inside TrainStep the metric is read when the step is over: its two launches leave the main stream's chain (the
            # side stream is joined at the end of the backward)
            box: dict = {}
            model._side.run(lambda: box.__setitem__("acc", ops.topk_acc(logits_c, lab_idx, lab_prob)), logits_c)
            model._side.flush()
            acc = box["acc"]
        else:
            acc = ops.topk_acc(logits_c, lab_idx, lab_prob)
        model._last = dict(logits_category=logits_c, logits_audio=logits_a, feats=feats, hidden=h)
        if need_grad:
            tape["head"] = dict(h=h, logits_c=logits_c, lse_c=lse_c, lab_idx=lab_idx, lab_prob=lab_prob, logits_a=None if fused_head else logits_a,
                                lse_a=lse_a, tok=tok, dims=(B, T, D, S, A, G, V, C), side_heads=side_heads)
            ctx.tape = tape
            ctx.model = model
            ctx.st = st
        ctx.mark_non_differentiable(acc)
        return loss_c, loss_a, acc

    @staticmethod
    def backward(ctx, g_cat, g_audio, _g_acc):
        model, st, tape = ctx.model, ctx.st, ctx.tape
        th = tape["head"]
        B, T, D, S, A, G, V, C = th["dims"]
        dev = th["h"].device
        use_tr = model.use_tr
        if not getattr(model, "accumulate_grads", False):
            st.zero_grad()
        st.rebind_grads()
        g_cat = (g_cat if g_cat is not None else torch.zeros((), device=dev)).float().contiguous()
        g_audio = (g_audio if g_audio is not None else torch.zeros((), device=dev)).float().contiguous()
        NA = A * G * V
        Cp = (C + 63) // 64 * 64
        dlc = torch.empty((B, Cp), dtype=BF16, device=dev)          # (svsr_ce_bwd writes the pad columns C .. Cp - 1 as zeros)
        dh = torch.empty((B * S, D), dtype=BF16, device=dev)
        side_heads = bool(th.get("side_heads")) and model._side.enabled
        if th.get("side_heads") and not side_heads:
            model._side.join()          # (the forward's half of the branch ran there)

        def word_head_bwd():
            ops.ce_bwd(th["logits_c"], C, th["lab_idx"], th["lab_prob"], B, C, model.label_smoothing, th["lse_c"], g_cat, dlc, Cp)
            if side_heads:      # (rows s = 0 of dh; the audio head's data gradient below writes rows 1..T)
                ops.linear_dgrad(dlc, st.t16("category_classifier.weight"), rows=B, N=C, K=D, dy_pitch=Cp, out=dh, seq=(S, 0, 1))

        if side_heads:
            model._side.run(word_head_bwd, dlc, dh, g_cat)
            model._side.flush()
        else:
            word_head_bwd()
        dla = torch.empty((B * T, NA), dtype=BF16, device=dev)
        if th["logits_a"] is None:        # fused head: the logits are recomputed, dla = g / (B T A G) * (softmax - onehot)
            ops.linear_ce_bwd(th["h"], st.s16("audio_projection.weight"), st.p32("audio_projection.bias"), th["tok"], B * T, D, A * G, V, th["lse_a"],
                              g_audio, dla, seq=(S, 1, T))
        else:
            ops.ce_bwd(th["logits_a"], V, th["tok"], None, B * T * A * G, V, 0.0, th["lse_a"], g_audio, dla, V)
        h = th["h"]
        grouped = ops.WGRAD_GROUP and model.encoder_type == "huggingface"
        model._wg_group = [] if grouped else None
        heads = (dict(x=h, dy=dla, dw=st.g32("audio_projection.weight"), rows=B * T, K=D, N=NA, x_pitch=D, dy_pitch=NA, seq=(S, 1, T), db=st.g32("audio_projection.bias")),
                 dict(x=h, dy=dlc, dw=st.g32("category_classifier.weight"), rows=B, K=D, N=C, x_pitch=D, dy_pitch=Cp, seq=(S, 0, 1), db=st.g32("category_classifier.bias")))
        for q in heads:
            if grouped:
                model._wg_group.append(q)
            else:
                ops.linear_wgrad(q["x"], q["dy"], q["dw"], rows=q["rows"], K=q["K"], N=q["N"], x_pitch=q["x_pitch"], dy_pitch=q["dy_pitch"], seq=q["seq"],
                                 use_tr=use_tr, db=q["db"])
        ops.linear_dgrad(dla, st.t16("audio_projection.weight"), rows=B * T, N=NA, K=D, dy_pitch=NA, out=dh, seq=(S, 1, T))
        if side_heads:
            model._side.join()
        else:
            ops.linear_dgrad(dlc, st.t16("category_classifier.weight"), rows=B, N=C, K=D, dy_pitch=Cp, out=dh, seq=(S, 0, 1))
        if not grouped:
            _ready(model, st, "audio_projection.weight")
        if model.encoder_type == "x-transformers":
            dfeats = _xt_encoder_backward(model, st, tape, dh, B, T)
        else:
            dfeats = _encoder_backward(model, st, tape, dh, B, T)
        _frontend_backward(model, st, tape, dfeats)
        ctx.tape = None
        return None, None, None, None, None, None, None, None
