"""Drop-in module for the reference's LRS model ``E2E`` (LRS/video/espnet/nets/pytorch_backend/e2e_asr_transformer.py:43-227), HIP-native.

Same constructor ``E2E(odim, args, ignore_id=-1)`` reading the ``model.visual_backbone`` keys of ``config/lrs3.yaml:14-39``, same
``forward(x, lengths, audios, label) -> (loss, loss_ctc, loss_att, loss_audio, acc)`` and the same state-dict names
(``encoder.frontend.{frontend3D,trunk}``, ``encoder.embed.0``, ``encoder.encoders.N.{self_attn,feed_forward,feed_forward_macaron,
conv_module,norm_*}``, ``encoder.after_norm``, ``decoder.*``, ``ctc.ctc_lo``, ``audio_classifier``).  Deviations, both from
SURVEY §8(b): the ``audios`` slot takes pre-computed audio tokens int64 [B, >=A*T, G] (the frozen wav2vec quantiser's weights are
not available offline), and ``acc`` is a 0-d device tensor instead of a Python float (no host sync inside the step).

As in ``model.py`` this file is orchestration only: one flat fp32 parameter buffer (+ gradient buffer + bf16 shadows), a hand
written tape, and a single autograd node for the whole model; every computation is a launch into libsyncvsr_hip.so.
"""
from __future__ import annotations

import math
import os
from typing import Any, Optional

import torch
import torch.nn as nn

from . import ops
from .config import Config
from .lrs_init import LRS_ODIM, lrs_audio_dims, lrs_buffer_specs, lrs_init_state_dict, lrs_param_specs
from . import model as _model_mod
from .model import BF16, _Holder, _SideStream, _ParamStore, _attach, _defer_list, _frontend_backward, _frontend_forward, _get, _bn_stats, _ready

LN_EPS = 1e-12          # transformer/layer_norm.py:19


def _sinusoid(positions: torch.Tensor, d_model: int) -> torch.Tensor:
    """transformer/embedding.py:54-76 / :180-199 — sin on even, cos on odd columns, computed in fp32 like the reference."""
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    ang = positions.float().unsqueeze(1) * div
    pe = torch.empty(positions.numel(), d_model)
    pe[:, 0::2] = torch.sin(ang)
    pe[:, 1::2] = torch.cos(ang)
    return pe


class LrsTargets:
    """Device-side target tensors derived from ``label`` once per batch (``E2E.prepare_targets``): CTC labels [B, L] padded
    with -1, decoder input/output [B, L+1] (add_sos_eos.py:10-31)."""

    def __init__(self, labels: torch.Tensor, ys_in: torch.Tensor, ys_out: torch.Tensor):
        self.labels, self.ys_in, self.ys_out = labels, ys_in, ys_out


class _EncoderFacade(_Holder):
    """`E2E.encoder` of the reference as a callable (transformer/encoder.py:257-289): model.encoder(xs, masks) -> (xs, masks)."""

    def forward(self, xs: torch.Tensor, masks: Optional[torch.Tensor] = None, extract_resnet_feats: bool = False):
        owner = self._owner()
        lengths = None if masks is None else masks.reshape(masks.size(0), -1).sum(-1)
        if extract_resnet_feats:
            return owner.encode(xs, lengths, resnet_feats=True)
        return owner.encode(xs, lengths), masks

    def forward_one_step(self, xs: torch.Tensor, masks: Optional[torch.Tensor] = None, cache=None):
        """encoder.py:291-318 for cache=None: (xs, masks, per-layer outputs).  Incremental re-use of a cache is not implemented —
        the reference's own inference path never passes one (lightning.py:100-101,114-118), and with the shipped rel_pos Conformer layers
        the reference's method itself raises on its first call (it hands the layers' (x, pos_emb) tuple to after_norm, encoder.py:311-317;
        checked against the imported reference): there is no reference behaviour for a cache to reproduce."""
        if cache is not None:
            raise NotImplementedError("Encoder.forward_one_step with a cache (streaming) is not implemented")
        owner = self._owner()
        lengths = None if masks is None else masks.reshape(masks.size(0), -1).sum(-1)
        outs: list = []
        h = owner.encode(xs, lengths, layer_outs=outs)
        return h, masks, outs


class _DecoderFacade(_Holder):
    """`E2E.decoder` as the beam-search scorer of the reference (transformer/decoder.py:153-220)."""

    def _scorer(self):
        from .lrs_infer import DecoderScorer

        sc = self.__dict__.get("_sc")
        if sc is None or sc.model is not self._owner():
            sc = self.__dict__["_sc"] = DecoderScorer(self._owner())         # kept: it owns the per-clip source key / value projections
        return sc

    def forward_one_step(self, tgt, tgt_mask, memory, memory_mask=None, cache=None):
        return self._scorer().forward_one_step(tgt, tgt_mask, memory, memory_mask, cache)

    def score(self, ys, state, x):
        return self._scorer().score(ys, state, x)

    def batch_score(self, ys, states, xs):
        return self._scorer().batch_score(ys, states, xs)

    def init_state(self, x):
        return None

    def batch_init_state(self, x):
        return None

    def select_state(self, state, i, new_id=None):
        return self._scorer().select_state(state, i, new_id)

    def select_states(self, states, prev, tok):
        return self._scorer().select_states(states, prev, tok)


class _CtcFacade(_Holder):
    """`E2E.ctc` inference helpers (ctc.py:154-181) on hs_pad [B, T, adim]."""

    def log_softmax(self, hs_pad: torch.Tensor) -> torch.Tensor:
        from .lrs_infer import CTCPrefixScorer

        sc = CTCPrefixScorer(self._owner(), self._owner().eos)
        return torch.stack([sc.ctc_log_softmax(h) for h in hs_pad])

    def softmax(self, hs_pad: torch.Tensor) -> torch.Tensor:
        return self.log_softmax(hs_pad).exp()

    def argmax(self, hs_pad: torch.Tensor) -> torch.Tensor:
        return self.log_softmax(hs_pad).argmax(dim=-1)


class E2E(nn.Module):
    def __init__(self, odim: int = LRS_ODIM, args: Any = None, ignore_id: int = -1, seed: Optional[int] = None):
        super().__init__()
        from .lrs_init import default_lrs_args

        if args is None:
            args = default_lrs_args()
        if not isinstance(args, Config):
            args = Config(vars(args) if hasattr(args, "__dict__") and not isinstance(args, dict) else args)
        self.args = args
        self.odim, self.ignore_id = int(odim), int(ignore_id)
        self.sos = self.eos = self.odim - 1
        self.adim, self.ddim = int(args.adim), int(args.ddim)
        self.aheads, self.dheads = int(args.aheads), int(args.dheads)
        self.eunits, self.dunits = int(args.eunits), int(args.dunits)
        self.elayers, self.dlayers = int(args.elayers), int(args.dlayers)
        self.kernel = int(args.cnn_module_kernel)
        self.mtlalpha, self.lsm_weight = float(args.mtlalpha), float(args.lsm_weight)
        self.length_norm = bool(args.transformer_length_normalized_loss)
        self.audio_weight = float(args.audio_weight)
        unsupported = []
        if args.transformer_input_layer not in ("conv3d", "conv3d-lrw"):
            unsupported.append("transformer_input_layer must be conv3d or conv3d-lrw (the visual front-ends, encoder.py:130-139)")
        if args.transformer_encoder_attn_layer_type != "rel_mha" or args.get("rel_pos_type", "latest") != "latest":
            unsupported.append("encoder attention must be rel_mha with rel_pos_type latest")
        if not args.macaron_style or not args.use_cnn_module:
            unsupported.append("macaron_style and use_cnn_module must be on")
        if args.get("relu_type", "swish") != "swish":
            unsupported.append("relu_type must be swish")
        if args.get("zero_triu", False):
            unsupported.append("zero_triu is not supported")
        if self.adim % 64 or self.ddim % 64 or self.adim // self.aheads != 64 or self.ddim // self.dheads != 64:
            unsupported.append("attention heads must be 64 wide")
        if not (0.0 <= self.mtlalpha < 1.0):
            unsupported.append("mtlalpha must be in [0, 1): with mtlalpha = 1 the reference builds no decoder (e2e_asr_transformer.py:96-109) "
                               "and its own forward then fails at `self.decoder(...)` (:214)")
        if self.kernel % 2 == 0 or self.kernel > 31:
            unsupported.append("cnn_module_kernel must be odd and <= 31")
        if unsupported:
            raise NotImplementedError("; ".join(unsupported))
        # nn.Dropout sites (all p = dropout_rate except the attention probabilities: transformer_attn_dropout_rate; e2e:54-55)
        from .dropout import lrs_sites

        self.drop_p = float(args.dropout_rate)
        attn_p = args.transformer_attn_dropout_rate
        self.attn_drop_p = self.drop_p if attn_p is None else float(attn_p)
        self._sites = lrs_sites(self.elayers, self.dlayers)
        self.dropout_seed = 0                     # base seed; the device-side word advances by one per training forward
        self._drop_word: Optional[torch.Tensor] = None
        self.audio_alignment, self.vq_groups, self.audio_vocab_size = lrs_audio_dims(args)
        self.codec = "vq" if self.audio_alignment == 4 else "wav2vec2"

        self._specs = lrs_param_specs(args, self.odim)
        self._bspecs = lrs_buffer_specs(args, self.odim)
        sd = lrs_init_state_dict(args, self.odim, seed=0 if seed is None else seed)
        for name, shape, kind in self._specs:
            t = sd[name]
            if kind == "conv" and len(shape) == 4:
                t = t.contiguous(memory_format=torch.channels_last)
            _attach(self, name, t, True)
        for name, shape, kind in self._bspecs:
            _attach(self, name, sd[name], False)
        self._store: Optional[_ParamStore] = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.mark_params_dirty())
        self.register_load_state_dict_pre_hook(lambda module, *a, **k: module._side.join())       # (an AdamW range may still be writing the flat buffer on the side stream)
        from .lrs_init import lrs_frontend_names

        self.stem_name, self.trunk_name = lrs_frontend_names(args)
        if args.transformer_input_layer == "conv3d-lrw":         # the word-level model's front-end: GELU stem, ReLU ResNet18 (encoder.py:132-139)
            self.stem_act, self.trunk_act = ops.ACT_GELU, ops.ACT_RELU
        else:
            self.stem_act = self.trunk_act = ops.ACT_SWISH       # backbones/conv3d_extractor.py:34, modules/resnet.py:76-78
        self.use_tr = True
        self._side = _SideStream()
        self.grad_ready_hook = None
        self._pos_cache: dict[tuple[str, int, str], torch.Tensor] = {}
        import weakref

        for name, cls in (("encoder", _EncoderFacade), ("decoder", _DecoderFacade), ("ctc", _CtcFacade)):
            if name not in self._modules:          # mtlalpha = 0: no CTC branch (self.ctc = None in the reference)
                continue
            node = self._modules[name]
            node.__class__ = cls
            object.__setattr__(node, "_owner", weakref.ref(self))

    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def _fwd_rank(name: str) -> int:
        if name.startswith(("encoder.frontend.frontend3D", "encoder.stem3d")):
            return 0
        if name.startswith(("encoder.frontend", "encoder.resnet")):
            return 1
        if name.startswith("encoder.embed"):
            return 2
        if name.startswith("encoder."):
            return 3
        if name.startswith(("ctc.", "audio_classifier", "proj_decoder")):
            return 4
        return 5                 # decoder: last in the forward pass, first to be final in the backward pass

    def _transposed_entries(self, offsets) -> list[tuple[str, int, int, int]]:
        D, Dd, out = self.adim, self.ddim, []

        def one(name: str) -> None:
            s = offsets[name][2]
            out.append((name, offsets[name][0], s[0], math.prod(s[1:])))

        def fused(key: str, names: list[str], d: int) -> None:
            offs = [offsets[f"{n}.weight"][0] for n in names]
            boffs = [offsets[f"{n}.bias"][0] for n in names]
            assert all(offs[i + 1] == offs[i] + d * d for i in range(len(offs) - 1)), f"{key}: weights must be adjacent"
            assert all(boffs[i + 1] == boffs[i] + d for i in range(len(boffs) - 1)), f"{key}: biases must be adjacent"
            out.append((key, offs[0], len(names) * d, d))

        one("encoder.embed.0.weight")
        for i in range(self.elayers):
            p = f"encoder.encoders.{i}"
            fused(f"{p}.self_attn.qkv", [f"{p}.self_attn.linear_{x}" for x in "qkv"], D)
            one(f"{p}.self_attn.linear_out.weight")
            for ff in ("feed_forward", "feed_forward_macaron"):
                one(f"{p}.{ff}.w_1.weight")
                one(f"{p}.{ff}.w_2.weight")
            one(f"{p}.conv_module.pointwise_cov1.weight")
            one(f"{p}.conv_module.pointwise_cov2.weight")
        for i in range(self.dlayers):
            p = f"decoder.decoders.{i}"
            fused(f"{p}.self_attn.qkv", [f"{p}.self_attn.linear_{x}" for x in "qkv"], Dd)
            one(f"{p}.self_attn.linear_out.weight")
            one(f"{p}.src_attn.linear_q.weight")
            fused(f"{p}.src_attn.kv", [f"{p}.src_attn.linear_{x}" for x in "kv"], Dd)
            one(f"{p}.src_attn.linear_out.weight")
            one(f"{p}.feed_forward.w_1.weight")
            one(f"{p}.feed_forward.w_2.weight")
        for n in ("decoder.output_layer.weight", "ctc.ctc_lo.weight", "audio_classifier.weight"):
            if n in offsets:
                one(n)
        if self.adim != self.ddim:
            one("proj_decoder.weight")
        return out

    def configure_optimizers(self):
        """LRS/video/lightning.py:89-96 — the two parameter groups (decay on ndim >= 2)."""
        do_decay = [p for p in self.parameters() if p.requires_grad and p.ndim >= 2]
        no_decay = [p for p in self.parameters() if p.requires_grad and p.ndim < 2]
        return [{"params": do_decay}, {"params": no_decay, "weight_decay": 0.0}]

    def mark_params_dirty(self) -> None:
        if self._store is not None:
            self._store.shadow_fresh = False

    def store(self) -> _ParamStore:
        dev = _get(self, self._specs[0][0]).device
        if self._store is None or self._store.device != dev or not self._store.owns(self):
            self._store = _ParamStore(self, dev)
        return self._store

    def state_dict(self, *args, **kwargs):
        self._side.join()             # (a TrainStep may have left the tail of its optimiser step on the side stream)
        return super().state_dict(*args, **kwargs)

    def _advance_dropout(self, dev: torch.device) -> None:
        if self._drop_word is None or self._drop_word.device != dev:
            self._drop_word = torch.tensor([self.dropout_seed], dtype=torch.int32, device=dev)
        ops.word_add(self._drop_word, 1)          # a device op (a library launch: recorded by a native step list): replays keep drawing fresh masks

    def reseed_dropout(self, seed: int) -> None:
        self.dropout_seed = int(seed)
        if self._drop_word is not None:          # in place: a captured graph holds this word's address
            self._drop_word.fill_(self.dropout_seed)

    def _d(self, site: str, attn: bool = False):
        """(seed word, site id, p) for ops.*(drop=...) or None when dropout is off (eval mode / p = 0)."""
        p = self.attn_drop_p if attn else self.drop_p
        if not self.training or p <= 0.0:
            return None
        return (self._drop_word, self._sites[site], p)

    def _pos_table(self, kind: str, n: int, dev: torch.device) -> torch.Tensor:
        """rel: bf16 [2n-1, adim], row r <-> relative position n-1-r (embedding.py:180-216); abs: fp32 [n, ddim]."""
        key = (kind, n, str(dev))
        if key not in self._pos_cache:
            if kind == "rel":
                t = _sinusoid(torch.arange(n - 1, -n, -1), self.adim).to(dev).to(BF16).contiguous()
            else:
                t = _sinusoid(torch.arange(n), self.ddim).to(dev).contiguous()
            self._pos_cache[key] = t
        return self._pos_cache[key]

    def prepare_targets(self, label: torch.Tensor) -> LrsTargets:
        """label int64 [B,1,L] or [B,L], padded with ignore_id (dropped wherever it sits in a row, as the reference's add_sos_eos does; a tail
        is the usual case) -> device tensors (no host sync).  Token ids must lie in [1, odim) — 0 is the CTC blank: torch's Embedding /
        CTCLoss stop on anything else with a device assert; the device kernel replaces such a token by eos and sets a sticky error word
        that check_targets() / TrainStep.state() turn into an exception."""
        B = label.size(0)
        lab = label.reshape(B, -1).long()
        if lab.is_cuda:         # one launch (svsr_lrs_targets) instead of the ~20 index operations below, which stay the host-side form
            return LrsTargets(*ops.lrs_targets(lab, self.odim, self.ignore_id, self.eos))
        live = lab != self.ignore_id
        if bool(((lab < 1) | (lab >= self.odim))[live].any()):
            raise ValueError(f"labels must lie in [1, {self.odim}) (0 is the CTC blank)")
        order = torch.argsort((~live).to(torch.int8), dim=1, stable=True)           # live tokens first, in their order
        lab, live = lab.gather(1, order), live.gather(1, order)
        n = live.sum(1, keepdim=True)
        eos = torch.full_like(lab[:, :1], self.eos)
        ys_in = torch.cat([eos, torch.where(live, lab, eos)], dim=1).contiguous()                  # sos == eos (e2e:111-112)
        ys_out = torch.cat([lab, torch.full_like(lab[:, :1], self.ignore_id)], dim=1)
        ys_out = torch.where(live.new_zeros(ys_out.shape).scatter_(1, n, True), torch.full_like(ys_out, self.eos), ys_out)
        labels = torch.where(live, lab, torch.full_like(lab, -1)).contiguous()
        return LrsTargets(labels, ys_in, ys_out.contiguous())

    def check_targets(self) -> None:
        """Raises if a batch handed to prepare_targets since the last check held a label outside [1, odim) (synchronises the device)."""
        if ops.lrs_target_errors(reset=True):
            raise ValueError(f"a label outside [1, {self.odim}) reached svsr_lrs_targets (0 is the CTC blank, ignore_id = {self.ignore_id} marks "
                             f"padding): it was replaced by eos, the losses of that batch are meaningless")

    # ------------------------------------------------------------------------------------------------
    # inference surface (syncvsr_amd/lrs_infer.py): model.encoder(xs, masks), model.decoder.batch_score(...), model.ctc.log_softmax(...)
    def scorers(self) -> dict:
        """e2e_asr_transformer.py:182-184: dict(decoder=self.decoder, ctc=CTCPrefixScorer(self.ctc, self.eos))."""
        from .lrs_infer import CTCPrefixScorer

        return dict(decoder=self.decoder, ctc=CTCPrefixScorer(self, self.eos))

    # ------------------------------------------------------------------------------------------------
    def encode(self, x: torch.Tensor, lengths: Optional[torch.Tensor] = None, layer_outs: Optional[list] = None,
               resnet_feats: bool = False) -> torch.Tensor:
        """`self.encoder(xs, masks)[0]` of the reference (what its inference path calls, LRS/video/lightning.py:100-101,113-116):
        x [B,T,1,H,W] -> fp32 [B,T,adim]; forward only (no autograd), honours train/eval mode for BatchNorm and dropout."""
        if x.device.type != "cuda":
            raise RuntimeError("syncvsr_amd runs on an MI355X HIP device only; there is no CPU fallback (use oracle/ for checking)")
        st = self.store()
        if not st.shadow_fresh:
            st.refresh_shadows()
        B, T = x.shape[:2]
        if lengths is None:
            lengths = torch.full((B,), T, dtype=torch.int32, device=x.device)
        ilen = lengths.to(device=x.device, dtype=torch.int32).contiguous()
        tape: dict[str, Any] = {}
        if layer_outs is not None:
            tape["layer_outs"] = []
        with torch.no_grad():
            res = _encoder_fwd(self, st, tape, x.float().contiguous(), ilen, self.training)
        if resnet_feats:                                   # Encoder.forward(extract_resnet_feats=True), encoder.py:272-273
            return res[0].float().view(B, T, 512)
        if layer_outs is not None:
            layer_outs.extend(t.float().view(B, T, self.adim) for t in tape["layer_outs"])
        return res[2].float().view(B, T, self.adim)

    def forward(self, x: torch.Tensor, lengths: torch.Tensor, audios: torch.Tensor, label):
        if x.device.type != "cuda":
            raise RuntimeError("syncvsr_amd runs on an MI355X HIP device only; there is no CPU fallback (use oracle/ for checking)")
        if x.dim() != 5 or x.size(2) != 1:
            raise ValueError("x must be [B, T, 1, H, W]")
        st = self.store()
        B, T = x.shape[:2]
        A = self.audio_alignment
        if audios.dtype != torch.int64 or audios.dim() != 3:
            raise ValueError("pass pre-computed audio tokens int64 [B, >= A*T, G] in the `audios` slot (SURVEY §8b)")
        if audios.size(1) < T * A:
            raise ValueError(f"audio tokens have {audios.size(1)} steps, need >= {T * A}")
        tokens = audios[:, : T * A].contiguous()
        tg = label if isinstance(label, LrsTargets) else self.prepare_targets(label.to(x.device))
        ilen = lengths.to(device=x.device, dtype=torch.int32).contiguous()
        anchor = _get(self, self._specs[0][0])
        loss_ctc, loss_att, loss_audio, counts = _LrsFunction.apply(anchor, self, st, x.float().contiguous(), ilen, tokens, tg,
                                                                    torch.is_grad_enabled())
        loss = self.mtlalpha * loss_ctc + (1 - self.mtlalpha) * loss_att + self.audio_weight * loss_audio
        acc = counts[0] / counts[1]
        return loss, loss_ctc, loss_att, loss_audio, acc


    # ------------------------------------------------------------------------------------------------
    # native step list (engine.TrainStep(native=True)): the same tape functions without autograd or torch kernels
    def prepare_batch(self, x, lengths, audios, label):
        """The input conversions of forward(), done ahead of it (outside any recorded region): what engine.TrainStep keeps as the static
        inputs of a recorded step.  The decoder / CTC targets (add_sos_eos, e2e_asr_transformer.py:203-215) are derived here, once per batch."""
        if x.dim() != 5 or x.size(2) != 1:
            raise ValueError("x must be [B, T, 1, H, W]")
        T, A = x.size(1), self.audio_alignment
        if audios.dtype != torch.int64 or audios.dim() != 3 or audios.size(1) < T * A:
            raise ValueError(f"pass pre-computed audio tokens int64 [B, >= {T * A}, G] in the `audios` slot (SURVEY §8b)")
        tg = label if isinstance(label, LrsTargets) else self.prepare_targets(label.to(x.device))
        self._pos_table("rel", T, x.device)                 # position tables of this shape exist before a step is recorded
        self._pos_table("abs", tg.ys_in.size(1), x.device)
        return (x.float().contiguous(), lengths.to(device=x.device, dtype=torch.int32).contiguous(), audios[:, : T * A].contiguous(),
                tg.labels, tg.ys_in, tg.ys_out)

    def direct_constants(self, dev) -> None:
        """d loss / d {loss_ctc, loss_att, loss_audio} as device scalars (made once, outside any recorded region)."""
        if getattr(self, "_g_consts", None) is None or self._g_consts[0].device != dev:
            self._g_consts = tuple(torch.full((), w, dtype=torch.float32, device=dev) for w in (self.mtlalpha, 1.0 - self.mtlalpha, self.audio_weight))

    def train_step_direct(self, x, ilen, tokens, labels, ys_in, ys_out):
        """forward + backward of the loss WITHOUT autograd (inputs as prepare_batch returns them): every device operation is a library call,
        so the step can be recorded into a native step list.  -> (loss, loss_ctc, loss_att, loss_audio, acc) as forward()."""
        if self.length_norm:
            raise NotImplementedError("transformer_length_normalized_loss divides by a device value with a torch kernel: use the autograd path")
        st = self.store()
        self.direct_constants(x.device)

        class _Ctx:
            def mark_non_differentiable(self, *a):
                pass

        ctx = _Ctx()
        tg = LrsTargets(labels, ys_in, ys_out)
        loss_ctc, loss_att, loss_audio, counts = _LrsFunction.forward(ctx, None, self, st, x, ilen, tokens, tg, True)
        loss, acc = ops.lincomb3_ratio(loss_ctc, self.mtlalpha, loss_att, 1.0 - self.mtlalpha, loss_audio, self.audio_weight, counts[0:1], counts[1:2])
        _LrsFunction.backward(ctx, *self._g_consts, None)
        return loss, loss_ctc, loss_att, loss_audio, acc


# ----------------------------------------------------------------------------------------------------
# tape helpers
# ----------------------------------------------------------------------------------------------------
def _lin(st: _ParamStore, x, name: str, rows: int, K: int, N: int, *, wkey: Optional[str] = None, n_fused: int = 1, bias: bool = True, **kw):
    """y = x @ W^T + b for the nn.Linear called `name` (or the `n_fused` adjacent linears starting at it)."""
    o = st.offsets[f"{name}.weight"][0]
    w16 = st.w16[o : o + N * K]
    b = st.flat[st.offsets[f"{name}.bias"][0] :][:N] if bias else None
    return ops.linear_fwd(x, w16, b, rows=rows, K=K, N=N, x_pitch=K, **kw)[0]


def _lin_bwd(model, st: _ParamStore, name: str, x, dy, rows: int, K: int, N: int, *, tkey: Optional[str] = None, bias: bool = True,
             need_dx: bool = True, dy_pitch: Optional[int] = None, addend=None, out=None, drop=None):
    """Weight / bias gradients of linear `name` into the flat gradient buffer; returns dx = dy @ W (or None)."""
    dy_pitch = dy_pitch or N
    gw = st.grad[st.offsets[f"{name}.weight"][0] :][: N * K]
    gb = st.grad[st.offsets[f"{name}.bias"][0] :][:N] if bias else None          # column sums of dy, fused into the wgrad launch
    # weight gradients only feed the flat gradient buffer: optionally on the side stream, next to the data-gradient GEMM
    if "lin_wgrad" not in _model_mod._ABLATE:       # (timing experiments only, see model._ABLATE)
        group = model.__dict__.get("_wg_group")
        if group is not None and model.use_tr:      # a decoder layer's weight gradients: collected, one grouped launch per layer (_flush_wg_group)
            group.append(dict(x=x, dy=dy, dw=gw, db=gb, rows=rows, K=K, N=N, x_pitch=K, dy_pitch=dy_pitch))
        else:
            model._side.run(lambda: ops.linear_wgrad(x, dy, gw, rows=rows, K=K, N=N, x_pitch=K, dy_pitch=dy_pitch, use_tr=model.use_tr, db=gb),
                            x, dy, small=True)
    if not need_dx:
        return None
    return ops.linear_dgrad(dy, st.t16(tkey or f"{name}.weight"), rows=rows, N=N, K=K, dy_pitch=dy_pitch, addend=addend, out=out, drop=drop)


WG_GROUP_DECODER = os.environ.get("SVSR_WG_GROUP_DECODER", "1") != "0"
FFN_DGRAD_FUSED = os.environ.get("SVSR_LRS_FFN_DGRAD_FUSED", "1") != "0"      # _ffn_bwd: relu' + bias-gradient partials in the epilogue of w_2's data gradient
PE_AHEAD = os.environ.get("SVSR_LRS_PE_AHEAD", "1") != "0"             # _encoder_fwd: every layer's linear_pos(pos_emb) + its transposed copy on the side stream, under the front-end
TAILS_ON_SIDE = os.environ.get("SVSR_LRS_TAILS_SIDE", "1") != "0"      # _encoder_layer_bwd: parameter-gradient tails of the layer on the side stream


def _flush_wg_group(model) -> None:
    """The weight gradients a decoder layer's backward collected, as ONE launch over a device table of problems (ops.linear_wgrad_group, the
    launch the word-level encoder uses): at ~800 target rows each of the layer's eight contractions is a 13-chunk K loop — 22 us of latency
    apiece as a launch of its own, 48 of them per step on the weight-gradient stream.  (Problems the grouped kernel does not take — the
    source-attention key / value projection over the 2,560 encoder rows — go out on their own inside that call.)"""
    group = model.__dict__.get("_wg_group")
    model._wg_group = None
    if not group:
        return
    keep = [t for q in group for t in (q["x"], q["dy"])]
    model._side.run(lambda: ops.linear_wgrad_group(group), *keep, small=True)


def _ln(st: _ParamStore, x, name: str):
    return ops.add_ln_fwd(x, None, st.p32(f"{name}.weight"), st.p32(f"{name}.bias"), LN_EPS)


def _ln_bwd(model, st: _ParamStore, dy, x, name: str, m, r, addend=None, branch=None):
    """LayerNorm backward (+ skip-path gradient `addend`) -> dx.  branch = (alpha, drop) of the residual branch whose sum x is: the launch
    also writes that branch's gradient alpha * mask / (1 - p) * dx (what _branch_grad(dx, alpha, drop) would launch svsr_scale_bf16 for); the
    result is then a _WithBranch that _branch_grad recognises."""
    # (the gamma / beta reduction is postponed to the side stream: model._defer_list, flushed by _ready at the end of the layer)
    dl = _defer_list(model)
    fuse = branch is not None and (branch[0] != 1.0 or branch[1] is not None) and ops.LN_BRANCH_FUSED and dl is not None
    out = ops.add_ln_bwd(dy, x, None, st.p32(f"{name}.weight"), m, r, st.g32(f"{name}.weight"), st.g32(f"{name}.bias"), addend=addend,
                         defer=dl, branch=branch if fuse else None)
    if fuse:
        dx, dbr = out
        dx._svsr_branch = (float(branch[0]), branch[1], dbr)          # (alpha, drop, gradient of the branch)
        return dx
    return out


def _branch_grad(dy, alpha: float, drop):
    """x' = x + alpha * dropout(y)  ->  dL/dy = alpha * mask/(1-p) * dL/dx'  (the mask is regenerated from its site id)."""
    pre = getattr(dy, "_svsr_branch", None)
    if pre is not None and pre[0] == float(alpha) and pre[1] is drop:       # already written by the launch that produced dy (_ln_bwd)
        return pre[2]
    return ops.scale_bf16(dy, alpha, drop=drop) if (alpha != 1.0 or drop is not None) else dy


def _ffn_fwd(model, st, t: dict, key: str, x, p: str, R: int, D: int, U: int, alpha: float, norm: str, site: str):
    tn, m, r = _ln(st, x, norm)
    dh, do = model._d(f"{site}.hidden"), model._d(f"{site}.out")
    h = _lin(st, tn, f"{p}.w_1", R, D, U, relu=True, drop=dh)                  # dropout(relu(w_1 x)), positionwise_feed_forward.py:30
    y = _lin(st, h, f"{p}.w_2", R, U, D, addend=x, alpha=alpha, drop=do)       # x + ff_scale * dropout(w_2 h)
    t[key] = dict(x=x, tn=tn, m=m, r=r, h=h, dh=dh, do=do)
    return y


def _ffn_bwd(model, st, t: dict, dy, p: str, R: int, D: int, U: int, alpha: float, norm: str, branch=None):
    """x' = x + alpha * dropout(FFN(LN(x))); dy = grad of x' -> grad of x.  branch: (alpha, drop) of the branch in front (see _ln_bwd)."""
    dys = _branch_grad(dy, alpha, t["do"])
    gs = 1.0 / (1.0 - t["dh"][2]) if t["dh"] is not None else 1.0           # dropped hidden units are the zeros of the saved h
    if FFN_DGRAD_FUSED and U % 64 == 0 and model.use_tr:
        # (round 6) relu' / the dropout mask and the bias gradient's partial rows in the epilogue of w_2's data gradient: one launch fewer
        # per feed-forward block in the main stream's chain, two passes over [R, U] fewer
        _lin_bwd(model, st, f"{p}.w_2", t["h"], dys, R, U, D, need_dx=False)
        dz, (part, tiles) = ops.linear_dgrad_relu(dys, st.t16(f"{p}.w_2.weight"), rows=R, N=D, K=U, dy_pitch=D, y=t["h"], gscale=gs)
        dl, gb = _defer_list(model), st.g32(f"{p}.w_1.bias")
        if dl is not None:
            dl.append((ops._deferred_colsum(part, tiles, 2 * U, gb, U), part))
        else:
            ops.colsum_rows(part, tiles, 2 * U, gb, U)
    else:
        dh = _lin_bwd(model, st, f"{p}.w_2", t["h"], dys, R, U, D)
        dz = ops.bias_act_bwd(dh, t["h"], st.g32(f"{p}.w_1.bias"), R=R, N=U, n_valid=U, ld=U, relu=True, gscale=gs, defer=_defer_list(model))
    dtn = _lin_bwd(model, st, f"{p}.w_1", t["tn"], dz, R, D, U, bias=False)
    return _ln_bwd(model, st, dtn, t["x"], norm, t["m"], t["r"], addend=dy, branch=branch)


def _encoder_layer_fwd(model: E2E, st: _ParamStore, tape: dict, i: int, x, pos16, ilen, B: int, T: int, training: bool):
    D, U, H, K = model.adim, model.eunits, model.aheads, model.kernel
    R = B * T
    p = f"encoder.encoders.{i}"
    t: dict[str, Any] = {}
    x1 = _ffn_fwd(model, st, t, "ffm", x, f"{p}.feed_forward_macaron", R, D, U, 0.5, f"{p}.norm_ff_macaron", f"enc.{i}.ffm")
    # relative-position self-attention
    t2, m2, r2 = _ln(st, x1, f"{p}.norm_mha")
    qkv = _lin(st, t2, f"{p}.self_attn.linear_q", R, D, 3 * D)
    ahead = tape.get("_pe_ahead")
    pe, pet = ahead[i] if ahead is not None else (_lin(st, pos16, f"{p}.self_attn.linear_pos", 2 * T - 1, D, D, bias=False), None)
    bu, bv = st.p32(f"{p}.self_attn.pos_bias_u"), st.p32(f"{p}.self_attn.pos_bias_v")
    dpr, dao = model._d(f"enc.{i}.attn.probs", attn=True), model._d(f"enc.{i}.attn.out")
    ctx, probs = ops.mha_fwd(qkv, 3 * D, qkv[:, D:], qkv[:, 2 * D:], 3 * D, B=B, H=H, Lq=T, Lk=T, pe=pe, bias_u=bu, bias_v=bv, klen=ilen,
                             drop=dpr, flash=True)
    x2 = _lin(st, ctx, f"{p}.self_attn.linear_out", R, D, D, addend=x1, drop=dao)
    t["mha"] = dict(x=x1, tn=t2, m=m2, r=r2, qkv=qkv, pe=pe, pet=pet, ctx=ctx, probs=probs, dpr=dpr, dao=dao)
    # convolution module
    t3, m3, r3 = _ln(st, x2, f"{p}.norm_conv")
    cm = f"{p}.conv_module"
    u = _lin(st, t3, f"{cm}.pointwise_cov1", R, D, 2 * D)
    bn = f"{cm}.norm"
    c, stats = ops.glu_dwconv_fwd(u, st.p32(f"{cm}.depthwise_conv.weight"), st.p32(f"{cm}.depthwise_conv.bias"), training, B, T, D, K)
    mean, rstd = _bn_stats(st, bn, training, R, stats)
    y = ops.bn_act_fwd(c, None, mean, rstd, st.p32(f"{bn}.weight"), st.p32(f"{bn}.bias"), ops.ACT_SWISH)
    dco = model._d(f"enc.{i}.conv.out")
    x3 = _lin(st, y, f"{cm}.pointwise_cov2", R, D, D, addend=x2, drop=dco)
    t["conv"] = dict(x=x2, tn=t3, m=m3, r=r3, u=u, c=c, y=y, mean=mean, rstd=rstd, dco=dco)
    x4 = _ffn_fwd(model, st, t, "ff", x3, f"{p}.feed_forward", R, D, U, 0.5, f"{p}.norm_ff", f"enc.{i}.ff")
    xo, m5, r5 = _ln(st, x4, f"{p}.norm_final")
    t["final"] = dict(x=x4, m=m5, r=r5)
    tape[p] = t
    return xo


def _encoder_layer_bwd(model: E2E, st: _ParamStore, tape: dict, i: int, dxo, pos16, B: int, T: int):
    D, U, H, K = model.adim, model.eunits, model.aheads, model.kernel
    R = B * T
    p = f"encoder.encoders.{i}"
    t = tape[p]
    tf = t["final"]
    dx4 = _ln_bwd(model, st, dxo, tf["x"], f"{p}.norm_final", tf["m"], tf["r"], branch=(0.5, t["ff"]["do"]))
    dx3 = _ffn_bwd(model, st, t["ff"], dx4, f"{p}.feed_forward", R, D, U, 0.5, f"{p}.norm_ff", branch=(1.0, t["conv"]["dco"]))
    # convolution module
    tc = t["conv"]
    cm, bn = f"{p}.conv_module", f"{p}.conv_module.norm"
    ws = st.bn[bn]
    if ops.BN_BWD_FUSED:
        # the data gradient of pointwise_conv2 is the gradient of swish(bn(c)): its launch multiplies by swish' and takes the
        # BatchNorm backward's first pass in its epilogue (ops.linear_dgrad_bn)
        dyo = _branch_grad(dx3, 1.0, tc["dco"])
        _lin_bwd(model, st, f"{cm}.pointwise_cov2", tc["y"], dyo, R, D, D, need_dx=False)
        g, gst = ops.linear_dgrad_bn(dyo, st.t16(f"{cm}.pointwise_cov2.weight"), rows=R, N=D, K=D, dy_pitch=D, x=tc["c"], mean=tc["mean"],
                                     rstd=tc["rstd"], gamma=st.p32(f"{bn}.weight"), beta=st.p32(f"{bn}.bias"), act=ops.ACT_SWISH)
        dc = ops.bn_bwd_from_stats(g, tc["c"], tc["mean"], tc["rstd"], st.p32(f"{bn}.weight"), gst, ws["coef"], st.g32(f"{bn}.weight"),
                                   st.g32(f"{bn}.bias"))
    else:
        dy = _lin_bwd(model, st, f"{cm}.pointwise_cov2", tc["y"], _branch_grad(dx3, 1.0, tc["dco"]), R, D, D)
        dc, _ = ops.bn_act_bwd(dy, tc["y"], tc["c"], tc["mean"], tc["rstd"], st.p32(f"{bn}.weight"), ws["coef"], st.g32(f"{bn}.weight"),
                               st.g32(f"{bn}.bias"), ops.ACT_SWISH, False, beta=st.p32(f"{bn}.bias"))
    # (round 6) the two parameter-gradient tails of this layer — the depthwise convolution's partial-row sum and the position-table pass of
    # the attention backward — feed nothing but gradients of parameters: with the weight gradients on the side stream they go there too
    # (24 launches fewer in the main stream's chain per layer pair; same launches, same bits)
    later = model._side.enabled and TAILS_ON_SIDE
    du = ops.glu_dwconv_bwd(dc, tc["u"], st.p32(f"{cm}.depthwise_conv.weight"), st.g32(f"{cm}.depthwise_conv.weight"),
                            st.g32(f"{cm}.depthwise_conv.bias"), B, T, D, K, reduce_later=later)
    if later:
        du, (fn_dw, keep_dw) = du
        model._side.run(fn_dw, *keep_dw)
    dt3 = _lin_bwd(model, st, f"{cm}.pointwise_cov1", tc["tn"], du, R, D, 2 * D)
    dx2 = _ln_bwd(model, st, dt3, tc["x"], f"{p}.norm_conv", tc["m"], tc["r"], addend=dx3, branch=(1.0, t["mha"]["dao"]))
    # attention
    tm = t["mha"]
    sa = f"{p}.self_attn"
    dctx = _lin_bwd(model, st, f"{sa}.linear_out", tm["ctx"], _branch_grad(dx2, 1.0, tm["dao"]), R, D, D)
    qkv = tm["qkv"]
    dqkv = torch.empty_like(qkv)
    res = ops.mha_bwd(dctx, qkv, 3 * D, qkv[:, D:], qkv[:, 2 * D:], 3 * D, tm["probs"], B=B, H=H, Lq=T, Lk=T, dq=dqkv,
                      dq_pitch=3 * D, dk=dqkv[:, D:], dv=dqkv[:, 2 * D:], dkv_pitch=3 * D, pe=tm["pe"],
                      bias_u=st.p32(f"{sa}.pos_bias_u"), bias_v=st.p32(f"{sa}.pos_bias_v"), drop=tm["dpr"], pe_later=later, pet=tm.get("pet"))
    dq_ac, dq_bd, dpe = res[:3]
    if len(res) == 4:          # dpe is filled on the side stream, in front of the weight gradient of linear_pos (same stream, later)
        model._side.run(res[3][0], *res[3][1])
    # pos_bias_u / pos_bias_v gradients: column sums nothing downstream reads — both launches on the side stream
    gu, gv = st.g32(f"{sa}.pos_bias_u"), st.g32(f"{sa}.pos_bias_v")
    model._side.run(lambda: (ops.bias_act_bwd(dq_ac, None, gu, R=R, N=D, n_valid=D, ld=D), ops.bias_act_bwd(dq_bd, None, gv, R=R, N=D, n_valid=D, ld=D)),
                    dq_ac, dq_bd, small=True)
    _lin_bwd(model, st, f"{sa}.linear_pos", pos16, dpe, 2 * T - 1, D, D, bias=False, need_dx=False)
    dt2 = _lin_bwd(model, st, f"{sa}.linear_q", tm["tn"], dqkv, R, D, 3 * D, tkey=f"{sa}.qkv")
    dx1 = _ln_bwd(model, st, dt2, tm["x"], f"{p}.norm_mha", tm["m"], tm["r"], addend=dx2, branch=(0.5, t["ffm"]["do"]))
    dx = _ffn_bwd(model, st, t["ffm"], dx1, f"{p}.feed_forward_macaron", R, D, U, 0.5, f"{p}.norm_ff_macaron")
    _ready(model, st, f"{p}.self_attn.linear_q.weight")
    return dx


def _decoder_fwd(model: E2E, st: _ParamStore, tape: dict, tg: LrsTargets, memory, ilen, B: int, T: int):
    D, U, H = model.ddim, model.dunits, model.dheads
    L = tg.ys_in.size(1)
    R = B * L
    pe = model._pos_table("abs", L, memory.device)
    x = ops.embed_pos_fwd(tg.ys_in, st.p32("decoder.embed.0.weight"), pe, L, D, math.sqrt(D))
    dde = model._d("dec.embed")
    if dde is not None:
        ops.scale_bf16(x, 1.0, drop=dde, out=x)                                 # PositionalEncoding's dropout, embedding.py:89
    tape["dec_embed_drop"] = dde
    for i in range(model.dlayers):
        p = f"decoder.decoders.{i}"
        t: dict[str, Any] = {}
        t1, m1, r1 = _ln(st, x, f"{p}.norm1")
        qkv = _lin(st, t1, f"{p}.self_attn.linear_q", R, D, 3 * D)
        dsp, dso = model._d(f"dec.{i}.self.probs", attn=True), model._d(f"dec.{i}.self.out")
        ctx, probs = ops.mha_fwd(qkv, 3 * D, qkv[:, D:], qkv[:, 2 * D:], 3 * D, B=B, H=H, Lq=L, Lk=L, causal=True, drop=dsp, flash=True)
        x1 = _lin(st, ctx, f"{p}.self_attn.linear_out", R, D, D, addend=x, drop=dso)
        t["self"] = dict(x=x, tn=t1, m=m1, r=r1, qkv=qkv, ctx=ctx, probs=probs, dpr=dsp, dao=dso)
        t2, m2, r2 = _ln(st, x1, f"{p}.norm2")
        q = _lin(st, t2, f"{p}.src_attn.linear_q", R, D, D)
        kv = _lin(st, memory, f"{p}.src_attn.linear_k", B * T, D, 2 * D)
        dcp, dco = model._d(f"dec.{i}.src.probs", attn=True), model._d(f"dec.{i}.src.out")
        ctx2, probs2 = ops.mha_fwd(q, D, kv, kv[:, D:], 2 * D, B=B, H=H, Lq=L, Lk=T, klen=ilen, drop=dcp, flash=True)
        x2 = _lin(st, ctx2, f"{p}.src_attn.linear_out", R, D, D, addend=x1, drop=dco)
        t["src"] = dict(x=x1, tn=t2, m=m2, r=r2, q=q, kv=kv, ctx=ctx2, probs=probs2, dpr=dcp, dao=dco)
        x = _ffn_fwd(model, st, t, "ff", x2, f"{p}.feed_forward", R, D, U, 1.0, f"{p}.norm3", f"dec.{i}.ff")
        t["out"] = x
        tape[p] = t
    tn, m, r = _ln(st, x, "decoder.after_norm")
    V = model.odim
    Vp = (V + 63) // 64 * 64
    pred = ops.linear_fwd(tn, st.s16("decoder.output_layer.weight"), st.p32("decoder.output_layer.bias"), rows=R, K=D, N=V, x_pitch=D,
                          out_f32=True, out_pitch=Vp)[0]
    tape["dec_out"] = dict(x=x, tn=tn, m=m, r=r, pred=pred, L=L, Vp=Vp)
    return pred


def _decoder_bwd(model: E2E, st: _ParamStore, tape: dict, tg: LrsTargets, dpred, memory, dmem, B: int, T: int):
    """dpred bf16 [B*L, Vp]; accumulates the source-attention key/value gradients into dmem [B*T, D]."""
    D, U, H = model.ddim, model.dunits, model.dheads
    to = tape["dec_out"]
    L, Vp, V = to["L"], to["Vp"], model.odim
    R = B * L
    model._wg_group = None      # (a backward that aborted inside a layer must not leave its half-filled group: the launches below would join it and be dropped)
    dtn = _lin_bwd(model, st, "decoder.output_layer", to["tn"], dpred, R, D, V, dy_pitch=Vp)
    # (every LayerNorm backward below also writes the gradient of the residual branch that follows it: _ln_bwd's `branch`)
    dx = _ln_bwd(model, st, dtn, to["x"], "decoder.after_norm", to["m"], to["r"],
                 branch=(1.0, tape[f"decoder.decoders.{model.dlayers - 1}"]["ff"]["do"]) if model.dlayers > 0 else None)
    _ready(model, st, "decoder.output_layer.weight")
    for i in reversed(range(model.dlayers)):
        p = f"decoder.decoders.{i}"
        t = tape[p]
        model._wg_group = [] if WG_GROUP_DECODER else None
        dx2 = _ffn_bwd(model, st, t["ff"], dx, f"{p}.feed_forward", R, D, U, 1.0, f"{p}.norm3", branch=(1.0, t["src"]["dao"]))
        ts = t["src"]
        dctx2 = _lin_bwd(model, st, f"{p}.src_attn.linear_out", ts["ctx"], _branch_grad(dx2, 1.0, ts["dao"]), R, D, D)
        dq = torch.empty_like(ts["q"])
        dkv = torch.empty_like(ts["kv"])
        ops.mha_bwd(dctx2, ts["q"], D, ts["kv"], ts["kv"][:, D:], 2 * D, ts["probs"], B=B, H=H, Lq=L, Lk=T, dq=dq, dq_pitch=D, dk=dkv,
                    dv=dkv[:, D:], dkv_pitch=2 * D, drop=ts["dpr"])
        _lin_bwd(model, st, f"{p}.src_attn.linear_k", memory, dkv, B * T, D, 2 * D, tkey=f"{p}.src_attn.kv", addend=dmem, out=dmem)
        dt2 = _lin_bwd(model, st, f"{p}.src_attn.linear_q", ts["tn"], dq, R, D, D)
        tsf = t["self"]
        dx1 = _ln_bwd(model, st, dt2, ts["x"], f"{p}.norm2", ts["m"], ts["r"], addend=dx2, branch=(1.0, tsf["dao"]))
        dctx = _lin_bwd(model, st, f"{p}.self_attn.linear_out", tsf["ctx"], _branch_grad(dx1, 1.0, tsf["dao"]), R, D, D)
        qkv = tsf["qkv"]
        dqkv = torch.empty_like(qkv)
        ops.mha_bwd(dctx, qkv, 3 * D, qkv[:, D:], qkv[:, 2 * D:], 3 * D, tsf["probs"], B=B, H=H, Lq=L, Lk=L, dq=dqkv, dq_pitch=3 * D,
                    dk=dqkv[:, D:], dv=dqkv[:, 2 * D:], dkv_pitch=3 * D, drop=tsf["dpr"])
        dt1 = _lin_bwd(model, st, f"{p}.self_attn.linear_q", tsf["tn"], dqkv, R, D, 3 * D, tkey=f"{p}.self_attn.qkv")
        nxt = tape[f"decoder.decoders.{i - 1}"]["ff"]["do"] if i > 0 else tape["dec_embed_drop"]
        dx = _ln_bwd(model, st, dt1, tsf["x"], f"{p}.norm1", tsf["m"], tsf["r"], addend=dx1, branch=(1.0, nxt))
        _flush_wg_group(model)
        _ready(model, st, f"{p}.self_attn.linear_q.weight")
    ops.embed_pos_bwd(tg.ys_in, _branch_grad(dx, 1.0, tape["dec_embed_drop"]), st.g32("decoder.embed.0.weight"), D, math.sqrt(D))
    _ready(model, st, "decoder.embed.0.weight")


def _encoder_fwd(model: E2E, st: _ParamStore, tape: dict, x, ilen, training: bool):
    """`Encoder.forward` (transformer/encoder.py:257-289): front-end, embed, Conformer layers, after_norm."""
    B, T = x.shape[:2]
    D, R = model.adim, B * T
    videos = x.view(B, 1, T, x.size(3), x.size(4))             # [B,T,1,H,W] and [B,1,T,H,W] are the same memory (C = 1)
    if training and (model.drop_p > 0.0 or model.attn_drop_p > 0.0):
        model._advance_dropout(x.device)
    pos16 = model._pos_table("rel", T, x.device)
    dpos = model._d("enc.embed.pos")
    if dpos is not None:
        pos16 = ops.scale_bf16(pos16, 1.0, drop=dpos)                                      # dropout(pos_emb), embedding.py:217
    if training and model._side.enabled and PE_AHEAD:
        # (round 6) every layer's projection of the position table — linear_pos(pos_emb), attention.py:238-250 — and its transposed copy for the
        # attention backward depend on the step's weights and on pos_emb alone: all of them are made on the side stream while the front-end
        # runs (behind the previous step's optimiser there; the join below is the one that existed), 24 launches off the main chain
        pre: dict = {}
        tape["_pe_ahead"] = pre

        def ahead(pos16=pos16) -> None:
            for i in range(model.elayers):
                pe = _lin(st, pos16, f"encoder.encoders.{i}.self_attn.linear_pos", 2 * T - 1, D, D, bias=False)
                pre[i] = (pe, ops.mha_pe_transpose(pe, model.aheads, T))

        model._side.run(ahead, pos16)
        model._side.flush()
    feats = _frontend_forward(model, st, tape, videos, training)          # [R, 512] bf16
    model._side.join()                # the previous step's optimiser may still be updating everything behind the front-end on the side stream (engine.TrainStep)
    dex = model._d("enc.embed.x")
    h = _lin(st, feats, "encoder.embed.0", R, 512, D, alpha=math.sqrt(D), drop=dex)        # dropout(x * xscale), embedding.py:208,217
    for i in range(model.elayers):
        h = _encoder_layer_fwd(model, st, tape, i, h, pos16, ilen, B, T, training)
        if "layer_outs" in tape:
            tape["layer_outs"].append(h)
    hx = h
    h, mA, rA = _ln(st, hx, "encoder.after_norm")
    return feats, hx, h, mA, rA, pos16, dex


class _LrsFunction(torch.autograd.Function):
    """One autograd node for E2E.forward: returns (loss_ctc, loss_att, loss_audio, counts); backward replays the tape."""

    @staticmethod
    def forward(ctx, _anchor, model: E2E, st: _ParamStore, x, ilen, tokens, tg: LrsTargets, need_grad: bool):
        training = model.training
        B, T = x.shape[:2]
        D = model.adim
        R = B * T
        A, G, V = model.audio_alignment, model.vq_groups, model.audio_vocab_size
        if not st.shadow_fresh:
            st.refresh_shadows()
        tape: dict[str, Any] = {}
        feats, hx, h, mA, rA, pos16, dex = _encoder_fwd(model, st, tape, x, ilen, training)
        # audio head (e2e_asr_transformer.py:194-201): every frame, no padding mask
        NA = A * G * V
        logits_a = _lin(st, h, "audio_classifier", R, D, NA)
        tok = tokens.reshape(-1)
        loss_a, lse_a = ops.ce_fwd(logits_a, V, tok, None, R * A * G, V, 0.0)
        # CTC head (ctc.py:83-151)
        Vo = model.odim
        Vp = (Vo + 63) // 64 * 64
        if model.mtlalpha > 0.0:
            dctc = model._d("ctc.in")
            box: dict[str, Any] = {}

            def ctc_branch():
                hc = ops.scale_bf16(h, 1.0, drop=dctc) if dctc is not None else h               # ctc_lo(dropout(hs_pad)), ctc.py:97
                lc = ops.linear_fwd(hc, st.s16("ctc.ctc_lo.weight"), st.p32("ctc.ctc_lo.bias"), rows=R, K=D, N=Vo, x_pitch=D, out_f32=True,
                                    out_pitch=Vp)[0]
                box["h_ctc"], box["logits_c"] = hc, lc
                box["loss_c"], box["ctc_state"] = ops.ctc_fwd(lc, Vp, tg.labels, ilen, B, T, Vo)

            # the CTC branch (its lattice is T sequential steps: ~300 us of latency, not of work) runs on the side stream next to the
            # attention decoder's forward; both only read the encoder output.  Joined below, before anything uses its results.
            if ops.CTC_SIDE:
                model._side.run(ctc_branch, h, small=True)
            else:
                ctc_branch()
        else:                                      # `loss_ctc = 0` (e2e_asr_transformer.py:205-208)
            dctc = h_ctc = logits_c = ctc_state = None
            loss_c = ops.zeros((), torch.float32, x.device)
        # attention decoder + label smoothing (decoder.py:122-151, label_smoothing_loss.py:41-63)
        # proj_decoder when the decoder is narrower/wider than the encoder (e2e_asr_transformer.py:93-95,209-210)
        memory = _lin(st, h, "proj_decoder", R, D, model.ddim) if model.adim != model.ddim else h
        pred = _decoder_fwd(model, st, tape, tg, memory, ilen, B, T)
        L = tg.ys_out.size(1)
        tgt = tg.ys_out.reshape(-1)
        # label_smoothing_loss.py:62: / batch size, or / number of live tokens (transformer_length_normalized_loss); the token
        # count stays on the device (counts[1]), so the division is a 0-d tensor op and nothing syncs
        inv_denom = 1.0 if model.length_norm else 1.0 / B
        loss_att, lse_p, counts = ops.ls_loss_fwd(pred, Vp, tgt, B * L, Vo, model.lsm_weight, inv_denom)
        if model.length_norm:
            loss_att = loss_att / counts[1]
        if model.mtlalpha > 0.0:
            if ops.CTC_SIDE:
                model._side.join()
            h_ctc, logits_c, loss_c, ctc_state = box["h_ctc"], box["logits_c"], box["loss_c"], box["ctc_state"]
        model._last = dict(feats=feats, enc_out=h, pred=pred, logits_audio=logits_a, logits_ctc=logits_c)
        if need_grad:
            tape["head"] = dict(hx=hx, h=h, mA=mA, rA=rA, logits_a=logits_a, lse_a=lse_a, tok=tok, logits_c=logits_c, ctc_state=ctc_state,
                                pred=pred, lse_p=lse_p, tgt=tgt, inv_denom=inv_denom, dims=(B, T, L, Vp), pos16=pos16, ilen=ilen, feats=feats,
                                h_ctc=h_ctc, dctc=dctc, dex=dex, memory=memory, counts=counts)
            ctx.tape, ctx.model, ctx.st, ctx.tg = tape, model, st, tg
        ctx.mark_non_differentiable(counts)
        return loss_c, loss_att, loss_a, counts

    @staticmethod
    def backward(ctx, g_ctc, g_att, g_audio, _g_counts):
        model, st, tape, tg = ctx.model, ctx.st, ctx.tape, ctx.tg
        th = tape["head"]
        B, T, L, Vp = th["dims"]
        D, Vo = model.adim, model.odim
        R = B * T
        A, G, V = model.audio_alignment, model.vq_groups, model.audio_vocab_size
        dev = th["h"].device
        if not getattr(model, "accumulate_grads", False):
            st.zero_grad()
        st.rebind_grads()

        def scalar(g):
            return (g if g is not None else torch.zeros((), device=dev)).float().contiguous()

        g_ctc, g_att, g_audio = scalar(g_ctc), scalar(g_att), scalar(g_audio)
        h = th["h"]
        dh = ops.zeros((R, D), BF16, dev)
        # decoder first: its parameters sit at the end of the flat gradient buffer
        if model.length_norm:
            g_att = (g_att / th["counts"][1]).contiguous()
        dpred = ops.ls_loss_bwd(th["pred"], Vp, th["tgt"], B * L, Vo, model.lsm_weight, th["inv_denom"], th["lse_p"], g_att, Vp)
        if model.adim != model.ddim:
            dmem = ops.zeros((R, model.ddim), BF16, dev)
            _decoder_bwd(model, st, tape, tg, dpred, th["memory"], dmem, B, T)
            _lin_bwd(model, st, "proj_decoder", h, dmem, R, D, model.ddim, addend=dh, out=dh)
        else:
            _decoder_bwd(model, st, tape, tg, dpred, h, dh, B, T)
        # CTC and audio heads
        if th["logits_c"] is not None:
            dlc = ops.ctc_grad(th["logits_c"], Vp, tg.labels, th["ilen"], B, T, Vo, th["ctc_state"], g_ctc, Vp)
            _lin_bwd(model, st, "ctc.ctc_lo", th["h_ctc"], dlc, R, D, Vo, dy_pitch=Vp, addend=dh, out=dh, drop=th["dctc"])   # dh += mask/(1-p) * (dlc W)
        NA = A * G * V
        dla = torch.empty((R, NA), dtype=BF16, device=dev)
        ops.ce_bwd(th["logits_a"], V, th["tok"], None, R * A * G, V, 0.0, th["lse_a"], g_audio, dla, V)
        _lin_bwd(model, st, "audio_classifier", h, dla, R, D, NA, addend=dh, out=dh)
        _ready(model, st, "ctc.ctc_lo.weight" if th["logits_c"] is not None else "audio_classifier.weight")
        dx = _ln_bwd(model, st, dh, th["hx"], "encoder.after_norm", th["mA"], th["rA"])
        for i in reversed(range(model.elayers)):
            dx = _encoder_layer_bwd(model, st, tape, i, dx, th["pos16"], B, T)
        dfeats = _lin_bwd(model, st, "encoder.embed.0", th["feats"], _branch_grad(dx, math.sqrt(D), th["dex"]), R, 512, D)
        _ready(model, st, "encoder.embed.0.weight")
        _frontend_backward(model, st, tape, dfeats)
        ctx.tape = None
        return None, None, None, None, None, None, None, None
