"""ORACLE — CPU restatement (plain torch fp32) of the SyncVSR LRW training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``syncvsr_amd/`` may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and only as the checker /
the timed CPU baseline ("port"), never as the product path.

Parity status: PINNED.  ``tests/golden/make_golden_lrw.py`` imports the reference itself
(``/root/reference/LRW/video/src/lightning.py`` with import stubs) in the build
container, loads the same seeded weights, and records outputs/gradients in ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this restatement against those vectors to ~1e-5.

Third-party arithmetic that is not in the reference tree:
  * timm ``resnet18`` (unpinned, ``LRW/video/setup.sh:35``)  -> restated from the topology-identical
    in-tree classes ``LRW/video/src/tcn/models/resnet.py:28-72`` (relu) — the golden generator
    instantiates exactly that class in place of timm.
  * HF ``transformers.BertModel`` (unpinned; 5.15.0 in the build container) -> restated below from
    its published post-LN algorithm; the golden generator runs the real ``BertModel``.
  * x-transformers 1.9.2 ``Encoder`` (``LRW/video/setup.sh``; selected by the shipped yamls) is neither vendored in the
    reference tree nor importable offline.  ``xt_encoder`` below restates its published algorithm (pre-norm residual
    blocks, RMSNorm, rotary embedding on the first 32 of 64 head dims, GEGLU feed-forward, stochastic layer skipping)
    from the package's documented behaviour, anchored on the reference's call site (lightning.py:93-105,157-158).
    PARITY UNPINNED for that branch: there is no golden vector for it; the HIP path is tested against this restatement only.

Every function cites the reference lines it follows.  All maths is fp32 on CPU.
"""
from __future__ import annotations

import math
from typing import Any

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = dict[str, Tensor]

BN_EPS = 1e-5       # torch default, never overridden by the reference (lightning.py:51)
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------------
# bf16-storage emulation (`emu=True`): the SAME restatement with every tensor the HIP path keeps in bf16 rounded to bf16 at the
# point where that path stores it — activations in the forward pass, their gradients in the backward pass — and the MFMA
# kernels' weight operands taken from the bf16 shadow.  Arithmetic stays fp32 (as the kernels' accumulators and BatchNorm /
# LayerNorm statistics are).  The reference itself trains under bf16 autocast (config `precision: bf16`), so neither mode is
# "the" reference result; this one isolates what the fp32 comparison cannot: whether the HIP kernels compute the right
# function of the rounded values (ReLU masks, pooling winners and BatchNorm sums then agree instead of flipping at random).
# --------------------------------------------------------------------------------------------
class _RoundBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class _RoundValueBf16(torch.autograd.Function):
    """bf16 rounding of the VALUE only: a tensor the HIP forward stores in bf16 whose gradient the backward never materialises."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGradBf16(torch.autograd.Function):
    """Identity whose GRADIENT is rounded to bf16: a point where only the HIP path's backward stores a bf16 tensor."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def _st(x: Tensor, emu: bool) -> Tensor:
    """A tensor the HIP path stores in bf16 (value and gradient)."""
    return _RoundBf16.apply(x) if emu else x


class _BnEmu(torch.autograd.Function):
    """Training-mode BatchNorm as the HIP path evaluates it around a bf16-stored input: statistics from the convolution's fp32
    accumulators `c32`, normalisation of its bf16 copy, and a backward pass that forms
        dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat))
    in fp32 and stores THAT in bf16 (norm_act.hip k_bn_act_bwd_apply) — one rounding of the whole expression, where composing
    autograd nodes would round the direct term only.  The difference is zero-mean noise per element, but a weight gradient sums
    it against the block input, and where that input has a large common component (the stem's GELU output feeding layer1) the
    sum has nothing else left: measured cosine 0.9946 -> 0.9999 on resnet.layer1.0.conv1.weight."""

    @staticmethod
    def forward(ctx, c32, w, b):
        dims = [d for d in range(c32.dim()) if d != 1]
        shape = [1, -1] + [1] * (c32.dim() - 2)
        mean = c32.mean(dims)
        rstd = torch.rsqrt(c32.var(dims, unbiased=False) + BN_EPS)
        xhat = (c32.bfloat16().float() - mean.view(shape)) * rstd.view(shape)
        ctx.save_for_backward(xhat, rstd, w)
        ctx.dims, ctx.shape = dims, shape
        return xhat * w.view(shape) + b.view(shape)

    @staticmethod
    def backward(ctx, g):
        xhat, rstd, w = ctx.saved_tensors
        dims, shape = ctx.dims, ctx.shape
        sg, sgx = g.sum(dims), (g * xhat).sum(dims)
        n = g.numel() // g.size(1)
        dx = (w * rstd).view(shape) * (g - (sg / n).view(shape) - xhat * (sgx / n).view(shape))
        return dx.bfloat16().float(), sgx, sg


def _w16(w: Tensor, emu: bool) -> Tensor:
    """The bf16 shadow of a weight the MFMA kernels read; its gradient reaches the fp32 master unrounded."""
    return w.bfloat16().float() if emu else w


# --------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------
def batch_norm(x: Tensor, sd: SD, prefix: str, training: bool, stats_out: dict | None = None, x_stats: Tensor | None = None) -> Tensor:
    """BatchNorm{2d,3d} over all non-channel dims (channel = dim 1).  SURVEY App. A.1.

    training: biased batch variance for normalisation; running stats updated with momentum 0.1 and the
    *unbiased* variance.  eval: running statistics.  x_stats (bf16-storage emulation, training): the convolution's fp32
    accumulators — the batch statistics are taken from them while their bf16-rounded copy is what gets normalised (_BnEmu; `x` is
    then only the eval-mode input).
    """
    w, b = sd[f"{prefix}.weight"], sd[f"{prefix}.bias"]
    dims = [d for d in range(x.dim()) if d != 1]
    shape = [1, -1] + [1] * (x.dim() - 2)
    if training and x_stats is not None:
        if stats_out is not None:
            with torch.no_grad():
                n = x_stats.numel() // x_stats.size(1)
                mean, var = x_stats.mean(dims), x_stats.var(dims, unbiased=False)
                rm, rv = sd[f"{prefix}.running_mean"], sd[f"{prefix}.running_var"]
                stats_out[f"{prefix}.running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
                stats_out[f"{prefix}.running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * n / max(n - 1, 1)
                stats_out[f"{prefix}.num_batches_tracked"] = sd[f"{prefix}.num_batches_tracked"] + 1
        return _BnEmu.apply(x_stats, w, b)
    if training:
        xs = x
        mean = xs.mean(dims)
        var = xs.var(dims, unbiased=False)
        if stats_out is not None:
            n = x.numel() // x.size(1)
            rm, rv = sd[f"{prefix}.running_mean"], sd[f"{prefix}.running_var"]
            stats_out[f"{prefix}.running_mean"] = ((1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean).detach()
            stats_out[f"{prefix}.running_var"] = ((1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * n / max(n - 1, 1)).detach()
            stats_out[f"{prefix}.num_batches_tracked"] = sd[f"{prefix}.num_batches_tracked"] + 1
    else:
        mean, var = sd[f"{prefix}.running_mean"], sd[f"{prefix}.running_var"]
    xhat = (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + BN_EPS)
    return xhat * w.view(shape) + b.view(shape)


def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() exact form (lightning.py:52)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


# --------------------------------------------------------------------------------------------
# visual front-end  (lightning.py:49-55, 112-119)
# --------------------------------------------------------------------------------------------
def stem3d(videos: Tensor, sd: SD, training: bool, stats_out: dict | None = None, keep: dict | None = None, emu: bool = False,
           prefix: str = "stem3d", act=None) -> Tensor:
    """Conv3d(1,64,(5,7,7),(1,2,2),(2,3,3)) -> BatchNorm3d -> GELU -> MaxPool3d((1,3,3),(1,2,2),(0,1,1)).
    emu: clip and weights enter the MFMA contraction as bf16; the conv output and the pooled output are stored in bf16.
    prefix / act: the sentence-level model's stem has the same shape under another name and with Swish
    (LRS/video/espnet/nets/pytorch_backend/backbones/conv3d_extractor.py:40-48)."""
    c32 = F.conv3d(videos.bfloat16().float() if emu else videos, _w16(sd[f"{prefix}.0.weight"], emu), None, stride=(1, 2, 2), padding=(2, 3, 3))
    x = _st(c32, emu)
    if keep is not None:
        keep["stem_conv"] = x
    x = batch_norm(x, sd, f"{prefix}.1", training, stats_out, x_stats=c32 if emu else None)
    if emu:
        # the HIP backward keeps g = dpool * gelu'(z) per pooled output in bf16 between its reduce and apply passes (norm_act.hip
        # k_stem_bwd_reduce_win) — as the reference's own bf16 autocast does with the GELU's input gradient
        x = _RoundGradBf16.apply(x)
    x = gelu_erf(x) if act is None else act(x)
    x = F.max_pool3d(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    return _st(x, emu)


def basic_block(x: Tensor, sd: SD, prefix: str, stride: int, training: bool, stats_out: dict | None = None, emu: bool = False,
                keep: dict | None = None, act=torch.relu) -> Tensor:
    """tcn/models/resnet.py:59-72 with relu_type='relu' (== timm BasicBlock).
    emu: each convolution output (pre-BatchNorm, statistics from its fp32 accumulators) and each BatchNorm(+ReLU) output is a bf16 tensor.
    act: the sentence-level trunk is the same block with Swish (LRS/video/espnet/nets/pytorch_backend/backbones/modules/resnet.py:90-107)."""
    c32 = F.conv2d(x, _w16(sd[f"{prefix}.conv1.weight"], emu), None, stride=stride, padding=1)
    out = batch_norm(_st(c32, emu), sd, f"{prefix}.bn1", training, stats_out, x_stats=c32 if emu else None)
    if emu:
        # the HIP backward never stores the gradient of the activation's OUTPUT: the data-gradient launch of conv2 multiplies its fp32
        # accumulators by act'(z) and stores that product in bf16 (igemm_p8.hip epilogue).  ReLU: the same numbers either way (a mask
        # commutes with rounding); Swish: one rounding instead of two
        out = _RoundValueBf16.apply(act(_RoundGradBf16.apply(out)))
    else:
        out = act(out)
    if keep is not None:
        keep[f"{prefix}.conv1.c"], keep[f"{prefix}.bn1.y"] = c32, out
    c32 = F.conv2d(out, _w16(sd[f"{prefix}.conv2.weight"], emu), None, stride=1, padding=1)
    if keep is not None:
        keep[f"{prefix}.conv2.c"] = c32
    out = batch_norm(_st(c32, emu), sd, f"{prefix}.bn2", training, stats_out, x_stats=c32 if emu else None)
    if f"{prefix}.downsample.0.weight" in sd:
        c32 = F.conv2d(x, _w16(sd[f"{prefix}.downsample.0.weight"], emu), None, stride=stride)
        res = _st(batch_norm(_st(c32, emu), sd, f"{prefix}.downsample.1", training, stats_out, x_stats=c32 if emu else None), emu)
    else:
        res = x
    z = out + res
    if keep is not None:
        keep[f"{prefix}.z"] = z           # pre-activation of the block's output
    return _st(act(z), emu)


def forward_videos(videos: Tensor, sd: SD, training: bool, stats_out: dict | None = None, keep: dict | None = None, emu: bool = False) -> Tensor:
    """lightning.py:112-119: stem -> (B*T) frames -> layer1..4 -> spatial mean -> [B,T,512]."""
    B = videos.size(0)
    h = stem3d(videos, sd, training, stats_out, keep, emu).transpose(1, 2).flatten(0, 1)
    if keep is not None:
        keep["stem_out"] = h
    for li in range(1, 5):
        for bi in range(2):
            stride = 2 if (bi == 0 and li > 1) else 1
            h = basic_block(h, sd, f"resnet.layer{li}.{bi}", stride, training, stats_out, emu, keep)
            if keep is not None:
                keep[f"resnet.layer{li}.{bi}.out"] = h
        if keep is not None:
            keep[f"layer{li}"] = h
    return _st(h.mean((2, 3)), emu).unflatten(0, (B, -1))


# --------------------------------------------------------------------------------------------
# BERT-style encoder  (lightning.py:92,152-156; HF BertModel(inputs_embeds=...))
# --------------------------------------------------------------------------------------------
class DropPlan:
    """Replays the HIP library's counter-based dropout masks (syncvsr_amd/dropout.py is the numpy twin of csrc/common.h's
    drop_keep) so a training forward WITH dropout can be compared element for element: the reference's nn.Dropout modules
    (lightning.py:45,150; HF BertEmbeddings / BertSelfAttention / BertSelfOutput / BertOutput dropouts, :92,152-156) draw from
    torch's generator, which no other implementation can reproduce — sharing the mask is the only way to pin the arithmetic
    around it.  Attention masks are indexed through the library's probability buffer (row pitch = keys padded to 8)."""

    def __init__(self, seed: int, p_hidden: float, p_attn: float, p_emb: float, sites: dict[str, int]):
        self.seed, self.p = int(seed), {"hidden": float(p_hidden), "attn": float(p_attn), "emb": float(p_emb)}
        self.sites = sites

    def __call__(self, x: Tensor, site: str, kind: str = "hidden", pitch: int | None = None) -> Tensor:
        from syncvsr_amd.dropout import keep_mask

        p = self.p[kind]
        if p <= 0.0:
            return x
        if kind == "attn":
            B, H, Lq, Lk = x.shape
            pitch = (Lk + 7) // 8 * 8
            m = keep_mask(self.seed, self.sites[site], p, B * H * Lq * pitch).reshape(B, H, Lq, pitch)[..., :Lk]
        elif pitch is not None and pitch != x.shape[-1]:      # rows stored with a padded pitch: element index = row * pitch + column
            rows = x.numel() // x.shape[-1]
            m = keep_mask(self.seed, self.sites[site], p, rows * pitch).reshape(rows, pitch)[:, : x.shape[-1]].reshape(tuple(x.shape))
        else:
            m = keep_mask(self.seed, self.sites[site], p, x.numel()).reshape(tuple(x.shape))
        return x * torch.from_numpy(m.copy()).to(x.dtype) / (1.0 - p)


def _dp(dp, x: Tensor, site: str, kind: str = "hidden", pitch: int | None = None) -> Tensor:
    return x if dp is None else dp(x, site, kind, pitch)


def bert_embeddings(x: Tensor, sd: SD, eps: float, dp=None, emu: bool = False) -> Tensor:
    S = x.size(1)
    e = _st(x + sd["encoder.embeddings.position_embeddings.weight"][:S] + sd["encoder.embeddings.token_type_embeddings.weight"][0], emu)
    return _st(_dp(dp, layer_norm(e, sd["encoder.embeddings.LayerNorm.weight"], sd["encoder.embeddings.LayerNorm.bias"], eps), "emb.out"), emu)


def bert_layer(x: Tensor, sd: SD, p: str, heads: int, eps: float, keep: dict | None = None, dp=None, i: int = 0, emu: bool = False) -> Tensor:
    B, S, D = x.shape
    dh = D // heads

    def lin(t: Tensor, name: str) -> Tensor:
        return F.linear(t, _w16(sd[f"{p}.{name}.weight"], emu), sd[f"{p}.{name}.bias"])

    q = _st(lin(x, "attention.self.query"), emu).view(B, S, heads, dh).transpose(1, 2)
    k = _st(lin(x, "attention.self.key"), emu).view(B, S, heads, dh).transpose(1, 2)
    v = _st(lin(x, "attention.self.value"), emu).view(B, S, heads, dh).transpose(1, 2)
    scores = q @ k.transpose(-1, -2) / math.sqrt(dh)
    probs = _dp(dp, _st(torch.softmax(scores, dim=-1), emu), f"enc.{i}.attn.probs", "attn")
    ctx = _st((probs @ v).transpose(1, 2).reshape(B, S, D), emu)
    if keep is not None:
        keep[f"{p}.ctx"] = ctx
    x = _st(layer_norm(_st(_dp(dp, lin(ctx, "attention.output.dense"), f"enc.{i}.attn.out"), emu) + x, sd[f"{p}.attention.output.LayerNorm.weight"],
                       sd[f"{p}.attention.output.LayerNorm.bias"], eps), emu)
    h = _st(gelu_erf(_st(lin(x, "intermediate.dense"), emu)), emu)
    x = _st(layer_norm(_st(_dp(dp, lin(h, "output.dense"), f"enc.{i}.ff.out"), emu) + x, sd[f"{p}.output.LayerNorm.weight"], sd[f"{p}.output.LayerNorm.bias"], eps), emu)
    return x


def bert_encoder(x: Tensor, sd: SD, cfg: Any, keep: dict | None = None, dp=None, emu: bool = False) -> Tensor:
    bert = cfg.model.bert
    eps = float(bert.get("layer_norm_eps", 1e-12))
    x = bert_embeddings(x, sd, eps, dp, emu)
    if keep is not None:
        keep["emb"] = x
    for i in range(int(bert.num_hidden_layers)):
        x = bert_layer(x, sd, f"encoder.encoder.layer.{i}", int(bert.num_attention_heads), eps, keep, dp, i, emu)
    return x


# --------------------------------------------------------------------------------------------
# x-transformers encoder (lightning.py:93-105,157-158) — parity UNPINNED, see the module header
# --------------------------------------------------------------------------------------------
def rms_norm(x: Tensor, g: Tensor, eps: float = 1e-8) -> Tensor:
    """x_transformers.RMSNorm: x / clamp(||x||_2 * dim^-0.5, min=eps) * g."""
    norm = torch.linalg.vector_norm(x, dim=-1, keepdim=True) * x.size(-1) ** -0.5
    return x / norm.clamp(min=eps) * g


def rotary_freqs(S: int, rot: int = 32, theta: float = 10000.0) -> Tensor:
    """x_transformers.RotaryEmbedding(dim=max(dim_head // 2, 32)): freqs[s] = cat(s * inv_freq, s * inv_freq)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, rot, 2).float() / rot))
    f = torch.einsum("i,j->ij", torch.arange(S).float(), inv_freq)
    return torch.cat((f, f), dim=-1)


def apply_rotary(t: Tensor, freqs: Tensor) -> Tensor:
    """First freqs.size(-1) dims of every head: t * cos + rotate_half(t) * sin, rotate_half((x1, x2)) = (-x2, x1)."""
    rot = freqs.size(-1)
    tl, tr = t[..., :rot], t[..., rot:]
    x1, x2 = tl[..., : rot // 2], tl[..., rot // 2:]
    tl = tl * freqs.cos() + torch.cat((-x2, x1), dim=-1) * freqs.sin()
    return torch.cat((tl, tr), dim=-1)


def xt_encoder(x: Tensor, sd: SD, cfg: Any, keep: dict | None = None, dp=None, skip: set | None = None) -> Tensor:
    """AttentionLayers.forward for Encoder(dim, depth, heads, use_rmsnorm, ff_glu, rotary_pos_emb): layers alternate attention
    ('a', even n) and feed-forward ('f', odd n) blocks, each `x = block(norm(x)) + x`; `skip` = the layer indices n that
    layer_dropout removed this step.  Padded row pitches of the HIP library are passed to `dp` so the same masks are drawn."""
    bert = cfg.model.bert
    B, S, D = x.shape
    H, dh = int(bert.heads), 64
    E, I = H * dh, 4 * D
    freqs = rotary_freqs(S)
    rot_v = bool(bert.get("rotate_value", True))
    skip = skip or set()
    for i in range(int(bert.depth)):
        a, f = f"encoder.layers.{2 * i}", f"encoder.layers.{2 * i + 1}"
        if 2 * i not in skip:
            h = rms_norm(x, sd[f"{a}.0.0.g"])
            q, k, v = (F.linear(h, sd[f"{a}.1.to_{n}.weight"]).view(B, S, H, dh).transpose(1, 2) for n in "qkv")
            q, k = apply_rotary(q, freqs), apply_rotary(k, freqs)
            if rot_v:
                v = apply_rotary(v, freqs)
            probs = _dp(dp, torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, dim=-1), f"enc.{i}.attn.probs", "attn")
            ctx = (probs @ v).transpose(1, 2).reshape(B, S, E)
            x = F.linear(ctx, sd[f"{a}.1.to_out.weight"]) + x
        if 2 * i + 1 not in skip:
            h = rms_norm(x, sd[f"{f}.0.0.g"])
            val, gate = F.linear(h, sd[f"{f}.1.ff.0.proj.weight"], sd[f"{f}.1.ff.0.proj.bias"]).chunk(2, dim=-1)
            y = _dp(dp, val * gelu_erf(gate), f"enc.{i}.ff.hidden", "hidden", pitch=(I + 63) // 64 * 64)
            x = F.linear(y, sd[f"{f}.1.ff.3.weight"], sd[f"{f}.1.ff.3.bias"]) + x
        if keep is not None:
            keep[f"xt.{i}"] = x
    if "encoder.final_norm.g" in sd:
        x = rms_norm(x, sd["encoder.final_norm.g"])
    return x


# --------------------------------------------------------------------------------------------
# heads + losses (lightning.py:161-191)
# --------------------------------------------------------------------------------------------
def cross_entropy(logits: Tensor, target: Tensor, label_smoothing: float = 0.0) -> Tensor:
    """F.cross_entropy restated: class-index or probability targets, mean reduction."""
    logp = torch.log_softmax(logits.float(), dim=-1)
    C = logits.size(-1)
    if target.dtype in (torch.long, torch.int32, torch.int64):
        nll = -logp.gather(-1, target.long().unsqueeze(-1)).squeeze(-1)
        smooth = -logp.mean(-1)
        return ((1.0 - label_smoothing) * nll + label_smoothing * smooth).mean()
    t = target.float() * (1.0 - label_smoothing) + label_smoothing / C
    return (-(t * logp).sum(-1)).mean()


def audio_dims(cfg: Any) -> tuple[int, int, int]:
    path = cfg.model.wav2vec.path
    if "vq" in path:
        return 4, 2, 320
    return 2, 2, 640


def forward(sd: SD, cfg: Any, videos: Tensor, audio_tokens: Tensor, labels: Tensor, word_mask: Tensor,
            training: bool = True, use_cutmix_metric: bool = False, keep: dict | None = None,
            stats_out: dict | None = None, dp=None, layer_skip: set | None = None, emu: bool = False) -> dict[str, Tensor]:
    """TransformerLightningModule.forward (lightning.py:133-191); dropout only through `dp` (a DropPlan replaying the
    library's masks), otherwise p = 0.  emu: bf16-storage emulation (see _RoundBf16; `type: huggingface` encoder only)."""
    A, G, V = audio_dims(cfg)
    if emu and str(cfg.model.bert.type) == "x-transformers":
        raise NotImplementedError("bf16-storage emulation covers the huggingface branch")
    feats = forward_videos(videos, sd, training, stats_out, keep, emu)                 # :136
    if keep is not None:
        keep["feats"] = feats
    if cfg.data.use_word_boundary:                                                       # :145
        feats = torch.cat((feats, word_mask.unsqueeze(-1).to(feats.dtype)), dim=-1)
    B, T, D = feats.shape
    audio_tokens = audio_tokens[:, : T * A]                                              # :148
    x = _dp(dp, torch.cat((sd["cls_token"].expand(B, -1, -1), feats), dim=1), "emb.in", "emb", pitch=(D + 63) // 64 * 64)      # :149-150
    if str(cfg.model.bert.type) == "x-transformers":
        h = xt_encoder(x, sd, cfg, keep, dp, layer_skip)                                 # :157-158
    else:
        h = bert_encoder(x, sd, cfg, keep, dp, emu)                                      # :152-156
    logits_category = F.linear(h[:, 0], _w16(sd["category_classifier.weight"], emu), sd["category_classifier.bias"]).float()
    loss_category = cross_entropy(logits_category, labels, float(cfg.train.label_smoothing))     # :161-165
    logits_audio = _st(F.linear(h[:, 1:], _w16(sd["audio_projection.weight"], emu), sd["audio_projection.bias"]).float(), emu)
    logits_audio = logits_audio.reshape(B, T, A * G, V)                                  # :168-170
    loss_audio = cross_entropy(logits_audio.reshape(-1, V), audio_tokens.flatten())      # :171
    loss_total = loss_category + loss_audio * float(cfg.optim.lambda_audio)              # :174
    hard = labels.argmax(-1) if (labels.dim() == 2 and (use_cutmix_metric or labels.dtype.is_floating_point)) else labels
    corrects = logits_category.topk(5, dim=1)[1] == hard.unsqueeze(1)                    # :177-181
    if keep is not None:
        keep["hidden"] = h
        keep["logits_category"] = logits_category
        keep["logits_audio"] = logits_audio
    return {
        "loss_total": loss_total,
        "loss_category": loss_category,
        "loss_audio": loss_audio,
        "accuracy_top1": corrects[:, 0].float().mean(),
        "accuracy_top5": corrects.float().amax(1).mean(),
    }


# --------------------------------------------------------------------------------------------
# optimiser step restated (lightning.py:216-223) — used by the CPU baseline leg
# --------------------------------------------------------------------------------------------
def clip_grad_norm(grads: list[Tensor], max_norm: float) -> Tensor:
    total = torch.sqrt(sum((g.float() ** 2).sum() for g in grads))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adamw_step(params: list[Tensor], grads: list[Tensor], m: list[Tensor], v: list[Tensor], step: int, lr: float,
               betas: tuple[float, float], eps: float, weight_decay: float) -> None:
    """torch.optim.AdamW semantics; weight decay only on ndim>=2 params (lightning.py:217-219)."""
    b1, b2 = betas
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    for p, g, mi, vi in zip(params, grads, m, v):
        wd = weight_decay if p.dim() >= 2 else 0.0
        p.mul_(1 - lr * wd)
        mi.mul_(b1).add_(g, alpha=1 - b1)
        vi.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (vi.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(mi, denom, value=-lr / bc1)


def cosine_lr(step: int, base_lr: float, warmup: int, total: int) -> float:
    """HF get_scheduler('cosine') multiplier (train config :38-41)."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    prog = (step - warmup) / max(1, total - warmup)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
