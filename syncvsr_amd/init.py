"""Parameter inventory + deterministic initialisation for the LRW hot-path model.

Names and shapes follow the reference state-dict (``LRW/video/src/lightning.py:49-55,82,92,107-108``;
HF ``BertModel`` sub-tree under ``encoder.``; timm/in-tree ResNet18 ``layer1-4`` under ``resnet.``,
``LRW/video/src/tcn/models/resnet.py:75-130``).  Parameters the reference allocates but its forward
never touches (BERT word embeddings / pooler, timm ``conv1/bn1/fc``) are deliberately not allocated
(SURVEY §2.1 note on ``find_unused_parameters``).

The same generator runs in the build container (to feed the imported reference when goldens are
made) and on the GPU box (to re-create identical weights from the seed), so goldens never carry weights.
"""
from __future__ import annotations

import math
from typing import Iterator

import torch

from .config import Config, audio_codec_dims

# (name, shape, kind)
Spec = tuple[str, tuple[int, ...], str]

RESNET_PLANES = (64, 128, 256, 512)


def hidden_dim(cfg: Config) -> int:
    return int(cfg.model.bert.dim) + (1 if cfg.data.use_word_boundary else 0)


def resnet_block_specs() -> Iterator[tuple[str, int, int, int, bool]]:
    """(prefix, inplanes, planes, stride, has_downsample) for the 8 BasicBlocks."""
    inplanes = 64
    for li, planes in enumerate(RESNET_PLANES, start=1):
        for bi in range(2):
            stride = 2 if (bi == 0 and li > 1) else 1
            down = bi == 0 and (stride != 1 or inplanes != planes)
            yield f"resnet.layer{li}.{bi}", inplanes, planes, stride, down
            inplanes = planes


def pad64(n: int) -> int:
    return (int(n) + 63) // 64 * 64


def encoder_type(cfg: Config) -> str:
    return str(cfg.model.bert.type)


def _frontend_specs() -> list[Spec]:
    specs: list[Spec] = [
        ("stem3d.0.weight", (64, 1, 5, 7, 7), "conv"),
        ("stem3d.1.weight", (64,), "norm_w"),
        ("stem3d.1.bias", (64,), "norm_b"),
    ]
    for prefix, inp, planes, stride, down in resnet_block_specs():
        specs += [
            (f"{prefix}.conv1.weight", (planes, inp, 3, 3), "conv"),
            (f"{prefix}.bn1.weight", (planes,), "norm_w"),
            (f"{prefix}.bn1.bias", (planes,), "norm_b"),
            (f"{prefix}.conv2.weight", (planes, planes, 3, 3), "conv"),
            (f"{prefix}.bn2.weight", (planes,), "norm_w"),
            (f"{prefix}.bn2.bias", (planes,), "norm_b"),
        ]
        if down:
            specs += [
                (f"{prefix}.downsample.0.weight", (planes, inp, 1, 1), "conv"),
                (f"{prefix}.downsample.1.weight", (planes,), "norm_w"),
                (f"{prefix}.downsample.1.bias", (planes,), "norm_b"),
            ]
    return specs


def xt_dims(cfg: Config) -> tuple[int, int, int, int]:
    """(D, attention inner width E = heads * 64, feed-forward inner width I = 4 D, depth) of the x-transformers encoder
    (x_transformers defaults: dim_head 64, ff mult 4)."""
    D = hidden_dim(cfg)
    return D, int(cfg.model.bert.heads) * 64, 4 * D, int(cfg.model.bert.depth)


def _xt_encoder_specs(cfg: Config) -> list[Spec]:
    """State-dict names of x_transformers.Encoder as `AttentionLayers` registers them: `layers.{n}` = ModuleList([norms, block,
    residual]) with n alternating attention ('a') and feed-forward ('f'); norms[0] = the pre-branch RMSNorm (`g`); Attention owns
    bias-free to_q / to_k / to_v / to_out; FeedForward.ff = Sequential(GLU(proj), Identity, Dropout, Linear).  (Recalled from the
    package, which is not vendored in the reference tree: parity unpinned.)"""
    D, E, I, depth = xt_dims(cfg)
    specs: list[Spec] = []
    for i in range(depth):
        a, f = f"encoder.layers.{2 * i}", f"encoder.layers.{2 * i + 1}"
        specs += [
            (f"{a}.0.0.g", (D,), "norm_w"),
            (f"{a}.1.to_q.weight", (E, D), "linear_w"),
            (f"{a}.1.to_k.weight", (E, D), "linear_w"),
            (f"{a}.1.to_v.weight", (E, D), "linear_w"),
            (f"{a}.1.to_out.weight", (D, E), "linear_w"),
            (f"{f}.0.0.g", (D,), "norm_w"),
            (f"{f}.1.ff.0.proj.weight", (2 * I, D), "linear_w"),
            (f"{f}.1.ff.0.proj.bias", (2 * I,), f"ubias{D}"),
            (f"{f}.1.ff.3.weight", (D, I), "linear_w"),
            (f"{f}.1.ff.3.bias", (D,), f"ubias{I}"),
        ]
    if bool(cfg.model.bert.get("final_norm", False)):
        specs.append(("encoder.final_norm.g", (D,), "norm_w"))
    return specs


def param_specs(cfg: Config) -> list[Spec]:
    D = hidden_dim(cfg)
    bert = cfg.model.bert
    _, A, G, V = audio_codec_dims(cfg.model.wav2vec.path)
    specs: list[Spec] = [("cls_token", (1, 1, D), "cls")] + _frontend_specs()
    specs += [
        ("audio_projection.weight", (A * G * V, D), "linear_w"),
        ("audio_projection.bias", (A * G * V,), "linear_b"),
    ]
    if encoder_type(cfg) == "x-transformers":
        specs += _xt_encoder_specs(cfg)
    else:
        H = int(bert.hidden_size) + (D - int(bert.dim))
        assert H == D, "encoder width must equal feature width (+1 with word boundary)"
        I = int(bert.intermediate_size)
        L = int(bert.num_hidden_layers)
        specs += [
            ("encoder.embeddings.position_embeddings.weight", (int(bert.max_position_embeddings), D), "emb"),
            ("encoder.embeddings.token_type_embeddings.weight", (int(bert.type_vocab_size), D), "emb"),
            ("encoder.embeddings.LayerNorm.weight", (D,), "norm_w"),
            ("encoder.embeddings.LayerNorm.bias", (D,), "norm_b"),
        ]
        for i in range(L):
            p = f"encoder.encoder.layer.{i}"
            for nm in ("query", "key", "value"):
                specs += [(f"{p}.attention.self.{nm}.weight", (D, D), "bert_w"), (f"{p}.attention.self.{nm}.bias", (D,), "bert_b")]
            specs += [
                (f"{p}.attention.output.dense.weight", (D, D), "bert_w"),
                (f"{p}.attention.output.dense.bias", (D,), "bert_b"),
                (f"{p}.attention.output.LayerNorm.weight", (D,), "norm_w"),
                (f"{p}.attention.output.LayerNorm.bias", (D,), "norm_b"),
                (f"{p}.intermediate.dense.weight", (I, D), "bert_w"),
                (f"{p}.intermediate.dense.bias", (I,), "bert_b"),
                (f"{p}.output.dense.weight", (D, I), "bert_w"),
                (f"{p}.output.dense.bias", (D,), "bert_b"),
                (f"{p}.output.LayerNorm.weight", (D,), "norm_w"),
                (f"{p}.output.LayerNorm.bias", (D,), "norm_b"),
            ]
    specs += [
        ("category_classifier.weight", (int(bert.num_labels), D), "linear_w"),
        ("category_classifier.bias", (int(bert.num_labels),), "linear_b"),
    ]
    return specs


def phys_shape(cfg: Config, name: str, shape: tuple[int, ...]) -> tuple[int, ...]:
    """Shape of a tensor's storage in the flat parameter buffer.  Encoder / head dimensions that are not multiples of 64 (513 with
    the word boundary, 2052 and 4104 in the feed-forward) are stored in rows / columns padded with zeros to the next multiple of
    64, so the contraction kernels see aligned operands; the nn.Parameter is the [:logical] view.  Pads stay zero under training:
    their gradients are exactly zero (zero activations times anything) and AdamW maps (p, g) = (0, 0) to 0."""
    if encoder_type(cfg) != "x-transformers" or len(shape) > 3:
        return tuple(shape)
    if not (name.startswith("encoder.") or name == "cls_token" or name.startswith(("audio_projection", "category_classifier"))):
        return tuple(shape)
    D, E, I, _ = xt_dims(cfg)
    wide = {D: pad64(D), I: pad64(I), 2 * I: pad64(2 * I)}
    if name.startswith(("audio_projection", "category_classifier")):     # output features of the heads stay as they are
        return tuple(shape[:1]) + tuple(wide.get(d, d) for d in shape[1:])
    return tuple(wide.get(d, d) for d in shape)


def buffer_specs(cfg: Config) -> list[Spec]:
    """BatchNorm running statistics (``running_mean``, ``running_var``, ``num_batches_tracked``)."""
    out: list[Spec] = []
    for name, shape, kind in param_specs(cfg):
        if kind == "norm_w" and ("bn" in name or name.startswith("stem3d.1") or "downsample.1" in name):
            base = name[: -len(".weight")]
            out += [
                (f"{base}.running_mean", shape, "bn_mean"),
                (f"{base}.running_var", shape, "bn_var"),
                (f"{base}.num_batches_tracked", (), "bn_count"),
            ]
    return out


def init_state_dict(cfg: Config, seed: int = 0, perturb_norm: bool = False) -> dict[str, torch.Tensor]:
    """Deterministic fp32 CPU state dict (parameters + BN buffers).

    Scales follow the libraries the reference instantiates: kaiming-normal(fan_out) convs and
    unit/zero BN (timm / ``tcn/models/resnet.py:91-97``), torch-default uniform Linear heads
    (``lightning.py:82,107``), N(0, 0.02) BERT linears/embeddings (HF ``initializer_range``),
    N(0,1) ``cls_token`` with the word-boundary channel zeroed (``lightning.py:108-110``).
    ``perturb_norm`` jitters every norm gamma/beta so parity tests are sensitive to them.
    """
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    sd: dict[str, torch.Tensor] = {}
    for name, shape, kind in param_specs(cfg):
        if kind == "conv":
            fan_out = shape[0] * math.prod(shape[2:])
            if name.startswith("stem3d"):
                fan_in = math.prod(shape[1:])
                bound = 1.0 / math.sqrt(fan_in)  # torch Conv3d default: kaiming_uniform(a=sqrt(5))
                t = (torch.rand(shape, generator=g) * 2 - 1) * bound
            else:
                t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
        elif kind == "norm_w":
            t = torch.ones(shape)
            if perturb_norm:
                t = t + 0.2 * (torch.rand(shape, generator=g) - 0.5)
        elif kind == "norm_b":
            t = torch.zeros(shape)
            if perturb_norm:
                t = 0.2 * (torch.rand(shape, generator=g) - 0.5)
        elif kind == "linear_w":
            bound = 1.0 / math.sqrt(shape[1])
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind == "linear_b":
            # fan_in of the matching weight = hidden dim
            bound = 1.0 / math.sqrt(hidden_dim(cfg))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind.startswith("ubias"):          # torch Linear bias: U(+-1/sqrt(fan_in)), fan_in carried in the kind
            bound = 1.0 / math.sqrt(int(kind[5:]))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind in ("bert_w", "emb"):
            t = torch.randn(shape, generator=g) * 0.02
        elif kind == "bert_b":
            t = torch.zeros(shape)
            if perturb_norm:
                t = 0.02 * torch.randn(shape, generator=g)
        elif kind == "cls":
            t = torch.randn(shape, generator=g)
            if cfg.data.use_word_boundary:
                t[0, 0, -1] = 0.0
        else:  # pragma: no cover
            raise AssertionError(kind)
        sd[name] = t.float().contiguous()
    for name, shape, kind in buffer_specs(cfg):
        if kind == "bn_mean":
            sd[name] = torch.zeros(shape)
        elif kind == "bn_var":
            sd[name] = torch.ones(shape)
        else:
            sd[name] = torch.zeros((), dtype=torch.long)
    return sd


def synthetic_batch(cfg: Config, batch: int, frames: int = 29, size: int = 88, seed: int = 1234,
                    soft_labels: bool = False) -> tuple[torch.Tensor, ...]:
    """SURVEY §8(d) synthetic inputs: N(0,1) clips, uniform audio tokens/labels, zero word mask."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    _, A, G, V = audio_codec_dims(cfg.model.wav2vec.path)
    videos = torch.randn(batch, 1, frames, size, size, generator=g)
    audio_tokens = torch.randint(0, V, (batch, frames * A, G), generator=g)
    n_cls = int(cfg.model.bert.num_labels)
    labels = torch.randint(0, n_cls, (batch,), generator=g)
    if soft_labels:  # CutMix-style probability targets (augment.py:27-79 mixes two one-hots)
        other = torch.randint(0, n_cls, (batch,), generator=g)
        lam = torch.rand(batch, generator=g)
        soft = torch.zeros(batch, n_cls)
        soft[torch.arange(batch), labels] += lam
        soft[torch.arange(batch), other] += 1 - lam
        labels = soft
    if cfg.data.use_word_boundary:
        word_mask = torch.zeros(batch, frames)
        for b in range(batch):
            n = int(torch.randint(5, min(20, frames) + 1, (1,), generator=g))
            s = (frames - n) // 2
            word_mask[b, s : s + n] = 1.0
    else:
        word_mask = torch.zeros(batch, 1)
    return videos, audio_tokens, labels, word_mask
