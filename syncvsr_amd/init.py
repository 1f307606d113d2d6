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


def param_specs(cfg: Config) -> list[Spec]:
    D = hidden_dim(cfg)
    bert = cfg.model.bert
    H = int(bert.hidden_size) + (D - int(bert.dim))
    assert H == D, "encoder width must equal feature width (+1 with word boundary)"
    I = int(bert.intermediate_size)
    L = int(bert.num_hidden_layers)
    _, A, G, V = audio_codec_dims(cfg.model.wav2vec.path)
    specs: list[Spec] = [
        ("cls_token", (1, 1, D), "cls"),
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
    specs += [
        ("audio_projection.weight", (A * G * V, D), "linear_w"),
        ("audio_projection.bias", (A * G * V,), "linear_b"),
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
