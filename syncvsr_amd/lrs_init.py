"""Parameter inventory + deterministic initialisation for the LRS hot-path model (``E2E``).

Names and shapes follow the reference state dict of ``LRS/video/espnet/nets/pytorch_backend/e2e_asr_transformer.py:43-144``
with ``transformer_input_layer: conv3d`` (``backbones/conv3d_extractor.py:18-38``, ``backbones/modules/resnet.py:45-160``),
the Conformer encoder (``transformer/encoder.py:90-247``, ``encoder_layer.py:40-73``, ``attention.py:26-36,191-203``,
``convolution.py:22-55``), the Transformer decoder (``decoder.py:59-119``, ``decoder_layer.py:31-58``), ``ctc.py:22-28``
and the cross-modal ``audio_classifier`` (``e2e_asr_transformer.py:142``).

As for LRW the same generator runs where the goldens are made and on the GPU box, so goldens never carry weights.
"""
from __future__ import annotations

import math

import torch

from .config import Config
from .init import RESNET_PLANES, Spec

LRS_ODIM = 5049          # len(unigram5000_units) + blank/unk/eos as the reference builds it (lightning.py:34)


def default_lrs_args(**kw) -> Config:
    """``config/lrs3.yaml:14-39`` (``model.visual_backbone``); dropout defaults to 0 here (the goldens and parity cases need a deterministic forward); the shipped recipe's 0.1 is what bench.py --workload lrs uses."""
    a = Config(
        audio_weight=10.0, adim=768, aheads=12, eunits=3072, elayers=12, transformer_input_layer="conv3d",
        dropout_rate=0.0, transformer_attn_dropout_rate=0.0, transformer_encoder_attn_layer_type="rel_mha",
        macaron_style=True, use_cnn_module=True, cnn_module_kernel=31, zero_triu=False, a_upsample_ratio=1,
        relu_type="swish", ddim=768, dheads=12, dunits=3072, dlayers=6, lsm_weight=0.1,
        transformer_length_normalized_loss=False, mtlalpha=0.1, ctc_type="builtin", rel_pos_type="latest", codec="vq",
    )
    for k, v in kw.items():
        a[k] = v
    return a


def lrs_audio_dims(args: Config) -> tuple[int, int, int]:
    """(A, G, V): ``e2e_asr_transformer.py:138-157``."""
    codec = str(args.codec).lower()
    if "vq" in codec:
        return 4, 2, 320
    if "wav2vec2" in codec:
        return 2, 2, 640
    raise ValueError(f"codec must name 'vq' or 'wav2vec2' for training, got {args.codec!r}")


def _frontend_block_specs(prefix: str):
    inplanes = 64
    for li, planes in enumerate(RESNET_PLANES, start=1):
        for bi in range(2):
            stride = 2 if (bi == 0 and li > 1) else 1
            down = bi == 0 and (stride != 1 or inplanes != planes)
            yield f"{prefix}.layer{li}.{bi}", inplanes, planes, stride, down
            inplanes = planes


def _mha_specs(p: str, d: int) -> list[Spec]:
    out: list[Spec] = []
    for n in ("linear_q", "linear_k", "linear_v", "linear_out"):
        out += [(f"{p}.{n}.weight", (d, d), "linear_w"), (f"{p}.{n}.bias", (d,), "linear_b")]
    return out


def _ffn_specs(p: str, d: int, u: int) -> list[Spec]:
    return [(f"{p}.w_1.weight", (u, d), "linear_w"), (f"{p}.w_1.bias", (u,), "linear_b"),
            (f"{p}.w_2.weight", (d, u), "linear_w"), (f"{p}.w_2.bias", (d,), "linear_b")]


def _ln_specs(p: str, d: int) -> list[Spec]:
    return [(f"{p}.weight", (d,), "norm_w"), (f"{p}.bias", (d,), "norm_b")]


def lrs_frontend_names(args: Config) -> tuple[str, str]:
    """(stem, trunk) module paths: `conv3d` = Conv3dResNet under encoder.frontend (encoder.py:130-131, backbones/conv3d_extractor.py);
    `conv3d-lrw` = the word-level model's stem3d + resnet18 directly under the encoder (encoder.py:132-139,248-255)."""
    if str(args.transformer_input_layer) == "conv3d-lrw":
        return "encoder.stem3d", "encoder.resnet"
    return "encoder.frontend.frontend3D", "encoder.frontend.trunk"


def lrs_param_specs(args: Config, odim: int = LRS_ODIM) -> list[Spec]:
    D, Dd = int(args.adim), int(args.ddim)
    H = int(args.aheads)
    K = int(args.cnn_module_kernel)
    A, G, V = lrs_audio_dims(args)
    stem, trunk = lrs_frontend_names(args)
    specs: list[Spec] = [
        (f"{stem}.0.weight", (64, 1, 5, 7, 7), "conv"),
        (f"{stem}.1.weight", (64,), "norm_w"),
        (f"{stem}.1.bias", (64,), "norm_b"),
    ]
    for prefix, inp, planes, stride, down in _frontend_block_specs(trunk):
        specs += [
            (f"{prefix}.conv1.weight", (planes, inp, 3, 3), "conv"),
            (f"{prefix}.bn1.weight", (planes,), "norm_w"), (f"{prefix}.bn1.bias", (planes,), "norm_b"),
            (f"{prefix}.conv2.weight", (planes, planes, 3, 3), "conv"),
            (f"{prefix}.bn2.weight", (planes,), "norm_w"), (f"{prefix}.bn2.bias", (planes,), "norm_b"),
        ]
        if down:
            specs += [
                (f"{prefix}.downsample.0.weight", (planes, inp, 1, 1), "conv"),
                (f"{prefix}.downsample.1.weight", (planes,), "norm_w"), (f"{prefix}.downsample.1.bias", (planes,), "norm_b"),
            ]
    specs += [("encoder.embed.0.weight", (D, 512), "linear_w"), ("encoder.embed.0.bias", (D,), "linear_b")]
    for i in range(int(args.elayers)):
        p = f"encoder.encoders.{i}"
        specs += _mha_specs(f"{p}.self_attn", D)
        specs += [(f"{p}.self_attn.linear_pos.weight", (D, D), "linear_w"),
                  (f"{p}.self_attn.pos_bias_u", (H, D // H), "pos_bias"),
                  (f"{p}.self_attn.pos_bias_v", (H, D // H), "pos_bias")]
        specs += _ffn_specs(f"{p}.feed_forward", D, int(args.eunits))
        specs += _ffn_specs(f"{p}.feed_forward_macaron", D, int(args.eunits))
        specs += [
            (f"{p}.conv_module.pointwise_cov1.weight", (2 * D, D, 1), "linear_w"),
            (f"{p}.conv_module.pointwise_cov1.bias", (2 * D,), "linear_b"),
            (f"{p}.conv_module.depthwise_conv.weight", (D, 1, K), "dw_w"),
            (f"{p}.conv_module.depthwise_conv.bias", (D,), "linear_b"),
            (f"{p}.conv_module.norm.weight", (D,), "norm_w"), (f"{p}.conv_module.norm.bias", (D,), "norm_b"),
            (f"{p}.conv_module.pointwise_cov2.weight", (D, D, 1), "linear_w"),
            (f"{p}.conv_module.pointwise_cov2.bias", (D,), "linear_b"),
        ]
        for n in ("norm_ff", "norm_mha", "norm_ff_macaron", "norm_conv", "norm_final"):
            specs += _ln_specs(f"{p}.{n}", D)
    specs += _ln_specs("encoder.after_norm", D)
    if D != Dd:            # e2e_asr_transformer.py:93-95
        specs += [("proj_decoder.weight", (Dd, D), "linear_w"), ("proj_decoder.bias", (Dd,), "linear_b")]
    specs += [("decoder.embed.0.weight", (odim, Dd), "emb")]
    for i in range(int(args.dlayers)):
        p = f"decoder.decoders.{i}"
        specs += _mha_specs(f"{p}.self_attn", Dd) + _mha_specs(f"{p}.src_attn", Dd)
        specs += _ffn_specs(f"{p}.feed_forward", Dd, int(args.dunits))
        for n in ("norm1", "norm2", "norm3"):
            specs += _ln_specs(f"{p}.{n}", Dd)
    specs += _ln_specs("decoder.after_norm", Dd)
    specs += [("decoder.output_layer.weight", (odim, Dd), "linear_w"), ("decoder.output_layer.bias", (odim,), "linear_b")]
    if float(args.mtlalpha) > 0.0:             # `self.ctc` exists only then (e2e_asr_transformer.py:127-132)
        specs += [("ctc.ctc_lo.weight", (odim, D), "linear_w"), ("ctc.ctc_lo.bias", (odim,), "linear_b")]
    specs += [("audio_classifier.weight", (A * G * V, D), "linear_w"), ("audio_classifier.bias", (A * G * V,), "linear_b")]
    return specs


def lrs_buffer_specs(args: Config, odim: int = LRS_ODIM) -> list[Spec]:
    """BatchNorm running statistics (front-end BN3d/BN2d and each ``conv_module.norm`` BN1d)."""
    out: list[Spec] = []
    for name, shape, kind in lrs_param_specs(args, odim):
        is_bn = kind == "norm_w" and (".bn" in name or "frontend3D.1" in name or "stem3d.1" in name or "downsample.1" in name
                                      or "conv_module.norm" in name)
        if is_bn:
            base = name[: -len(".weight")]
            out += [(f"{base}.running_mean", shape, "bn_mean"), (f"{base}.running_var", shape, "bn_var"),
                    (f"{base}.num_batches_tracked", (), "bn_count")]
    return out


def lrs_init_state_dict(args: Config, odim: int = LRS_ODIM, seed: int = 0, perturb_norm: bool = False) -> dict[str, torch.Tensor]:
    """Deterministic fp32 CPU state dict.  Scales follow torch defaults for the layer kinds the reference
    instantiates (kaiming-uniform Linear/Conv, N(0,1) Embedding, xavier-uniform ``pos_bias_*`` — ``attention.py:201-202``);
    the ResNet convs use the He-normal fan-out rule of ``backbones/modules/resnet.py:146-150``."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    sd: dict[str, torch.Tensor] = {}
    last_fan_in = 1
    for name, shape, kind in lrs_param_specs(args, odim):
        if kind == "conv":
            if "frontend3D" in name or "stem3d" in name:
                bound = 1.0 / math.sqrt(math.prod(shape[1:]))
                t = (torch.rand(shape, generator=g) * 2 - 1) * bound
            else:
                t = torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[0] * math.prod(shape[2:])))
        elif kind == "norm_w":
            t = torch.ones(shape)
            if perturb_norm:
                t = t + 0.2 * (torch.rand(shape, generator=g) - 0.5)
        elif kind == "norm_b":
            t = torch.zeros(shape)
            if perturb_norm:
                t = 0.2 * (torch.rand(shape, generator=g) - 0.5)
        elif kind in ("linear_w", "dw_w"):
            last_fan_in = math.prod(shape[1:])
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(last_fan_in)
        elif kind == "linear_b":
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(last_fan_in)
        elif kind == "pos_bias":
            bound = math.sqrt(6.0 / (shape[0] + shape[1]))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind == "emb":
            t = torch.randn(shape, generator=g)
        else:  # pragma: no cover
            raise AssertionError(kind)
        sd[name] = t.float().contiguous()
    for name, shape, kind in lrs_buffer_specs(args, odim):
        sd[name] = torch.zeros(shape) if kind == "bn_mean" else torch.ones(shape) if kind == "bn_var" else torch.zeros((), dtype=torch.long)
    return sd


def lrs_synthetic_batch(args: Config, batch: int, t_max: int, odim: int = LRS_ODIM, size: int = 88, seed: int = 1234,
                        min_len_frac: float = 0.5, label_len: tuple[int, int] = (5, 40), lengths=None):
    """SURVEY §8(d) LRS inputs: x [B,T,1,H,W] N(0,1) zero-padded past each length; lengths in [min_len_frac·T, T]
    (first clip full length) unless `lengths` (<= t_max, e.g. one step of lrs_data.LengthBucketBatchSampler) is given;
    audio tokens [B, A·T, G]; targets in [1, odim-2] with length U{lo..hi}, padded with -1."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    A, G, V = lrs_audio_dims(args)
    if lengths is not None:
        lengths = torch.as_tensor(lengths, dtype=torch.long).clone()
        assert lengths.numel() == batch and int(lengths.max()) <= t_max
    else:
        lengths = torch.randint(max(1, int(t_max * min_len_frac)), t_max + 1, (batch,), generator=g)
        lengths[0] = t_max
    x = torch.randn(batch, t_max, 1, size, size, generator=g)
    for b in range(batch):
        x[b, int(lengths[b]):] = 0.0
    tokens = torch.randint(0, V, (batch, t_max * A, G), generator=g)
    lo, hi = label_len
    olens = torch.randint(lo, hi + 1, (batch,), generator=g)
    # CTC needs olen (+repeats) <= input length; keep targets comfortably shorter than the shortest clip
    olens = torch.minimum(olens, (lengths // 2).clamp(min=1))
    L = int(olens.max())
    label = torch.full((batch, 1, L), -1, dtype=torch.long)
    for b in range(batch):
        label[b, 0, : int(olens[b])] = torch.randint(1, odim - 1, (int(olens[b]),), generator=g)
    return x, lengths, tokens, label
