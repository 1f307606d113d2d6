"""Golden-vector generator for the LRS hot path (``E2E.forward``).  RUNS ONLY IN THE BUILD CONTAINER.

Imports the reference implementation itself (``/root/reference/LRS/video/espnet/nets/pytorch_backend/e2e_asr_transformer.py``)
with the import stub of SURVEY.md Appendix C (``timm`` is imported at module level by ``transformer/encoder.py`` but unused
for ``transformer_input_layer: conv3d``), builds ``E2E`` with ``codec=None`` (skips the wav2vec download,
``e2e_asr_transformer.py:135-136``) and then attaches exactly what ``:138-144`` would have: ``audio_classifier``,
``codec``, ``audio_alignment``, ``audio_vocab_size``, ``audio_weight``; ``forward_audios`` is overridden to return the
supplied token tensor (SURVEY §8(b) audio-input deviation).  Weights come from ``syncvsr_amd.lrs_init`` and are never stored.
Nothing of the reference's source is copied — only numbers it computes.

    python tests/golden/make_golden_lrs.py [case ...]
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch
import torch.nn as nn
import transformers  # noqa: F401  (before the stubs, SURVEY App. C pitfall 1)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/LRS/video"

from golden_cases import LRS_CASES, build_lrs_case, sample_idx  # noqa: E402
from syncvsr_amd.lrs_init import lrs_audio_dims, lrs_param_specs  # noqa: E402


def import_reference():
    m = types.ModuleType("timm")
    m.__spec__ = importlib.machinery.ModuleSpec("timm", None)
    # `transformer_input_layer: conv3d-lrw` builds timm.create_model("resnet18") (encoder.py:139); timm is not installed, so — as in
    # make_golden_lrw.py — the topology-identical in-tree ResNet18 of the reference's LRW tree stands in for it
    sys.path.insert(0, "/root/reference/LRW/video/src")
    from tcn.models.resnet import BasicBlock, ResNet

    m.create_model = lambda name, **k: ResNet(BasicBlock, [2, 2, 2, 2], relu_type="relu")
    sys.modules["timm"] = m
    sys.path.insert(0, REF)
    from espnet.nets.pytorch_backend.e2e_asr_transformer import E2E

    return E2E


def run_case(E2E, name: str) -> dict[str, np.ndarray]:
    args, odim, sd, batch, training, _ = build_lrs_case(name, load_golden=False)
    x, lengths, tokens, label = batch
    ns = Namespace(**{k: v for k, v in args.items() if k != "codec"}, codec=None)
    torch.manual_seed(0)
    model = E2E(odim, ns)
    A, G, V = lrs_audio_dims(args)
    model.audio_classifier = nn.Linear(int(args.adim), A * G * V)
    model.codec = "vq" if A == 4 else "wav2vec2"
    model.audio_alignment, model.audio_vocab_size, model.audio_weight = A, V, float(args.audio_weight)
    model.forward_audios = lambda wav: tokens
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    model.train(training)
    model.double()            # see make_golden_lrw.py: fp64 removes the reference's own fp32 BN-backward noise from the pins
    x = x.double()

    keep: dict[str, torch.Tensor] = {}

    def hook(key):
        def f(_m, _i, o):
            while isinstance(o, tuple):
                o = o[0]
            keep[key] = o.detach()
        return f

    enc = model.encoder
    if hasattr(enc, "stem3d"):                 # conv3d-lrw: the features come out of a method (forward_videos), not a module
        enc.embed.register_forward_pre_hook(lambda _m, i: keep.__setitem__("feats", i[0].detach()))
        enc.stem3d.register_forward_hook(hook("stem_out5d"))
    else:
        enc.frontend.register_forward_hook(hook("feats"))
        enc.frontend.frontend3D.register_forward_hook(hook("stem_out5d"))
    for i, layer in enumerate(enc.encoders):
        layer.register_forward_hook(hook(f"enc{i}"))
    enc.encoders[0].self_attn.register_forward_hook(hook("enc0.self_attn"))
    enc.encoders[0].conv_module.register_forward_hook(hook("enc0.conv_module"))
    enc.register_forward_hook(hook("enc_out"))
    for i, layer in enumerate(model.decoder.decoders):
        layer.register_forward_hook(hook(f"dec{i}"))
    model.decoder.register_forward_hook(hook("pred"))
    if model.ctc is not None:
        model.ctc.ctc_lo.register_forward_hook(hook("ctc_logits"))
    model.audio_classifier.register_forward_hook(hook("logits_audio"))

    dummy_audio = torch.zeros(x.size(0), 1, 16)
    # E2E.forward casts to fp32 before the CTC / attention losses (`x.float()`, `pred_pad.float()`, :207,:215); for the
    # fp64 evaluation those casts are made no-ops while the reference's forward runs.
    orig_float = torch.Tensor.float
    torch.Tensor.float = lambda self: self
    try:
        loss, loss_ctc, loss_att, loss_audio, acc = model(x, lengths, dummy_audio, label)
    finally:
        torch.Tensor.float = orig_float
    model.zero_grad()
    if training:
        loss.backward()
    res: dict[str, np.ndarray] = {"loss": np.float64(loss.item()), "loss_ctc": np.float64(float(loss_ctc)),
                                  "loss_att": np.float64(loss_att.item()), "loss_audio": np.float64(loss_audio.item()),
                                  "acc": np.float64(acc)}
    keep["stem_out"] = keep.pop("stem_out5d").transpose(1, 2).flatten(0, 1)
    small = not name.startswith("lrs_full")
    for k, v in keep.items():
        v = v.double()
        res[f"sum.{k}"] = np.float64(v.sum().item())
        res[f"abssum.{k}"] = np.float64(v.abs().sum().item())
        flat = v.flatten()
        idx = sample_idx(flat.numel())
        res[f"sample.{k}"] = flat[idx].numpy()
        if small and v.numel() <= 8000:
            res[f"full.{k}"] = v.float().numpy()
    if training:
        names = [n for n, _, _ in lrs_param_specs(args, odim)]
        params = dict(model.named_parameters())
        assert set(names) == set(params), set(names) ^ set(params)
        res["grad_names"] = np.array(names)
        res["grad_norms"] = np.array([params[n].grad.double().norm().item() for n in names])
        res["grad_heads"] = np.stack([
            np.pad(params[n].grad.flatten()[:32].double().numpy(), (0, max(0, 32 - params[n].numel()))) for n in names])
        if small:
            for n in names:
                if params[n].numel() <= 4096 and (".0." in n or "after_norm" in n or "ctc" in n):
                    res[f"grad.{n}"] = params[n].grad.float().numpy()
        for n, mod in model.named_modules():
            if isinstance(mod, nn.modules.batchnorm._BatchNorm) and (n.endswith("frontend3D.1") or n.endswith("stem3d.1") or "layer2.0.downsample.1" in n
                                                                     or n.endswith("encoders.0.conv_module.norm")):
                res[f"buf.{n}.running_mean"] = mod.running_mean.numpy().copy()
                res[f"buf.{n}.running_var"] = mod.running_var.numpy().copy()
                res[f"buf.{n}.num_batches_tracked"] = np.int64(mod.num_batches_tracked.item())
    return res


def main() -> None:
    E2E = import_reference()
    torch.set_num_threads(8)
    for name in (sys.argv[1:] or list(LRS_CASES)):
        res = run_case(E2E, name)
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **res)
        print(f"{name}: loss={float(res['loss']):.6f} ctc={float(res['loss_ctc']):.6f} att={float(res['loss_att']):.6f} "
              f"audio={float(res['loss_audio']):.6f} acc={float(res['acc']):.4f} -> {path} ({os.path.getsize(path)/1024:.0f} KB)")


if __name__ == "__main__":
    main()
