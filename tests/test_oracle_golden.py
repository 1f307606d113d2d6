"""Pins oracle/lrw_oracle.py against vectors produced by the reference itself (tests/golden/*.npz)."""
import numpy as np
import pytest
import torch

from golden_cases import CASES, HEAVY_CASES, build_case, sample_idx
from oracle import lrw_oracle as O

SCALARS = ("loss_total", "loss_category", "loss_audio", "accuracy_top1", "accuracy_top5")


def _run_oracle(name, dtype=torch.float64):
    cfg, sd, batch, training, gold = build_case(name)
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    batch = tuple(t.to(dtype) if t.is_floating_point() else t for t in batch)
    keep, stats = {}, {}
    out = O.forward(sd, cfg, *batch, training=training, keep=keep, stats_out=stats,
                    use_cutmix_metric=bool(cfg.train.use_cutmix) and training)
    if training:
        out["loss_total"].backward()
    return cfg, sd, out, keep, stats, gold, training


@pytest.mark.parametrize("name", [n for n in CASES if n not in HEAVY_CASES])
def test_oracle_matches_reference(name):
    torch.set_num_threads(8)
    cfg, sd, out, keep, stats, gold, training = _run_oracle(name)
    for k in SCALARS:
        assert abs(out[k].item() - float(gold[k])) <= 1e-6 * max(1.0, abs(float(gold[k]))), k
    for key in ("stem_conv", "stem_out", "layer1", "layer4", "emb", "hidden", "logits_category", "logits_audio"):
        v = keep[key].detach().float()
        if key == "stem_out":  # golden hooks the nn.Sequential output [B,64,T,h,w]; oracle keeps the frame-major view
            B = gold["full.logits_category"].shape[0] if "full.logits_category" in gold else 2
            v = v.unflatten(0, (-1, keep["feats"].shape[1])).transpose(1, 2)
        if key == "logits_audio":
            v = v.reshape(v.shape[0], v.shape[1], -1)
        ref_sum, ref_abs = float(gold[f"sum.{key}"]), float(gold[f"abssum.{key}"])
        assert abs(v.double().abs().sum().item() - ref_abs) <= 1e-6 * ref_abs + 1e-7, key
        assert abs(v.double().sum().item() - ref_sum) <= 1e-6 * ref_abs + 1e-7, key
        flat = v.flatten()
        idx = sample_idx(flat.numel())
        np.testing.assert_allclose(flat[idx].double().numpy(), gold[f"sample.{key}"], rtol=1e-5, atol=1e-6, err_msg=key)
        if f"full.{key}" in gold:
            np.testing.assert_allclose(v.numpy(), gold[f"full.{key}"], rtol=1e-5, atol=1e-6, err_msg=key)
    if training:
        names = [str(n) for n in gold["grad_names"]]
        norms = np.array([sd[n].grad.double().norm().item() for n in names])
        np.testing.assert_allclose(norms, gold["grad_norms"], rtol=1e-5, atol=1e-9)
        heads = np.stack([np.pad(sd[n].grad.flatten()[:32].numpy(), (0, max(0, 32 - sd[n].numel()))) for n in names])
        scale = gold["grad_norms"][:, None] / np.sqrt(np.array([sd[n].numel() for n in names]))[:, None]
        assert np.all(np.abs(heads - gold["grad_heads"]) <= 1e-4 * scale + 1e-9)  # key.bias grads are analytically 0
        for k in gold.files:
            if k.startswith("grad.") and k not in ("grad_names", "grad_norms", "grad_heads"):
                g = sd[k[5:]].grad.numpy()
                np.testing.assert_allclose(g, gold[k], rtol=1e-4, atol=1e-6 * max(1e-3, float(np.abs(gold[k]).max())), err_msg=k)
            if k.startswith("buf."):
                got = stats[k[4:]]
                np.testing.assert_allclose(np.asarray(got, dtype=np.float64), np.asarray(gold[k], dtype=np.float64), rtol=1e-5, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("name", ["lrw_full_b2", "lrw_tiny"])
def test_oracle_fp32_losses(name):
    """The fp32 oracle (what the CPU baseline leg times and what GPU parity is judged against) agrees with
    the fp64 reference goldens on every returned scalar to 2e-5 relative."""
    cfg, sd, out, keep, stats, gold, training = _run_oracle(name, torch.float32)
    for k in SCALARS:
        assert abs(out[k].item() - float(gold[k])) <= 2e-5 * max(1.0, abs(float(gold[k]))), k


# ---------------------------------------------------------------------------------------------
# LRS (E2E.forward) — oracle/lrs_oracle.py against vectors produced by the reference's own E2E
# ---------------------------------------------------------------------------------------------
from golden_cases import LRS_CASES, build_lrs_case  # noqa: E402
from oracle import lrs_oracle as OS  # noqa: E402

LRS_SCALARS = ("loss", "loss_ctc", "loss_att", "loss_audio")


def _run_lrs_oracle(name, dtype=torch.float64):
    args, odim, sd, batch, training, gold = build_lrs_case(name)
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    x, lengths, tokens, label = batch
    keep, stats = {}, {}
    out = OS.forward(sd, args, x.to(dtype), lengths, tokens, label, training=training, stats_out=stats, keep=keep)
    if training:
        out["loss"].backward()
    return args, sd, out, keep, stats, gold, training


@pytest.mark.parametrize("name", [n for n in LRS_CASES if n not in HEAVY_CASES])
def test_lrs_oracle_matches_reference(name):
    torch.set_num_threads(8)
    args, sd, out, keep, stats, gold, training = _run_lrs_oracle(name)
    for k in LRS_SCALARS:
        assert abs(out[k].item() - float(gold[k])) <= 1e-6 * max(1.0, abs(float(gold[k]))), k
    assert abs(out["acc"] - float(gold["acc"])) < 1e-9
    keys = ["stem_out", "feats", "enc0", "enc_out", "pred"] + [k for k in keep if k.startswith("dec")]
    for key in keys:
        v = keep[key].detach().double()
        ref_sum, ref_abs = float(gold[f"sum.{key}"]), float(gold[f"abssum.{key}"])
        assert abs(v.abs().sum().item() - ref_abs) <= 1e-6 * ref_abs + 1e-7, key
        assert abs(v.sum().item() - ref_sum) <= 1e-6 * ref_abs + 1e-7, key
        flat = v.flatten()
        idx = sample_idx(flat.numel())
        np.testing.assert_allclose(flat[idx].numpy(), gold[f"sample.{key}"], rtol=1e-5, atol=1e-6, err_msg=key)
        if f"full.{key}" in gold:
            np.testing.assert_allclose(v.numpy(), gold[f"full.{key}"], rtol=1e-5, atol=1e-6, err_msg=key)
    if training:
        names = [str(n) for n in gold["grad_names"]]
        norms = np.array([sd[n].grad.double().norm().item() for n in names])
        # biases feeding a BatchNorm / the softmax-invariant key bias have analytically-zero gradients: compare on the layer's scale
        floor = 1e-9 * max(1.0, float(gold["grad_norms"].max()))
        assert np.all(np.abs(norms - gold["grad_norms"]) <= 1e-5 * gold["grad_norms"] + floor)
        heads = np.stack([np.pad(sd[n].grad.flatten()[:32].numpy(), (0, max(0, 32 - sd[n].numel()))) for n in names])
        scale = gold["grad_norms"][:, None] / np.sqrt(np.array([sd[n].numel() for n in names]))[:, None]
        assert np.all(np.abs(heads - gold["grad_heads"]) <= 1e-4 * scale + floor)
        for k in gold.files:
            if k.startswith("grad.") and k not in ("grad_names", "grad_norms", "grad_heads"):
                g = sd[k[5:]].grad.numpy()
                np.testing.assert_allclose(g, gold[k], rtol=1e-4, atol=floor + 1e-6 * max(1e-3, float(np.abs(gold[k]).max())), err_msg=k)
            if k.startswith("buf."):
                np.testing.assert_allclose(np.asarray(stats[k[4:]], dtype=np.float64), np.asarray(gold[k], dtype=np.float64),
                                           rtol=1e-5, atol=1e-6, err_msg=k)


def test_lrs_oracle_fp32_losses():
    args, sd, out, keep, stats, gold, training = _run_lrs_oracle("lrs_tiny", torch.float32)
    for k in LRS_SCALARS:
        assert abs(out[k].item() - float(gold[k])) <= 5e-5 * max(1.0, abs(float(gold[k]))), k


def test_ctc_recursion_matches_builtin():
    """The explicit alpha recursion in the oracle equals torch's builtin CTC (which the reference calls, ctc.py:44-47)."""
    g = torch.Generator().manual_seed(3)
    for T, V, y in ((7, 11, [3, 3, 5]), (6, 5, [1]), (9, 6, [2, 4, 4, 1])):
        lp = torch.randn(T, V, generator=g, dtype=torch.float64).log_softmax(-1)
        yt = torch.tensor(y)
        ref = torch.nn.functional.ctc_loss(lp.unsqueeze(1), yt.unsqueeze(0), torch.tensor([T]), torch.tensor([len(y)]),
                                           blank=0, reduction="sum", zero_infinity=True)
        assert abs(OS.ctc_nll_reference(lp, yt).item() - ref.item()) < 1e-9
