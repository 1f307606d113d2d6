"""`model.bert.type: x-transformers` (+ use_word_boundary: 513-wide rows stored at pitch 576) — the encoder the reference's
shipped LRW yamls select (LRW/video/config/bert-12l-512d_LRW_96_bf16_rrc_{WB,noWB}.yaml; lightning.py:93-105,145-158).

PARITY UNPINNED: x_transformers is a third-party package that is neither vendored in the reference tree nor importable offline, so
no golden vector exists for this branch.  These tests compare the HIP path with oracle/lrw_oracle.py::xt_encoder (the
restatement of the package's published algorithm, fp32 autograd) — self-consistency of forward AND hand-written backward — and
check the padded-storage invariants the contraction kernels rely on.  Tolerances: bf16 storage / fp32 accumulation as in
tests/test_gpu_model.py (losses 2e-3 relative at these tiny BatchNorm populations, encoder gradients cosine >= 0.99).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().float().cpu().flatten(), b.detach().float().cpu().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _cos(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))


# ------------------------------------------------------------------------------------------------------------------
# row passes against plain torch fp32 on the same bf16 inputs
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,ld,R", [(513, 576, 90), (512, 512, 960), (513, 576, 4100)])
def test_rmsnorm_forward_backward(dev, D, ld, R):
    from oracle import lrw_oracle as O
    from syncvsr_amd import ops

    g = torch.Generator().manual_seed(D + R)
    x = torch.zeros(R, ld)
    x[:, :D] = torch.randn(R, D, generator=g)
    gain = torch.zeros(ld)
    gain[:D] = 1 + 0.2 * torch.randn(D, generator=g)
    dy = torch.zeros(R, ld)
    dy[:, :D] = torch.randn(R, D, generator=g)
    add = torch.zeros(R, ld)
    add[:, :D] = torch.randn(R, D, generator=g)
    xb, dyb, addb = x.to(BF16), dy.to(BF16), add.to(BF16)
    xr = xb.float()[:, :D].clone().requires_grad_(True)
    gr = gain[:D].clone().requires_grad_(True)
    yr = O.rms_norm(xr, gr)
    yr.backward(dyb.float()[:, :D])
    y, inv = ops.rmsnorm_fwd(xb.to(dev), gain.to(dev), D)
    dg = torch.zeros(ld, device=dev)
    dx = ops.rmsnorm_bwd(dyb.to(dev), xb.to(dev), gain.to(dev), inv, dg, D, addend=addb.to(dev))
    dg2 = torch.zeros(ld, device=dev)
    dx2 = ops.rmsnorm_bwd(dyb.to(dev), xb.to(dev), gain.to(dev), inv, dg2, D, addend=addb.to(dev))
    torch.cuda.synchronize()
    assert _rel(y[:, :D], yr) <= 4e-3                      # one bf16 rounding of the output
    assert _rel(dx[:, :D], xr.grad + addb.float()[:, :D]) <= 4e-3
    assert _rel(dg[:D], gr.grad) <= 1e-4                   # fp32 partial rows, fixed order
    assert torch.equal(dx, dx2) and torch.equal(dg, dg2)   # reproducible
    if ld > D:                                             # pads stay exactly zero
        assert not y[:, D:].any() and not dx[:, D:].any() and not dg[D:].any()


@pytest.mark.parametrize("ntens", [3, 2])
def test_rotary_matches_restatement_and_its_transpose(dev, ntens):
    from oracle import lrw_oracle as O
    from syncvsr_amd import ops

    B, S, H = 3, 30, 8
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B * S, 3 * H * 64, generator=g).to(BF16)
    tab = ops.rotary_table(S, dev)
    out = qkv.to(dev).clone()
    ops.rotary_(out, tab, S, ntens * H, 1)
    back = out.clone()
    ops.rotary_(back, tab, S, ntens * H, -1)
    torch.cuda.synchronize()
    freqs = O.rotary_freqs(S)
    ref = qkv.float().view(B, S, 3, H, 64).clone()
    for t in range(ntens):
        ref[:, :, t] = O.apply_rotary(ref[:, :, t].transpose(1, 2), freqs).transpose(1, 2)     # [B,H,S,64] convention of the package
    assert _rel(out, ref.view(B * S, -1)) <= 4e-3
    assert _rel(back, qkv) <= 6e-3                         # rotation followed by its transpose = identity up to two bf16 roundings
    if ntens == 2:
        assert torch.equal(out[:, 2 * H * 64:].cpu(), qkv[:, 2 * H * 64:])        # v untouched


@pytest.mark.parametrize("I,ldu,ldy,p", [(2052, 4160, 2112, 0.3), (2048, 4096, 2048, 0.0)])
def test_geglu_forward_backward(dev, I, ldu, ldy, p):
    from oracle import lrw_oracle as O
    from syncvsr_amd import ops
    from syncvsr_amd.dropout import keep_mask

    R = 77
    g = torch.Generator().manual_seed(I)
    u = torch.zeros(R, ldu)
    u[:, : 2 * I] = torch.randn(R, 2 * I, generator=g)
    dy = torch.randn(R, ldy, generator=g)
    ub, dyb = u.to(BF16), dy.to(BF16)
    seed = torch.tensor([1234], dtype=torch.int32, device=dev)
    drop = (seed, 7, p) if p > 0 else None
    y = ops.geglu_fwd(ub.to(dev), I, ldy, drop=drop)
    du = ops.geglu_bwd(dyb.to(dev), ub.to(dev), I, drop=drop)
    torch.cuda.synchronize()
    ur = ub.float()[:, : 2 * I].clone().requires_grad_(True)
    val, gate = ur.chunk(2, dim=-1)
    yr = val * O.gelu_erf(gate)
    if p > 0:
        m = torch.from_numpy(keep_mask(1234, 7, p, R * ldy).reshape(R, ldy)[:, :I].copy()).float() / (1 - p)
        yr = yr * m
        assert abs(float((y[:, :I] == 0).float().mean()) - p) < 0.02
    yr.backward(dyb.float()[:, :I])
    assert _rel(y[:, :I], yr) <= 6e-3
    assert _rel(du[:, : 2 * I], ur.grad) <= 6e-3
    assert not y[:, I:].any() and not du[:, 2 * I:].any()


@pytest.mark.parametrize("wb", [True, False])
def test_embed_concat_forward_backward(dev, wb):
    from syncvsr_amd import ops

    B, S, F = 5, 30, 512
    D = F + 1 if wb else F
    ld = (D + 63) // 64 * 64
    g = torch.Generator().manual_seed(11)
    feats = torch.randn(B * (S - 1), F, generator=g).to(BF16)
    wm = (torch.rand(B, S - 1, generator=g) > 0.5).float()
    cls = torch.zeros(ld)
    cls[:D] = torch.randn(D, generator=g)
    x0 = ops.xt_embed_fwd(feats.to(dev), wm.to(dev) if wb else None, cls.to(dev), B, S, F, D, ld)
    ref = torch.zeros(B, S, ld)
    ref[:, 0] = cls.to(BF16).float()
    ref[:, 1:, :F] = feats.float().view(B, S - 1, F)
    if wb:
        ref[:, 1:, F] = wm
    assert torch.equal(x0.float().cpu(), ref.view(B * S, ld))
    dx0 = torch.randn(B * S, ld, generator=g).to(BF16)
    dcls = torch.zeros(ld, device=dev)
    dfe = ops.xt_embed_bwd(dx0.to(dev), dcls, B, S, F, D)
    torch.cuda.synchronize()
    assert torch.equal(dfe.cpu(), dx0.view(B, S, ld)[:, 1:, :F].reshape(B * (S - 1), F))
    assert _rel(dcls[:D], dx0.float().view(B, S, ld)[:, 0, :D].sum(0)) <= 1e-6 and not dcls[D:].any()


# ------------------------------------------------------------------------------------------------------------------
# whole model against the restatement (forward, every gradient), with layer skipping and the feed-forward dropout
# ------------------------------------------------------------------------------------------------------------------
def _pair(dev, wb, depth, B, frames, size, skip, ff_drop, seed=3):
    from oracle import lrw_oracle as O
    from syncvsr_amd.config import xtransformers_lrw_config
    from syncvsr_amd.init import init_state_dict, synthetic_batch
    from syncvsr_amd.model import Model

    cfg = xtransformers_lrw_config(wb, model__bert__depth=depth, model__bert__ff_dropout=ff_drop)
    sd = init_state_dict(cfg, seed=seed, perturb_norm=True)
    batch = synthetic_batch(cfg, B, frames=frames, size=size, seed=seed + 100)
    model = Model(cfg, seed=77)
    model.keep_audio_logits = True
    model.load_state_dict(sd, strict=True)
    model.to(dev).train()
    model.layer_skip_override = set(skip)
    out = model(*[t.to(dev) for t in batch])
    out["loss_total"].backward()
    torch.cuda.synchronize()
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    dp = None
    if ff_drop > 0:
        dp = O.DropPlan(int(model._drop_word.item()), ff_drop, 0.0, 0.0, model._sites)
    keep = {}
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    ref = O.forward(osd, cfg, *batch, training=True, keep=keep, dp=dp, layer_skip=set(skip))
    ref["loss_total"].backward()
    return cfg, model, out, osd, ref, keep


@pytest.mark.parametrize("wb,depth,B,frames,size,skip,ff_drop", [
    (True, 2, 2, 5, 32, (), 0.0),
    (True, 3, 3, 6, 32, (1, 2), 0.3),            # one feed-forward and one attention block skipped, dropout after the gate
    (False, 2, 2, 5, 32, (), 0.3),
    (True, 12, 4, 29, 88, (3, 8, 17), 0.3),      # the shipped yaml's encoder at the clip size of BASELINE.json configs[1]
])
def test_xt_model_matches_restatement(dev, wb, depth, B, frames, size, skip, ff_drop):
    cfg, model, out, osd, ref, keep = _pair(dev, wb, depth, B, frames, size, skip, ff_drop)
    D = model.dim
    for k in ("loss_total", "loss_category", "loss_audio"):
        assert abs(out[k].item() - ref[k].item()) <= 2e-3 * abs(ref[k].item()), (k, out[k].item(), ref[k].item())
    last = model._last
    hid = last["hidden"].view(B, frames + 1, -1)
    # tiny clips (32 x 32, <= 18 frames): BatchNorm statistics over as few as 96 values per channel amplify the bf16 noise of the
    # trunk (tests/test_gpu_model.py gives its tiny cases the same slack); the 88 x 88 case must meet the full-size bound
    assert _rel(hid[..., :D], keep["hidden"]) <= (3e-2 if size >= 88 else 6e-2)
    assert not hid[..., D:].any()
    tol = 1.0 if size >= 88 else 2.0
    assert _rel(last["logits_category"], keep["logits_category"]) <= 3e-2 * tol
    assert _rel(last["logits_audio"], keep["logits_audio"]) <= 5e-2 * tol
    worst = []
    for n, p in model.named_parameters():
        r = osd[n].grad if osd[n].grad is not None else torch.zeros_like(osd[n])          # autograd leaves skipped blocks at None
        assert tuple(p.shape) == tuple(r.shape) == tuple(p.grad.shape), n
        if r.norm().item() <= 1e-7:
            assert p.grad.float().norm().item() <= 1e-6, ("skipped block must have zero gradient", n)
            continue
        c = _cos(p.grad, r)
        worst.append((c, n, float(p.grad.float().norm().cpu() / r.norm())))
        if n.startswith("encoder.") or n in ("cls_token", "audio_projection.weight", "category_classifier.weight"):
            assert c >= 0.99, (n, c)
    worst.sort()
    print("worst gradient cosines:", worst[:4])
    # trunk tensors: the bf16 ReLU-mask floor of tests/test_gpu_model.py (tiny clips measured median 0.94, the 88 x 88 case 0.99+)
    assert worst[0][0] >= 0.80 and worst[len(worst) // 2][0] >= (0.99 if size >= 88 else 0.92), worst[:3]
    for s in skip:           # skipped blocks: all their parameters untouched
        for n, p in model.named_parameters():
            if n.startswith(f"encoder.layers.{s}."):
                assert not p.grad.any(), n


def test_padded_storage_stays_zero_and_state_dict_is_logical(dev):
    """513-wide tensors are the [:513] corners of 576-wide storage; after optimiser steps the pads are still exactly zero and the
    state dict carries the logical shapes of the reference's modules (nn.Linear(513, ...), RMSNorm(513), cls_token [1,1,513])."""
    from syncvsr_amd.config import xtransformers_lrw_config
    from syncvsr_amd.engine import TrainStep
    from syncvsr_amd.init import synthetic_batch
    from syncvsr_amd.model import Model

    cfg = xtransformers_lrw_config(True, model__bert__depth=2, optim__optimizer__lr=1e-3, optim__scheduler__num_warmup_steps=1)
    model = Model(cfg, seed=1).to(dev).train()
    ts = TrainStep(model, cfg)
    batch = [t.to(dev) for t in synthetic_batch(cfg, 2, frames=5, size=32)]
    losses = [float(ts.step(*batch)["loss_total"]) for _ in range(4)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    sd = model.state_dict()
    assert sd["cls_token"].shape == (1, 1, 513)
    assert sd["encoder.layers.0.1.to_q.weight"].shape == (512, 513)
    assert sd["encoder.layers.0.1.to_out.weight"].shape == (513, 512)
    assert sd["encoder.layers.1.1.ff.0.proj.weight"].shape == (4104, 513)
    assert sd["encoder.layers.1.1.ff.3.weight"].shape == (513, 2052)
    assert sd["encoder.layers.1.0.0.g"].shape == (513,)
    assert sd["audio_projection.weight"].shape == (2560, 513) and sd["category_classifier.weight"].shape == (500, 513)
    st = model.store()
    for n, (o, numel, shape) in st.offsets.items():
        ph = st.phys[n]
        if ph == shape or len(shape) == 4:
            continue
        full = st.flat[o:o + numel].view(ph).clone()
        full[tuple(slice(0, d) for d in shape)] = 0
        assert not full.any(), f"pad region of {n} is no longer zero"
        gfull = st.grad[o:o + numel].view(ph).clone()
        gfull[tuple(slice(0, d) for d in shape)] = 0
        assert not gfull.any(), f"pad region of the gradient of {n} is not zero"
    # reload into a fresh model: same loss
    model2 = Model(cfg, seed=9)
    model2.load_state_dict({k: v.cpu() for k, v in sd.items()}, strict=True)
    model2.to(dev).eval()
    model.eval()
    with torch.no_grad():
        a, b = model(*batch)["loss_total"].item(), model2(*batch)["loss_total"].item()
    assert a == b


def test_layer_dropout_draws_like_the_package_and_is_reproducible(dev):
    """layer_dropout: python's random() once per layer in order (AttentionLayers.forward); with the same seed two models skip the
    same blocks and produce bit-identical losses; the skip rate matches the configured probability."""
    from syncvsr_amd.config import xtransformers_lrw_config
    from syncvsr_amd.init import synthetic_batch
    from syncvsr_amd.model import Model, _xt_skips

    cfg = xtransformers_lrw_config(True, model__bert__depth=12)
    m = Model(cfg, seed=5).train()
    n = sum(len(_xt_skips(m)) for _ in range(400))
    assert abs(n / (400 * 24) - 0.2) < 0.02
    m.eval()
    assert _xt_skips(m) == set()
    cfg2 = xtransformers_lrw_config(True, model__bert__depth=3)
    batch = [t.to(dev) for t in synthetic_batch(cfg2, 2, frames=5, size=32)]
    runs = []
    for _ in range(2):
        mm = Model(cfg2, seed=21).to(dev).train()
        runs.append([mm(*batch)["loss_total"].item() for _ in range(3)])
    assert runs[0] == runs[1]
    assert len(set(runs[0])) > 1          # different blocks / masks from step to step
