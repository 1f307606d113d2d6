"""Block-by-block backward parity of the visual front-end at the benchmark batch (-m gpu).

The whole-model comparisons (test_gpu_model.py) cross 17 ReLU stages: one bf16 rounding that falls the other way flips a mask bit or
moves a value by one ulp, the difference is amplified from block to block, and the trunk's gradient cosines against ANY CPU
restatement settle around 0.90-0.97 — a bound on the chaos, not on the kernels.  Here every residual block (and the stem) is checked on
its own: the oracle's bf16-storage emulation (oracle.lrw_oracle, emu=True) is run on the block's INPUT exactly as the HIP forward stored
it, and back-propagated from the gradient exactly as the HIP backward received it (model._frontend_backward records both when asked).
One block deep the two must agree tightly: parameter gradients and the gradient handed to the block below to cosine >= 0.9999 and norm
ratio within 0.2 % (measured: >= 0.99999, within 0.02 %) — a 10 % error in the fused BatchNorm-backward epilogues, the 64-channel kernel, the stem's gather-form backward or
any weight-gradient kernel cannot hide behind the ReLU-flip argument here.  Shapes: B = 32 clips (928 frames), what bench.py times.
Reference: LRW/video/src/lightning.py:49-55,112-119 (stem3d, resnet18 trunk), tcn/models/resnet.py:28-72 (BasicBlock)."""
import os

import pytest
import torch

from golden_cases import build_case

pytestmark = pytest.mark.gpu


def _nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def _cmp(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300)), float(a.norm() / (b.norm() + 1e-300))


def _swish(t):
    return t * torch.sigmoid(t)


def _blockwise(model, sd, videos, relu: bool, cos_floor: float, ratio_band: float, bn_floor=None):
    """videos [B,1,T,H,W] fp32 (CPU).  relu: ReLU trunk + GELU stem (word-level) or Swish everywhere (sentence-level)."""
    from oracle import lrw_oracle as O
    from syncvsr_amd import model as M

    dev = torch.device("cuda:0")
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    st = model.store()
    st.refresh_shadows()
    B, _, T = videos.shape[:3]
    tape = {"_record_grads": {}}
    feats = M._frontend_forward(model, st, tape, videos.to(dev).float().contiguous(), True)
    g = torch.Generator().manual_seed(3)
    dfeats = (torch.randn(feats.shape, generator=g) * 1e-2).to(torch.bfloat16).to(dev)
    st.zero_grad()
    M._frontend_backward(model, st, tape, dfeats)
    torch.cuda.synchronize()
    rec = tape["_record_grads"]
    stem, trunk = model.stem_name, model.trunk_name
    hip_grad = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if n.startswith((trunk + ".", stem + "."))}
    blocks = list(M._trunk_blocks(model))
    act = torch.relu if relu else _swish
    worst = []

    def check(what, a, b, cos_min=cos_floor, ratio_tol=ratio_band):
        cos, ratio = _cmp(a, b)
        worst.append((cos, ratio, what))
        if os.environ.get("SVSR_BLOCKWISE_REPORT") == "1":
            print(f"{what:60s} cos {cos:.6f} ratio {ratio:.5f}")
            return
        assert cos >= cos_min and abs(ratio - 1.0) <= ratio_tol, (what, cos, ratio)

    def act_grad_below(bi, x):
        """act'(z) of the block below, whose output is this block's input x: what the recorded gradient was multiplied by."""
        if relu:
            return (x.detach() > 0).float()
        tp = tape[f"{blocks[bi - 1][0]}.conv2"]            # z = bn2(c) + residual, rebuilt from what the forward kept
        gam, bet = (sd[f"{tp['bn']}.{k}"].float().to(dev) for k in ("weight", "bias"))
        z = (tp["c"].float() - tp["mean"]) * tp["rstd"] * gam + bet + tp["res"].float()
        sg = torch.sigmoid(z)
        return _nchw(sg * (1.0 + z * (1.0 - sg)))

    for bi, (prefix, inp, planes, stride, down) in enumerate(blocks):
        x_hip = tape[f"{prefix}.conv1"]["x"]                       # the block's input as the HIP forward stored it (bf16, NHWC)
        x = _nchw(x_hip).requires_grad_(True)
        names = [n for n in hip_grad if n.startswith(prefix + ".")]
        osd = {k: v for k, v in sd.items() if k.startswith(prefix + ".")}
        for n in names:
            osd[n] = sd[n].clone().requires_grad_(True)
        keep = {}
        y = O.basic_block(x, osd, prefix, stride, True, None, emu=True, keep=keep, act=act)
        up, masked = rec[prefix]
        (keep[f"{prefix}.z"] if masked else y).backward(_nchw(up))      # masked: the recorded gradient already carries act'(z)
        for n in names:
            if bn_floor is not None and ".bn1." in n:
                check(n, hip_grad[n], osd[n].grad, *bn_floor)
            else:
                check(n, hip_grad[n], osd[n].grad)
        # what the block hands down: the gradient of its input, as recorded at the block below (or at the stem)
        below = rec[blocks[bi - 1][0]] if bi > 0 else rec["stem"]
        gx = x.grad * act_grad_below(bi, x) if below[1] else x.grad
        check(f"{prefix}: gradient of the block input", _nchw(below[0]), gx)
    # ---- stem: Conv3d -> BatchNorm3d -> GELU / Swish -> MaxPool3d, from the gradient of the pooled output ---------------------------
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items() if k.startswith(stem + ".")}
    pooled = O.stem3d(videos.float(), osd, True, None, None, emu=True, prefix=stem, act=None if relu else _swish)          # [B, 64, T, 22, 22]
    up = _nchw(rec["stem"][0])                                                        # [B*T, 64, 22, 22]
    pooled.backward(up.unflatten(0, (B, T)).transpose(1, 2).contiguous())
    for n in (f"{stem}.0.weight", f"{stem}.1.weight", f"{stem}.1.bias"):
        # BatchNorm parameters of the stem: sums of g = dpool * act'(z) over 450 k positions per channel that cancel to a few per cent
        # of their terms; g is a bf16 tensor between the backward's two passes (as under the reference's bf16 autocast), the HIP
        # path rounds it per pooled output and the oracle per convolution element — two realisations of the same rounding noise,
        # 0.2 % of the sums' norm at B = 32
        check(n, hip_grad[n], osd[n].grad, ratio_tol=ratio_band if n.endswith(".0.weight") else max(ratio_band, 0.005))
    worst.sort()
    print("lowest cosines:", [(round(c, 5), round(r, 4), n) for c, r, n in worst[:6]])
    print("largest norm deviations:", [(round(c, 5), round(r, 4), n) for c, r, n in sorted(worst, key=lambda w: -abs(w[1] - 1.0))[:6]])


@pytest.mark.parametrize("case", ["lrw_full_b32", "lrw_full_b2"])
def test_front_end_backward_block_by_block(case):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from syncvsr_amd import model as M

    cfg, sd, batch, training, gold = build_case(case)
    model = M.Model(cfg)
    model.load_state_dict(sd, strict=True)
    model.to(torch.device("cuda:0")).train(True)
    # (B = 2: 58 frames, sixteen times fewer terms per sum than the benchmark batch — the rounding noise is ~4x larger: measured 0.99994 / 0.27 %)
    cos_floor, ratio_band = (0.9999, 0.002) if case == "lrw_full_b32" else (0.9995, 0.01)
    _blockwise(model, sd, batch[0], True, cos_floor, ratio_band)


def test_lrs_swish_front_end_backward_block_by_block():
    """The sentence-level model's front-end (Swish stem and trunk: backbones/conv3d_extractor.py:40-48, backbones/modules/resnet.py:90-107)
    at the shape bench.py's LRS leg times: 16 clips x 160 frames = 2560 frames of the shipped config (LRS/video/config/lrs3.yaml).
    Swish has no mask to flip, but the fused epilogues multiply by swish'(z) of a RECOMPUTED pre-activation; block by block they must
    match the oracle's bf16-storage emulation as tightly as the ReLU trunk does."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from syncvsr_amd.lrs_init import LRS_ODIM, default_lrs_args, lrs_init_state_dict, lrs_synthetic_batch
    from syncvsr_amd.lrs_model import E2E

    args = default_lrs_args(dropout_rate=0.0, transformer_attn_dropout_rate=0.0)
    x, lengths, _, _ = lrs_synthetic_batch(args, 16, 160, seed=1234)
    sd = lrs_init_state_dict(args, LRS_ODIM, seed=0)
    model = E2E(LRS_ODIM, args)
    model.load_state_dict(sd, strict=True)
    model.to(torch.device("cuda:0")).train()
    # bn1 parameters: sums of g = dout * swish'(z) over 1.2 M positions per channel that cancel to a fraction of a per cent of their terms;
    # the HIP epilogue sums the fp32 products and stores their bf16 rounding, the oracle sums the rounded ones
    # (measured: layer1.0.bn1.bias cosine 0.99971, norm ratio 1.0021; everything else >= 0.99994 / within 0.1 %)
    _blockwise(model, sd, x.view(16, 1, 160, 88, 88), False, 0.9999, 0.002, bn_floor=(0.9995, 0.004))


def test_lrs_conformer_layer_backward_alone():
    """One Conformer layer of the shipped sentence-level config on its own, at the LRS benchmark shape (16 clips x 160 frames, adim 768,
    12 heads, 3072 units, kernel 31; LRS/video/config/lrs3.yaml, transformer/encoder_layer.py:90-165): forward from a given bf16
    input, backward from a given bf16 gradient, against the fp32 oracle layer on the same numbers (ragged lengths: the attention
    masks padded keys, everything else — the convolution module's BatchNorm included — sees every frame, as in the reference).
    One layer deep the bf16 storage of ~25 intermediates is all that separates the two: output and input gradient to relative L2
    <= 1 %, every parameter gradient to cosine >= 0.999 (0.998 behind a ReLU mask, see below) and norm within 1 %."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from oracle import lrs_oracle as O
    from syncvsr_amd import lrs_model as L
    from syncvsr_amd import model as M
    from syncvsr_amd.lrs_init import LRS_ODIM, default_lrs_args, lrs_init_state_dict

    dev = torch.device("cuda:0")
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    args = default_lrs_args(dropout_rate=0.0, transformer_attn_dropout_rate=0.0)
    sd = lrs_init_state_dict(args, LRS_ODIM, seed=0)
    model = L.E2E(LRS_ODIM, args)
    model.load_state_dict(sd, strict=True)
    model.to(dev).train()
    st = model.store()
    st.refresh_shadows()
    B, T, D = 16, 160, model.adim
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B * T, D, generator=g).to(torch.bfloat16)
    dy = (torch.randn(B * T, D, generator=g) * 1e-2).to(torch.bfloat16)
    lengths = torch.randint(T // 2, T + 1, (B,), generator=g)
    lengths[0] = T
    ilen = lengths.to(device=dev, dtype=torch.int32)
    pos16 = model._pos_table("rel", T, dev)
    tape = {}
    p = "encoder.encoders.0"
    y = L._encoder_layer_fwd(model, st, tape, 0, x.to(dev), pos16, ilen, B, T, True)
    st.zero_grad()
    st.rebind_grads()
    dx = L._encoder_layer_bwd(model, st, tape, 0, dy.to(dev), pos16, B, T)
    M._flush_deferred(model)
    model._side.join()
    torch.cuda.synchronize()
    hip_grad = {n: q.grad.detach().float().cpu() for n, q in model.named_parameters() if n.startswith(p + ".")}
    assert len(hip_grad) >= 30
    osd = {k: v for k, v in sd.items() if k.startswith(p + ".")}
    for n in hip_grad:
        osd[n] = sd[n].clone().requires_grad_(True)
    xo = x.float().view(B, T, D).requires_grad_(True)
    mask = (torch.arange(T).unsqueeze(0) < lengths.view(-1, 1)).unsqueeze(-2)
    pos = pos16.float().cpu().view(1, 2 * T - 1, D)
    yo = O.encoder_layer(xo, pos, mask, osd, p, int(args.aheads), True, None, None, "enc.0")
    yo.backward(dy.float().view(B, T, D))

    def rel(a, b):
        a, b = a.float().cpu().flatten().double(), b.flatten().double()
        return float((a - b).norm() / b.norm())

    ry, rdx = rel(y, yo.detach()), rel(dx, xo.grad)
    worst = []
    kb, cb = f"{p}.self_attn.linear_k.bias", f"{p}.conv_module.depthwise_conv.bias"
    for n, gh in hip_grad.items():
        assert osd[n].grad is not None, n
        if n not in (kb, cb):
            worst.append((*_cmp(gh, osd[n].grad), n))
    # two gradients that are zero in exact arithmetic, rounding noise on both sides: the key bias shifts every score of a query by
    # the same amount, which the softmax removes; the depthwise convolution's bias is a per-channel constant in front of a
    # training-mode BatchNorm.  Bounded against a neighbouring gradient instead of compared
    kq = float(hip_grad[kb].norm() / hip_grad[f"{p}.self_attn.linear_q.bias"].norm())
    cq = float(hip_grad[cb].norm() / hip_grad[f"{p}.conv_module.norm.bias"].norm())
    print("key-bias / query-bias gradient norm", kq, "depthwise bias / BatchNorm bias gradient norm", cq)
    assert kq <= 0.02 and cq <= 0.02, (kq, cq)
    worst.sort()
    print("output rel L2", ry, "input gradient rel L2", rdx)
    print("lowest cosines:", [(round(c, 5), round(r, 4), n) for c, r, n in worst[:6]])
    print("largest norm deviations:", [(round(c, 5), round(r, 4), n) for c, r, n in sorted(worst, key=lambda w: -abs(w[1] - 1.0))[:6]])
    assert ry <= 0.01 and rdx <= 0.01, (ry, rdx)
    # the two feed-forward modules are ReLU networks (positionwise_feed_forward.py:30): ~0.3 % of the 2560 x 3072 hidden units sit
    # close enough to zero for the bf16 rounding of their inputs to flip the mask against an fp32 evaluation, and a flipped fraction f
    # moves what is summed THROUGH the mask (w_1, its bias, the LayerNorm in front) by ~sqrt(f) — measured cosine 0.9986-0.9992 there,
    # >= 0.9999 on everything the mask does not gate
    def floor(n):
        return 0.998 if (".w_1." in n or ".norm_ff" in n) else 0.999

    bad = [(n, cos, ratio) for cos, ratio, n in worst if not (cos >= floor(n) and abs(ratio - 1.0) <= 0.01)]
    assert not bad, bad
