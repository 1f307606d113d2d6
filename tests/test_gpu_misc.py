"""-m gpu: shadow cast / table-driven transpose-cast, bias-gradient tiling at encoder sizes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def test_transpose_cast_multi(dev):
    from syncvsr_amd import ops

    g = torch.Generator().manual_seed(3)
    shapes = [(64, 9, 64), (128, 9, 64), (500, 1, 512), (1536, 1, 512), (40, 3, 72)]     # (A, T, Bd)
    src_parts, entries, soff, doff = [], [], 0, 0
    for A, T, Bd in shapes:
        src_parts.append(torch.randn(A * T * Bd, generator=g))
        Apad = (A + 63) // 64 * 64
        entries.append((soff, doff, A, T, Bd, Apad))
        soff += A * T * Bd
        doff += Bd * T * Apad
    src = torch.cat(src_parts).to(dev)
    dst = torch.zeros(doff, dtype=BF, device=dev)
    tab = np.zeros(len(entries), dtype=np.dtype([("src", "<i8"), ("dst", "<i8"), ("A", "<i4"), ("T", "<i4"), ("Bd", "<i4"), ("Apad", "<i4")]))
    for i, e in enumerate(entries):
        tab[i] = e
    table = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)
    ops.transpose_cast_multi(src, dst, table, len(entries))
    shadow = torch.empty(soff, dtype=BF, device=dev)
    ops.cast_bf16(src, shadow)
    assert torch.equal(shadow.cpu(), src.cpu().to(BF))
    for (so, do, A, T, Bd, Apad), part in zip(entries, src_parts):
        ref = torch.zeros(Bd, T, Apad, dtype=BF)
        ref[:, :, :A] = part.view(A, T, Bd).permute(2, 1, 0).to(BF)
        got = dst[do : do + Bd * T * Apad].view(Bd, T, Apad).cpu()
        assert torch.equal(got, ref), (A, T, Bd)


@pytest.mark.parametrize("R,N", [(960, 512), (960, 2048), (928, 2560), (32, 512), (60, 1536)])
def test_bias_grad_sizes(dev, R, N):
    from syncvsr_amd import ops

    g = torch.Generator().manual_seed(5)
    dy = torch.randn(R, N, generator=g).to(BF)
    db = torch.zeros(N, device=dev)
    ops.bias_act_bwd(dy.to(dev), None, db, R=R, N=N, n_valid=N, ld=N)
    ref = dy.float().sum(0)
    err = (db.cpu() - ref).abs().max().item()
    assert err <= 1e-3 * ref.abs().max().item() + 1e-3, err


def test_training_step_with_cutmix(dev):
    """training_step with `use_cutmix: true` (the shipped LRW recipe): device CutMix -> soft-label CE -> loss vs the oracle on the
    same mixed batch."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from golden_cases import build_case
    from oracle import lrw_oracle as O
    from syncvsr_amd.model import Model

    cfg, sd, batch, training, gold = build_case("lrw_tiny")
    cfg.train.use_cutmix = True
    hard_batch = [batch[0], batch[1], batch[2], batch[3]]
    model = Model(cfg)
    model.load_state_dict(sd)
    model.to(dev).train()
    torch.manual_seed(123)
    mixed = model.cutmix(*[t.to(dev) for t in hard_batch])
    assert mixed[2].shape == (batch[0].shape[0], 500) and mixed[2].dtype == torch.float32
    out = model(*mixed)
    out["loss_total"].backward()
    ref = O.forward({k: v.clone() for k, v in sd.items()}, cfg, *[t.cpu() for t in mixed], training=True, use_cutmix_metric=True)
    for k in ("loss_total", "loss_category", "loss_audio"):
        assert abs(out[k].item() - ref[k].item()) <= 2e-2 * abs(ref[k].item()), (k, out[k].item(), ref[k].item())
    torch.manual_seed(123)
    loss = model.training_step([t.to(dev) for t in hard_batch])
    assert torch.isfinite(loss) and abs(loss.item() - out["loss_total"].item()) <= 1e-3 * abs(loss.item())


@pytest.mark.parametrize("case", [(2, 7, 9), (1, 22, 22), (300, 22, 22), (5, 3, 3), (3, 29, 13)])
def test_conv3x3_c64_persistent(dev, case):
    """Weights-in-LDS persistent conv3x3(64,64): forward (+BN partial sums) and data-gradient (+addend) vs torch."""
    import math

    import torch.nn.functional as F

    from syncvsr_amd import ops

    N, H, W = case
    g = torch.Generator().manual_seed(21)
    x = torch.randn(N, H, W, 64, generator=g).to(BF)
    w = (torch.randn(64, 3, 3, 64, generator=g) / math.sqrt(576)).to(BF)
    assert ops.C64_CONV
    out, stats = ops.conv2d_fwd(x.to(dev), w.to(dev), 3, 1, 1, want_stats=True)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=1)
    refn = ref.permute(0, 2, 3, 1)
    err = (out.float().cpu() - refn).abs().max().item()
    assert err <= 1.5e-2 * refn.abs().max().item(), err
    st = stats[0][: stats[1] * 128].view(stats[1], 2, 64).double().sum(0).float().cpu()
    q_ref = (ref * ref).sum((0, 2, 3))
    assert ((st[1] - q_ref).abs() <= 3e-3 * q_ref).all()
    assert ((st[0] - ref.sum((0, 2, 3))).abs() <= 2e-3 * torch.sqrt(ref[0].numel() / 64 * N * q_ref)).all()
    # data gradient with residual-gradient addend (in place)
    dy = torch.randn(N, H, W, 64, generator=g).to(BF)
    add = torch.randn(N, H, W, 64, generator=g).to(BF)
    xs = torch.zeros(N, 64, H, W, requires_grad=True)
    F.conv2d(xs, w.float().permute(0, 3, 1, 2), padding=1).backward(dy.float().permute(0, 3, 1, 2))
    wt = w.permute(3, 1, 2, 0).contiguous()
    dx = ops.conv2d_dgrad(dy.to(dev), wt.to(dev), 3, 1, 1, (H, W), addend=add.to(dev).clone())
    ref_dx = xs.grad.permute(0, 2, 3, 1) + add.float()
    err = (dx.float().cpu() - ref_dx).abs().max().item()
    assert err <= 1.5e-2 * ref_dx.abs().max().item(), err
    # and the generic implicit-GEMM kernel agrees
    ops.C64_CONV = False
    try:
        out2, _ = ops.conv2d_fwd(x.to(dev), w.to(dev), 3, 1, 1)
    finally:
        ops.C64_CONV = True
    assert (out2.float() - out.float()).abs().max().item() <= 1e-2 * refn.abs().max().item()


@pytest.mark.parametrize("case", [(2, 6, 6, 128, 256), (3, 11, 11, 128, 64), (1, 22, 22, 64, 64), (2, 7, 9, 64, 128), (33, 6, 6, 64, 64)])
def test_wgrad3x3_halo(dev, case):
    """Nine-taps-per-pass weight gradient (zero-padded coordinates) against torch's conv2d weight gradient."""
    import torch.nn.functional as F

    from syncvsr_amd import ops

    N, H, W, Ci, Co = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, H, W, Ci, generator=g).to(BF)
    dy = torch.randn(N, H, W, Co, generator=g).to(BF)
    ws = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(x.float().permute(0, 3, 1, 2), ws, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    ref = ws.grad.permute(0, 2, 3, 1)
    assert ops.HALO_WGRAD
    dw = torch.zeros(Co, 3, 3, Ci, device=dev)
    ops.conv2d_wgrad(x.to(dev), dy.to(dev), dw, 3, 1, 1, use_tr=True)
    l2 = ((dw.cpu() - ref).norm() / ref.norm()).item()
    assert l2 <= 2e-3, l2
    # accumulation semantics: a second call adds
    ops.conv2d_wgrad(x.to(dev), dy.to(dev), dw, 3, 1, 1, use_tr=True)
    l2 = ((dw.cpu() - 2 * ref).norm() / (2 * ref).norm()).item()
    assert l2 <= 2e-3, l2


def test_device_clip_pipeline(dev):
    """svsr_clip_prep: uint8 stored frames -> x/255 -> flip -> crop window resized (bilinear, align_corners=False) -> normalise,
    against torch's F.interpolate on the CPU (reference chain: LRW/video/src/data.py:150,157-171)."""
    import torch.nn.functional as F

    from syncvsr_amd import ops
    from syncvsr_amd.augment import DeviceClipPipeline

    g = torch.Generator().manual_seed(5)
    B, T, Hs, Ws, S = 5, 7, 96, 112, 88
    frames = torch.randint(0, 256, (B, T, Hs, Ws), dtype=torch.uint8, generator=g)
    pipe = DeviceClipPipeline(S, train=True, seed=11)
    params = pipe.draw(B, Hs, Ws)
    params[0] = torch.tensor([4, 12, 88, 88, 0])            # a plain centre crop
    params[1] = torch.tensor([0, 0, 96, 112, 1])            # whole frame, flipped
    out = pipe(frames.to(dev), params).cpu()
    assert out.shape == (B, 1, T, S, S)
    for b in range(B):
        top, left, h, w, flip = (int(v) for v in params[b])
        x = frames[b].float() / 255.0
        # windows are in stored-frame coordinates and the flip mirrors the output: the same distribution as the reference's
        # flip-then-crop, whose window position is uniform
        crop = x[:, top:top + h, left:left + w]
        ref = F.interpolate(crop.unsqueeze(1), size=(S, S), mode="bilinear", align_corners=False, antialias=False).squeeze(1)
        if flip:
            ref = ref.flip(-1)
        ref = (ref - 0.421) / 0.165
        err = (out[b, 0] - ref).abs().max().item()
        assert err <= 2e-5 * max(1.0, ref.abs().max().item()), (b, err)
    assert torch.equal(out[0, 0], ((frames[0, :, 4:92, 12:100].float() / 255.0) - 0.421) / 0.165) or (out[0, 0] - ((frames[0, :, 4:92, 12:100].float() / 255.0) - 0.421) / 0.165).abs().max() < 1e-6
    ev = DeviceClipPipeline(S, train=False)(frames.to(dev)).cpu()
    assert (ev[:, 0] - ((frames[:, :, 4:92, 12:100].float() / 255.0) - 0.421) / 0.165).abs().max() < 1e-6
