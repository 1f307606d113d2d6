"""Kernel parity AT THE BENCHMARK SHAPES (-m gpu): every contraction launch of the B = 32 LRW step (928 frames; BASELINE.json
configs[1]) and the large LRS linears (2,400 rows), each against torch fp32 on the same bf16-rounded operands.

bench.py's kernel instantiations are chosen from the problem shape (svsr_conv_plan / svsr_rows_plan / svsr_wgrad_plan), so running
the benchmark's own shapes here runs the benchmark's own instantiations; each test also asks the library which instantiation it
launched, asserts the ones the step is known to depend on and records all of them in gpurun_out/bench_shape_instantiations.json
so that coverage cannot silently move when the launch heuristics change.  Tolerances are those of tests/test_gpu_kernels.py.
"""
import json
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16
N_FRAMES = 928          # 32 clips x 29 frames
_SEEN: dict = {}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from syncvsr_amd import _lib

    _lib.load()
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    yield torch.device("cuda:0")
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "bench_shape_instantiations.json"), "w") as f:
        json.dump(_SEEN, f, indent=1, sort_keys=True)


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(BF)


def check(got, ref, name, max_tol=1.5e-2, l2_tol=6e-3):
    got = got.detach().float().cpu()
    ref = ref.detach().float()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{name}: non-finite values"
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    l2 = ((got - ref).norm() / (ref.norm() + 1e-30)).item()
    assert err <= max_tol * scale and l2 <= l2_tol, f"{name}: max err {err:.3e} (scale {scale:.3e}), rel L2 {l2:.3e}"


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


# every convolution of the ResNet18 trunk at 928 frames of 22x22 after the stem (88x88 clips):
# name: (H, W, Ci, Co, k, stride, pad)
TRUNK = {
    "layer1.conv": (22, 22, 64, 64, 3, 1, 1),
    "layer2.0.conv1": (22, 22, 64, 128, 3, 2, 1),
    "layer2.0.downsample": (22, 22, 64, 128, 1, 2, 0),
    "layer2.conv": (11, 11, 128, 128, 3, 1, 1),
    "layer3.0.conv1": (11, 11, 128, 256, 3, 2, 1),
    "layer3.0.downsample": (11, 11, 128, 256, 1, 2, 0),
    "layer3.conv": (6, 6, 256, 256, 3, 1, 1),
    "layer4.0.conv1": (6, 6, 256, 512, 3, 2, 1),
    "layer4.0.downsample": (6, 6, 256, 512, 1, 2, 0),
    "layer4.conv": (3, 3, 512, 512, 3, 1, 1),
}
# the instantiations the B = 32 step is known to run today; a change of the launch heuristics must update this table
# deliberately (and keeps the parity coverage, because the shapes stay the benchmark's)
EXPECT_FWD = {
    "layer2.conv": "k_igemm_p8<256,128,3>",          # the persistent 8-wave kernel (csrc/igemm_p8.hip)
    "layer3.conv": "k_igemm_p8<256,128,3>",
    "layer4.conv": "k_igemm_p8<256,64,3>",            # 132 items of 256 x 128 would leave half the CUs idle: 264 of 256 x 64
}




@pytest.mark.parametrize("name", list(TRUNK))
def test_trunk_conv_fwd_dgrad_wgrad_at_928_frames(dev, name):
    from syncvsr_amd import ops

    H, W, Ci, Co, k, s, p = TRUNK[name]
    N = N_FRAMES
    Ho, Wo = ops.conv_out_size(H, k, s, p), ops.conv_out_size(W, k, s, p)
    x = rnd((N, H, W, Ci), 1)
    w = rnd((Co, k, k, Ci), 2, 1.0 / math.sqrt(k * k * Ci))
    wf = w.float().permute(0, 3, 1, 2)
    xd, wd = x.to(dev), w.to(dev)
    # ---- forward + BatchNorm partial sums --------------------------------------------------------------------------
    out, stats = ops.conv2d_fwd(xd, wd, k, s, p, want_stats=True)
    c64 = ops._c64_ok(Ci, Co, k, s, p, W)
    _SEEN[f"{name}.fwd"] = "k_conv3x3_c64" if c64 else ops.conv_plan(0, N, H, W, Co, k, s, p).label
    if name in EXPECT_FWD:
        assert _SEEN[f"{name}.fwd"] == EXPECT_FWD[name], _SEEN[f"{name}.fwd"]
    ref = F.conv2d(nchw(x.float()), wf, stride=s, padding=p)
    check(out, nhwc(ref), f"{name}.fwd")
    buf, rows = stats
    st = buf[: rows * 2 * Co].view(rows, 2, Co).double().sum(0).float().cpu()
    count = ref.numel() // Co
    q_ref = (ref * ref).sum((0, 2, 3))
    assert bool(((st[0] - ref.sum((0, 2, 3))).abs() <= 2e-3 * torch.sqrt(count * q_ref)).all()), f"{name}.stats.sum"
    check(st[1], q_ref, f"{name}.stats.sumsq", 3e-3, 2e-3)
    # ---- data gradient (stride 2: one launch per input-parity class) ----------------------------------------------
    dy = rnd((N, Ho, Wo, Co), 3)
    w2 = rnd((Co, k, k, Ci), 4, 1.0 / math.sqrt(k * k * Co))
    wt = w2.permute(3, 1, 2, 0).contiguous()
    xs = torch.zeros(N, Ci, H, W, requires_grad=True)
    F.conv2d(xs, w2.float().permute(0, 3, 1, 2), stride=s, padding=p).backward(nchw(dy.float()))
    add = rnd((N, H, W, Ci), 5)
    dx = ops.conv2d_dgrad(dy.to(dev), wt.to(dev), k, s, p, (H, W), addend=add.to(dev).clone())
    _SEEN[f"{name}.dgrad"] = "k_conv3x3_c64" if c64 else ops.conv_plan(2, N, H, W, Ci, k, s, p).label
    check(dx, nhwc(xs.grad) + add.float(), f"{name}.dgrad+addend")
    dx0 = ops.conv2d_dgrad(dy.to(dev), wt.to(dev), k, s, p, (H, W))          # without addend: untouched pixels (k < stride) are zeros
    check(dx0, nhwc(xs.grad), f"{name}.dgrad")
    # ---- weight gradient ------------------------------------------------------------------------------------------
    ws = torch.zeros(Co, Ci, k, k, requires_grad=True)
    F.conv2d(nchw(x.float()), ws, stride=s, padding=p).backward(nchw(dy.float()))
    dw = torch.zeros(Co, k, k, Ci, device=dev)
    ops.conv2d_wgrad(xd, dy.to(dev), dw, k, s, p, use_tr=True)
    halo = ops.HALO_WGRAD and k == 3 and s == 1 and p == 1 and W <= 29 and H * W >= 100
    if halo:
        _SEEN[f"{name}.wgrad"] = "k_wgrad3x3_halo"
    else:
        wp = ops.wgrad_conv_plan(N, H, W, Ci, Co, k, s, p)
        _SEEN[f"{name}.wgrad"] = f"{wp.label} x{wp.splits} splits"
    check(dw, ws.grad.permute(0, 2, 3, 1), f"{name}.wgrad", 3e-3, 2e-3)
    # a second identical launch gives the bit-identical result (fixed-order split-K reduction; accumulate semantics -> 2x)
    dw2 = torch.zeros_like(dw)
    ops.conv2d_wgrad(xd, dy.to(dev), dw2, k, s, p, use_tr=True)
    assert torch.equal(dw, dw2), f"{name}.wgrad is not reproducible"


def test_benchmark_instantiations_cover_the_big_tiles(dev):
    """The B = 32 shapes must select the tiles bench.py reports: layer2/3 stride-1 3x3 forward and data-gradient on the persistent
    8-wave kernel (256x128 tiles) — also the stride-2 forward launches that open layer2/3 —, the other stride-2 / 1x1 launches on <128,128,2>, layer4 on its 256x64 tiles, the generic weight gradient on
    128-wide tiles."""
    from syncvsr_amd import ops

    assert ops.conv_plan(0, N_FRAMES, 11, 11, 128, 3, 1, 1).label == "k_igemm_p8<256,128,3>"
    assert ops.conv_plan(0, N_FRAMES, 6, 6, 256, 3, 1, 1).label == "k_igemm_p8<256,128,3>"
    assert ops.conv_plan(2, N_FRAMES, 6, 6, 256, 3, 1, 1).label == "k_igemm_p8<256,128,3>"
    assert ops.conv_plan(0, N_FRAMES, 11, 11, 256, 3, 2, 1).label == "k_igemm_p8<256,128,3>"            # stride-2 FORWARD: the same tap classes
    assert "p8" not in ops.conv_plan(2, N_FRAMES, 11, 11, 128, 3, 2, 1).label      # its data gradient: parity classes of 1-2 taps, too short
    assert ops.conv_plan(0, N_FRAMES, 6, 6, 512, 3, 2, 1).label == "k_igemm_p8<256,64,3>"       # layer4.0.conv1 forward: 264 items of 256 x 64
    assert ops.conv_plan(0, N_FRAMES, 3, 3, 512, 3, 1, 1).label == "k_igemm_p8<256,64,3>"
    assert ops.wgrad_conv_plan(N_FRAMES, 6, 6, 256, 256, 3, 1, 1).bc == 128
    assert ops.wgrad_conv_plan(N_FRAMES, 3, 3, 512, 512, 3, 1, 1).bc == 128


# (rows, K, N, gelu, bias): the linear layers of the LRW encoder + heads at 32 x 30 tokens, and the LRS ones at 16 x 150 frames
LINEARS = [
    (960, 512, 1536, False, True), (960, 512, 512, False, True), (960, 512, 2048, True, True), (960, 2048, 512, False, True),
    (928, 512, 2560, False, True),
    (2400, 768, 2304, False, True), (2400, 768, 3072, False, True), (2400, 3072, 768, False, True), (2400, 768, 5049, False, True),
]


@pytest.mark.parametrize("rows,K,N,gelu,bias", LINEARS)
def test_linear_fwd_dgrad_wgrad_at_benchmark_rows(dev, rows, K, N, gelu, bias):
    from syncvsr_amd import ops

    x = rnd((rows, K), 8)
    w = rnd((N, K), 9, 1 / math.sqrt(K))
    b = torch.randn(N, generator=torch.Generator().manual_seed(10))
    ref = F.linear(x.float(), w.float(), b)
    Np = (N + 7) // 8 * 8
    out, pre = ops.linear_fwd(x.to(dev), w.to(dev), b.to(dev), rows=rows, K=K, N=N, x_pitch=K, out_pitch=Np, gelu=gelu)
    _SEEN[f"linear {rows}x{K}->{N}.fwd"] = ops.rows_plan(rows, 1, 0, 0, N).label
    if rows == 2400 and N == 768:       # many rows, too few 128x128 tiles: the 128x64 shape (LRS 768-wide outputs)
        assert _SEEN[f"linear {rows}x{K}->{N}.fwd"] == "k_igemm_fwd_glds<128,64,3>"
    if gelu:
        check(pre[:, :N], ref, "linear.pre")
        check(out[:, :N], F.gelu(ref), "linear.gelu")
    else:
        check(out[:, :N], ref, "linear")
    # data gradient through the transposed (64-padded) shadow
    dy = torch.zeros(rows, Np, dtype=BF)
    dy[:, :N] = rnd((rows, N), 12)
    Npad = (N + 63) // 64 * 64
    wt = torch.zeros(K, 1, Npad, dtype=BF)
    wt[:, 0, :N] = w.t()
    dyp = torch.zeros(rows, Npad, dtype=BF)
    dyp[:, :N] = dy[:, :N]
    dx = ops.linear_dgrad(dyp.to(dev), wt.to(dev), rows=rows, N=N, K=K, dy_pitch=Npad)
    _SEEN[f"linear {rows}x{K}->{N}.dgrad"] = ops.rows_plan(rows, 1, 0, 0, K).label
    check(dx, dy[:, :N].float() @ w.float(), "linear.dgrad")
    dw = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    ops.linear_wgrad(x.to(dev), dyp.to(dev), dw, rows=rows, K=K, N=N, x_pitch=K, dy_pitch=Npad, db=db)
    wp = ops.wgrad_rows_plan(rows, 1, 0, 0, K, N, True)
    _SEEN[f"linear {rows}x{K}->{N}.wgrad"] = f"{wp.label} x{wp.splits} splits"
    check(dw, dy[:, :N].float().t() @ x.float(), "linear.wgrad", 3e-3, 2e-3)
    check(db, dy[:, :N].float().sum(0), "linear.bias_grad", 1e-2, 6e-3)
    dw2, db2 = torch.zeros_like(dw), torch.zeros_like(db)
    ops.linear_wgrad(x.to(dev), dyp.to(dev), dw2, rows=rows, K=K, N=N, x_pitch=K, dy_pitch=Npad, db=db2)
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "linear.wgrad is not reproducible"


def test_stem_at_benchmark_batch(dev):
    """Stem conv forward (+statistics) and weight gradient on a full B = 32 batch of 29 x 88 x 88 clips (the persistent grids
    and slab counts of the benchmark launch)."""
    from syncvsr_amd import ops

    B, T, H, W = 32, 29, 88, 88
    g = torch.Generator().manual_seed(20)
    vid = torch.randn(B, 1, T, H, W, generator=g)
    w = (torch.rand(64, 1, 5, 7, 7, generator=g) - 0.5) * 0.2
    out, stats = ops.stem_conv_fwd(vid.to(dev), w.to(dev).reshape(-1), want_stats=True)
    ref = F.conv3d(vid.to(BF).float(), w.to(BF).float(), stride=(1, 2, 2), padding=(2, 3, 3))      # [B,64,T,44,44]
    ref_nhwc = ref.permute(0, 2, 3, 4, 1).reshape(B * T, H // 2, W // 2, 64)
    check(out, ref_nhwc, "stem_conv_fwd@B32")
    buf, rows = stats
    st = buf[: rows * 128].view(rows, 2, 64).double().sum(0).float().cpu()
    check(st[1], (ref * ref).sum((0, 2, 3, 4)), "stem.stats.sumsq", 3e-3, 2e-3)
    dy = rnd((B * T, H // 2, W // 2, 64), 21)
    ws = w.to(BF).float().clone().requires_grad_(True)
    F.conv3d(vid.to(BF).float(), ws, stride=(1, 2, 2), padding=(2, 3, 3)).backward(dy.float().view(B, T, H // 2, W // 2, 64).permute(0, 4, 1, 2, 3))
    dw = torch.zeros(64 * 245, device=dev)
    ops.stem_conv_wgrad(vid.to(dev), dy.to(dev), dw, use_tr=True)
    check(dw.view(64, 1, 5, 7, 7), ws.grad, "stem_conv_wgrad@B32", 4e-3, 3e-3)
    dw2 = torch.zeros_like(dw)
    ops.stem_conv_wgrad(vid.to(dev), dy.to(dev), dw2, use_tr=True)
    assert torch.equal(dw, dw2), "stem wgrad is not reproducible"


# the data-gradient launches of the B = 32 step that carry the first pass of a BatchNorm backward in their epilogue (ReLU trunk):
# every 3x3 data gradient; the expected instantiation is the one bench.py lists as "<kernel>+bn"
EXPECT_DGRAD_BN = {
    "layer1.conv": "k_conv3x3_c64",
    "layer2.conv": "k_igemm_p8<256,128,3>",
    "layer3.conv": "k_igemm_p8<256,128,3>",
    "layer3.0.conv1": "k_igemm_fwd_glds<128,128,2>",
    "layer4.0.conv1": "k_igemm_fwd_glds<128,128,2>",
    "layer4.conv": "k_igemm_p8<256,64,3>",
}


@pytest.mark.parametrize("name", [n for n in TRUNK if TRUNK[n][4] == 3])
@pytest.mark.parametrize("residual", [False, True])
def test_trunk_dgrad_with_bn_backward_epilogue_at_928_frames(dev, name, residual):
    """ops.conv2d_dgrad_bn + bn_bwd_from_stats at the benchmark's shapes == data gradient, then BatchNorm+ReLU backward in fp32."""
    from syncvsr_amd import ops

    H, W, Ci, Co, k, s, p = TRUNK[name]
    N = N_FRAMES
    Ho, Wo = ops.conv_out_size(H, k, s, p), ops.conv_out_size(W, k, s, p)
    dy = rnd((N, Ho, Wo, Co), 3)
    w2 = rnd((Co, k, k, Ci), 4, 1.0 / math.sqrt(k * k * Co))
    wt = w2.permute(3, 1, 2, 0).contiguous()
    xs = torch.zeros(N, Ci, H, W, requires_grad=True)
    F.conv2d(xs, w2.float().permute(0, 3, 1, 2), stride=s, padding=p).backward(nchw(dy.float()))
    add = rnd((N, H, W, Ci), 5) if residual else None
    dout = nhwc(xs.grad) + (add.float() if residual else 0.0)
    xb = rnd((N, H, W, Ci), 6, 2.0) + 0.3
    res = rnd((N, H, W, Ci), 7) if residual else None
    g_ = torch.Generator().manual_seed(8)
    gamma = 1 + 0.2 * torch.randn(Ci, generator=g_)
    beta = 0.2 * torch.randn(Ci, generator=g_)
    xf = xb.float()
    mean, var = xf.mean((0, 1, 2)), xf.var((0, 1, 2), unbiased=False)
    rstd = torch.rsqrt(var + 1e-5)
    m, r = mean.to(dev), rstd.to(dev)
    y_dev = ops.bn_act_fwd(xb.to(dev), None if res is None else res.to(dev), m, r, gamma.to(dev), beta.to(dev), 1)
    mask = y_dev.float().cpu() > 0
    gref = torch.where(mask, dout.to(BF).float(), torch.zeros(()))
    xhat = (xf - mean) * rstd
    cnt = N * H * W
    s1, s2 = gref.sum((0, 1, 2)), (gref * xhat).sum((0, 1, 2))
    dxb_ref = gamma * rstd * (gref - s1 / cnt - xhat * s2 / cnt)
    # residual outputs read the mask from y; residual-free ones recompute it from x (as the model's backward does)
    g, stats = ops.conv2d_dgrad_bn(dy.to(dev), wt.to(dev), k, s, p, (H, W), None if add is None else add.to(dev).clone(),
                                   y_dev if residual else None, xb.to(dev), m, r, gamma.to(dev), beta.to(dev))
    label = "k_conv3x3_c64" if (s == 1 and ops._c64_ok(Ci, Co, k, s, p, W)) else ops.conv_plan(1, N, H, W, Ci, k, s, p).label
    _SEEN[f"{name}.dgrad+bn"] = label
    if name in EXPECT_DGRAD_BN:
        assert label == EXPECT_DGRAD_BN[name], label
    check(g, gref, f"{name}.dgrad_bn.g")
    buf, rows = stats
    sums = buf[: rows * 2 * Ci].view(rows, 2, Ci).double().sum(0).float().cpu()
    check(sums[0], s1, f"{name}.dgrad_bn.sum_g", 1e-2, 6e-3)
    check(sums[1], s2, f"{name}.dgrad_bn.sum_g_xhat", 1e-2, 6e-3)
    coef = torch.empty(3 * Ci, device=dev)
    dg = torch.zeros(Ci, device=dev); db = torch.zeros(Ci, device=dev)
    dxb = ops.bn_bwd_from_stats(g, xb.to(dev), m, r, gamma.to(dev), stats, coef, dg, db)
    check(dxb, dxb_ref, f"{name}.dgrad_bn.dx", 2e-2, 8e-3)
    check(dg, s2, f"{name}.dgrad_bn.dgamma", 1e-2, 6e-3)
    check(db, s1, f"{name}.dgrad_bn.dbeta", 1e-2, 6e-3)
    g2, stats2 = ops.conv2d_dgrad_bn(dy.to(dev), wt.to(dev), k, s, p, (H, W), None if add is None else add.to(dev).clone(),
                                     y_dev if residual else None, xb.to(dev), m, r, gamma.to(dev), beta.to(dev))
    assert torch.equal(g2, g) and torch.equal(stats2[0][: rows * 2 * Ci], buf[: rows * 2 * Ci]), f"{name}: not reproducible"
