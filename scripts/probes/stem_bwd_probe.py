"""Stem backward at the bench shape: apply + weight-gradient launches vs the fused pass (us, torch events, nothing else on the GPU)."""
import sys, torch
sys.path.insert(0, ".")
from syncvsr_amd import ops

B, T, H, W = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 29, 88, 88)))
dev = torch.device("cuda:0")
C, Ho, Wo = 64, H // 2, W // 2
g = torch.Generator().manual_seed(1)
vid = torch.randn((B, 1, T, H, W), generator=g).to(dev)
x = (1.5 * torch.randn((B * T, Ho, Wo, C), generator=g)).to(torch.bfloat16).to(dev)
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
m = x.float().mean((0, 1, 2)); r = torch.rsqrt(x.float().var((0, 1, 2), unbiased=False) + 1e-5)
y, amax, xwin = ops.stem_bn_gelu_pool_fwd(x, m, r, gamma, beta, want_win=True)
dpool = torch.randn(tuple(y.shape), generator=g).to(torch.bfloat16).to(dev)
coef = torch.empty(3 * C, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
dw = torch.zeros(64 * 245, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def two():
    dx = ops.stem_bn_gelu_pool_bwd(dpool, amax, x, m, r, gamma, beta, coef, dg, db, xwin=xwin)
    ops.stem_conv_wgrad(vid, dx, dw, use_tr=True)


def one():
    gp = ops.stem_bn_gelu_pool_bwd(dpool, amax, x, m, r, gamma, beta, coef, dg, db, xwin=xwin, want_dx=False)
    ops.stem_bwd_wgrad(vid, gp, amax, x, m, r, coef, dw)


def reduce_only():
    ops.stem_bn_gelu_pool_bwd(dpool, amax, x, m, r, gamma, beta, coef, dg, db, xwin=xwin, want_dx=False)


print(f"B={B} T={T} {H}x{W}: two launches {timeit(two):.1f} us, fused {timeit(one):.1f} us, reduce+finalise alone {timeit(reduce_only):.1f} us")
