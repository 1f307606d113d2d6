#!/usr/bin/env python
"""De-phased conv3x3_c64 kernel (tune c64_dephased=1) against the lock-step one (=0): outputs bitwise, statistics to rounding, launch times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from syncvsr_amd import ops

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
torch.manual_seed(0)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def stat_sum(st, C):
    return st[0][: st[1] * 2 * C].view(st[1], 2, C).sum(0).clone()


def run(N, H, W, act=1, reps=30):
    C = 64
    x = (torch.randn(N, H, W, C, device=dev) * 0.5).to(BF16)
    w = (torch.randn(C, 3, 3, C, device=dev) * 0.05).to(BF16)
    dy = (torch.randn(N, H, W, C, device=dev) * 0.5).to(BF16)
    add = (torch.randn(N, H, W, C, device=dev) * 0.5).to(BF16)
    xb = torch.randn(N, H, W, C, device=dev).to(BF16)
    yb = torch.relu(torch.randn(N, H, W, C, device=dev)).to(BF16)
    mean, rstd = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    flops = 2.0 * N * H * W * C * C * 9
    res = {}
    for mode in (0, 1):
        ops.tune("c64_dephased", mode)
        out, st = ops.conv2d_fwd(x, w, 3, 1, 1, want_stats=True)
        t_f = timeit(lambda: ops.conv2d_fwd(x, w, 3, 1, 1, want_stats=True), reps)
        dx = ops.conv2d_dgrad(dy, w, 3, 1, 1, (H, W), addend=add.clone())
        sa = add.clone()
        t_d = timeit(lambda: ops.conv2d_dgrad(dy, w, 3, 1, 1, (H, W), addend=sa), reps)
        g, gst = ops.conv2d_dgrad_bn(dy, w, 3, 1, 1, (H, W), add.clone(), yb, xb, mean, rstd, gamma, beta, act)
        g2, gst2 = ops.conv2d_dgrad_bn(dy, w, 3, 1, 1, (H, W), None, None, xb, mean, rstd, gamma, beta, act)
        t_b = timeit(lambda: ops.conv2d_dgrad_bn(dy, w, 3, 1, 1, (H, W), None, None, xb, mean, rstd, gamma, beta, act), reps)
        sa2 = add.clone()
        t_b2 = timeit(lambda: ops.conv2d_dgrad_bn(dy, w, 3, 1, 1, (H, W), sa2, yb, xb, mean, rstd, gamma, beta, act), reps)
        res[mode] = dict(out=out.clone(), stats=stat_sum(st, C), dx=dx.clone(), g=g.clone(), gs=stat_sum(gst, C), g2=g2.clone(), gs2=stat_sum(gst2, C),
                         t=(t_f, t_d, t_b, t_b2))
    a, b = res[0], res[1]
    ok = True
    for k in ("out", "dx", "g", "g2"):
        same = torch.equal(a[k], b[k])
        ok &= same
        if not same:
            d = (a[k].float() - b[k].float()).abs()
            print(f"  {k}: MISMATCH max {d.max().item():.4g} frac {(d > 0).float().mean().item():.4g}")
    for k in ("stats", "gs", "gs2"):
        rel = ((a[k] - b[k]).abs().max() / a[k].abs().max()).item()
        ok &= rel < 1e-4
        print(f"  {k}: rel diff {rel:.2e}")
    ops.tune("c64_dephased", 1)
    o1, s1 = ops.conv2d_fwd(x, w, 3, 1, 1, want_stats=True)
    s1c = s1[0][: s1[1] * 2 * C].clone()
    o2, s2 = ops.conv2d_fwd(x, w, 3, 1, 1, want_stats=True)
    rep = torch.equal(o1, o2) and torch.equal(s1c, s2[0][: s2[1] * 2 * C])
    print(f"N={N} {H}x{W} act={act}: {'OK' if ok else 'FAIL'} reproducible={rep} | lock-step fwd/dgrad+add/bn(x)/bn(y,add) us {a['t'][0]:.1f} {a['t'][1]:.1f} {a['t'][2]:.1f} {a['t'][3]:.1f} "
          f"({flops / a['t'][0] / 1e6:.0f} TF) | de-phased {b['t'][0]:.1f} {b['t'][1]:.1f} {b['t'][2]:.1f} {b['t'][3]:.1f} ({flops / b['t'][0] / 1e6:.0f} TF)")
    return ok and rep


if __name__ == "__main__":
    if "--trace" in sys.argv:
        import numpy as np
        from syncvsr_amd import _lib
        x = (torch.randn(928, 22, 22, 64, device=dev) * 0.5).to(BF16); w = (torch.randn(64, 3, 3, 64, device=dev) * 0.05).to(BF16)
        ops.tune("c64_dephased", 1); ops.tune("p8_trace", 9)
        for _ in range(3):
            ops.conv2d_fwd(x, w, 3, 1, 1, want_stats=True)
        torch.cuda.synchronize()
        buf = np.zeros(512, dtype=np.int64)
        _lib.load().svsr_debug_c64_trace(buf.ctypes.data)
        ops.tune("p8_trace", 0)
        t = buf.reshape(32, 2, 8)
        base = t[0, 0, 0]
        print("phase | group: start, [MFMA end] or [request done, staged, math done, reduced], waited, barrier passed  (s_memtime ticks, 100 MHz)")
        for ph in range(11):
            print(ph, "| g0", [int(v - base) if v else 0 for v in t[ph, 0]], "| g1", [int(v - base) if v else 0 for v in t[ph, 1]])
        sys.exit(0)
    if "--ablate" in sys.argv:
        x = (torch.randn(928, 22, 22, 64, device=dev) * 0.5).to(BF16); w = (torch.randn(64, 3, 3, 64, device=dev) * 0.05).to(BF16)
        ops.tune("c64_dephased", 1)
        for ab, name in ((0, "product"), (1, "no MFMA"), (2, "no epilogue math / stores"), (3, "no tile DMA"), (4, "no reduce-scatter")):
            ops.tune("p8_trace", ab)
            print(f"{name:28s}: fwd+stats {timeit(lambda: ops.conv2d_fwd(x, w, 3, 1, 1, want_stats=True)):.1f} us   fwd {timeit(lambda: ops.conv2d_fwd(x, w, 3, 1, 1, want_stats=False)):.1f} us")
        ops.tune("p8_trace", 0)
        sys.exit(0)
    good = True
    good &= run(3, 5, 7, reps=5)
    good &= run(16, 22, 22, reps=5)
    good &= run(928, 22, 22)
    if "--lrs" in sys.argv:
        good &= run(2400, 22, 22, act=2, reps=10)
    print("ALL OK" if good else "SOME FAILED")
