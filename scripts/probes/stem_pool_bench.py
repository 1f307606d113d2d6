#!/usr/bin/env python
"""Stem BN + GELU + max-pool forward / backward at the benchmark shape, with and without the kept winners (same process, same box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from syncvsr_amd import ops

dev = torch.device("cuda:0")
N, Hc, Wc, C = int(os.environ.get("FRAMES", "928")), 44, 44, 64
torch.manual_seed(0)
c = torch.randn(N, Hc, Wc, C, device=dev).to(torch.bfloat16)
mean, rstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
coef = torch.empty(3 * C, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)


def t(fn, n=20):
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


for rep in range(2):
    f0 = t(lambda: ops.stem_bn_gelu_pool_fwd(c, mean, rstd, g, b))
    f1 = t(lambda: ops.stem_bn_gelu_pool_fwd(c, mean, rstd, g, b, want_win=True))
    y, amax, xwin = ops.stem_bn_gelu_pool_fwd(c, mean, rstd, g, b, want_win=True)
    dp = torch.randn_like(y)
    b0 = t(lambda: ops.stem_bn_gelu_pool_bwd(dp, amax, c, mean, rstd, g, b, coef, dg, db))
    b1 = t(lambda: ops.stem_bn_gelu_pool_bwd(dp, amax, c, mean, rstd, g, b, coef, dg, db, xwin=xwin))
    print(f"forward {f0:.1f} us, with winners {f1:.1f} us | backward (3 launches) gather form {b0:.1f} us, winner form {b1:.1f} us | sum {f0 + b0:.1f} -> {f1 + b1:.1f}")
