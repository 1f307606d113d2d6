#!/usr/bin/env python
"""LRS linear shapes (2,400 rows): this library's launches next to torch.mm (hipBLASLt), forward / data gradient / weight gradient."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from syncvsr_amd import ops

dev = torch.device("cuda:0")
BF = torch.bfloat16


def t(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for a in sys.argv[1:]:
    k, v = a.split("=")
    ops.tune(k, int(v))
R = int(os.environ.get("ROWS", "2400"))
for nm, K, N in (("qkv", 768, 2304), ("attn_out", 768, 768), ("ffn1", 768, 3072), ("ffn2", 3072, 768), ("pw1", 768, 1536), ("ctc", 768, 5049)):
    Np = (N + 63) // 64 * 64
    x = torch.randn(R, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.03).to(BF); b = torch.randn(N, device=dev)
    wt = torch.zeros(K, Np, device=dev, dtype=BF); wt[:, :N] = w.t()
    dy = torch.randn(R, Np, device=dev).to(BF); dy[:, N:] = 0
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    fl = 2.0 * R * K * N
    f = t(lambda: ops.linear_fwd(x, w, b, rows=R, K=K, N=N, x_pitch=K, out_pitch=Np))
    plan_label = ops._TIMING if False else ""
    d = t(lambda: ops.linear_dgrad(dy, wt, rows=R, N=N, K=K, dy_pitch=Np))
    g = t(lambda: ops.linear_wgrad(x, dy, dw, rows=R, K=K, N=N, x_pitch=K, dy_pitch=Np, db=db))
    wm = torch.randn(N, K, device=dev, dtype=BF); dym = torch.randn(R, N, device=dev, dtype=BF)
    bf = t(lambda: torch.mm(x, wm.t())); bd = t(lambda: torch.mm(dym, wm)); bg = t(lambda: torch.mm(dym.t(), x))
    print(f"{nm:9s} K={K:5d} N={N:5d} | fwd {f:6.1f} us {fl / f / 1e6:5.0f} TF (mm {bf:6.1f} {fl / bf / 1e6:5.0f}) | dgrad {d:6.1f} {fl / d / 1e6:5.0f} (mm {bd:6.1f} {fl / bd / 1e6:5.0f})"
          f" | wgrad {g:6.1f} {fl / g / 1e6:5.0f} (mm {bg:6.1f} {fl / bg / 1e6:5.0f})")
