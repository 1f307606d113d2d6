#!/usr/bin/env python
"""Encoder-sized dense layers (960 rows) through svsr_igemm_fwd: time per launch back to back, per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from syncvsr_amd import ops
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
torch.manual_seed(0)

def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

R = 960
shapes = [("qkv fwd", 512, 1536, False), ("ao fwd", 512, 512, False), ("ffn1 fwd+gelu", 512, 2048, True), ("ffn2 fwd", 2048, 512, False), ("qkv dgrad", 1536, 512, False)]
for tag in sys.argv[1:] or ["default"]:
    for kv in filter(None, tag.split(",")):
        if "=" in kv:
            k, v = kv.split("="); ops.tune(k, int(v))
    tot = 0
    out = []
    for name, K, N, gelu in shapes:
        x = (torch.randn(R, K, device=dev) * 0.5).to(BF16); w = (torch.randn(N, K, device=dev) * 0.05).to(BF16); b = torch.randn(N, device=dev)
        y = torch.empty(R, N, dtype=BF16, device=dev)
        t = timeit(lambda: ops.linear_fwd(x, w, b, rows=R, K=K, N=N, x_pitch=K, out=y, gelu=gelu))
        pl = ops.rows_plan(R, 1, 0, 0, N)
        out.append(f"{name} {t:.1f}us ({2.0*R*K*N/t/1e6:.0f} TF, {pl.label})")
        tot += t
    print(tag, "|", " | ".join(out), "| sum", round(tot, 1))
