#!/usr/bin/env python
"""Halo weight gradient (svsr_conv3x3_wgrad) with the dense contraction (w3_dense=1) against the padded walk (0): results and launch time at the
benchmark shapes (layer1 22 x 22 x 64, layer2 11 x 11 x 128 at 928 frames; the sentence-level front-end at 2,560 frames)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from syncvsr_amd import ops
dev = torch.device("cuda:0")
BF16 = torch.bfloat16

def timeit(fn, n=20):
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

for N, H, C in ((928, 22, 64), (928, 11, 128), (2560, 22, 64), (2560, 11, 128), (37, 22, 64), (3, 11, 128)):
    g = torch.Generator(device="cpu").manual_seed(1)
    x = (torch.randn(N, H, H, C, generator=g) * 0.5).to(BF16).to(dev)
    dy = (torch.randn(N, H, H, C, generator=g) * 0.5).to(BF16).to(dev)
    res = {}
    for mode in (0, 1):
        ops.tune("w3_dense", mode)
        dw = torch.zeros(C, 3, 3, C, device=dev)
        ops.conv2d_wgrad(x, dy, dw, 3, 1, 1)
        t = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, 3, 1, 1))
        dw.zero_()
        ops.conv2d_wgrad(x, dy, dw, 3, 1, 1)
        res[mode] = (dw.clone(), t)
    a, b = res[0][0], res[1][0]
    rel = float((a - b).norm() / a.norm())
    flops = 2.0 * N * H * H * C * C * 9
    print(f"N={N:5d} {H}x{H}x{C}: padded {res[0][1]:7.1f} us ({flops / res[0][1] / 1e6:5.0f} TF)  dense {res[1][1]:7.1f} us ({flops / res[1][1] / 1e6:5.0f} TF)  rel diff {rel:.2e}")
ops.tune("w3_dense", 1)
