#!/usr/bin/env python
"""Layer1 / layer2 halo weight gradient (svsr_conv3x3_wgrad) at the LRW and LRS frame counts: launch time and effective operand rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from syncvsr_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for tune in sys.argv[1:]:
    k, v = tune.split("="); ops.tune(k, int(v))
for N in (232, 464, 928, 1392, 1856, 2560):
    for (H, C) in ((22, 64), (11, 128)):
        x = (torch.randn(N, H, H, C, device=dev) * 0.5).to(BF)
        dy = (torch.randn(N, H, H, C, device=dev) * 0.5).to(BF)
        dw = torch.zeros(C, 3, 3, C, device=dev)
        fl = 2.0 * N * H * H * C * C * 9
        t = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, 3, 1, 1))
        mb = 2 * x.numel() * 2 / 1e6
        print(f"N={N:5d} {H}x{H}x{C}: {t:7.1f} us  {fl / t / 1e6:5.0f} TF  operands {mb:6.1f} MB -> {mb / t:5.2f} TB/s", flush=True)
