#!/usr/bin/env python
"""Launch time of every trunk convolution (forward, data gradient, weight gradient) at the benchmark batch: where the outliers are."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from syncvsr_amd import ops

for _a in sys.argv[1:]:          # tuning knobs: key=value
    _k, _v = _a.split("=")
    ops.tune(_k, int(_v))
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
N = 928
TRUNK = {
    "layer1.conv": (22, 22, 64, 64, 3, 1, 1), "layer2.0.conv1": (22, 22, 64, 128, 3, 2, 1), "layer2.0.downsample": (22, 22, 64, 128, 1, 2, 0),
    "layer2.conv": (11, 11, 128, 128, 3, 1, 1), "layer3.0.conv1": (11, 11, 128, 256, 3, 2, 1), "layer3.0.downsample": (11, 11, 128, 256, 1, 2, 0),
    "layer3.conv": (6, 6, 256, 256, 3, 1, 1), "layer4.0.conv1": (6, 6, 256, 512, 3, 2, 1), "layer4.0.downsample": (6, 6, 256, 512, 1, 2, 0),
    "layer4.conv": (3, 3, 512, 512, 3, 1, 1),
}


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


for name, (H, W, Ci, Co, k, s, p) in TRUNK.items():
    Ho, Wo = ops.conv_out_size(H, k, s, p), ops.conv_out_size(W, k, s, p)
    x = (torch.randn(N, H, W, Ci, device=dev) * 0.5).to(BF16)
    w = (torch.randn(Co, k, k, Ci, device=dev) / math.sqrt(k * k * Ci)).to(BF16)
    wt = w.permute(3, 1, 2, 0).contiguous()
    dy = (torch.randn(N, Ho, Wo, Co, device=dev) * 0.5).to(BF16)
    add = torch.zeros(N, H, W, Ci, device=dev, dtype=BF16)
    dw = torch.zeros(Co, k, k, Ci, device=dev)
    flops = 2.0 * N * Ho * Wo * Co * Ci * k * k
    t_f = timeit(lambda: ops.conv2d_fwd(x, w, k, s, p, want_stats=True))
    t_d = timeit(lambda: ops.conv2d_dgrad(dy, wt, k, s, p, (H, W), addend=add))
    t_d0 = timeit(lambda: ops.conv2d_dgrad(dy, wt, k, s, p, (H, W)))
    t_w = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, k, s, p))
    print(f"{name:22s} {flops / 1e9:6.1f} GF | fwd {t_f:6.1f} us {flops / t_f / 1e6:5.0f} TF | dgrad+add {t_d:6.1f} ({t_d0:6.1f} plain) {flops / t_d / 1e6:5.0f} TF | wgrad {t_w:6.1f} {flops / t_w / 1e6:5.0f} TF"
          f" | in {x.numel() * 2 / 1e6:5.1f} MB out {dy.numel() * 2 / 1e6:5.1f} MB")
