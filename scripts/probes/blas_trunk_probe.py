"""What does the vendor GEMM (torch.mm -> hipBLASLt) reach on the trunk convolutions written as plain GEMMs (im2col operand materialised)?
A yardstick for k_igemm_fwd_glds on the same M x N x K: python scripts/probes/blas_trunk_probe.py"""
import torch
dev = torch.device("cuda:0")
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for name, M, N, K in (("layer1 conv", 928 * 484, 64, 576), ("layer2 conv", 928 * 121, 128, 1152), ("layer3 conv", 928 * 36, 256, 2304), ("layer4 conv", 928 * 9, 512, 4608),
                      ("qkv", 960, 1536, 512), ("ffn1", 960, 2048, 512), ("ffn2", 960, 512, 2048), ("audio", 928, 2560, 512)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    us = t(lambda: torch.mm(a, b.t()))
    print(f"{name:12s} M={M:7d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
