"""Library GEMM speed (torch.mm -> hipBLASLt/rocBLAS) at the LRS linear shapes, for comparison with the hand-written igemm kernels."""
import torch
dev = torch.device("cuda:0")
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
R = 2400
for nm, K, N in (("qkv", 768, 2304), ("attn_out", 768, 768), ("ffn1", 768, 3072), ("ffn2", 3072, 768), ("ctc", 768, 5056)):
    x = torch.randn(R, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(R, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * R * K * N
    us = t(lambda: torch.mm(x, w.t())); print(f"{nm:9s} fwd   {us:7.1f} us {fl/us/1e6:7.1f} TF")
    us = t(lambda: torch.mm(dy, w)); print(f"{nm:9s} dgrad {us:7.1f} us {fl/us/1e6:7.1f} TF")
    us = t(lambda: torch.mm(dy.t(), x)); print(f"{nm:9s} wgrad {us:7.1f} us {fl/us/1e6:7.1f} TF")
