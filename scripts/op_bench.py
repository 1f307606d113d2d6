#!/usr/bin/env python
"""Per-op micro-benchmark at the BASELINE shapes (B=32 -> 928 frames): times the C-ABI entry points with HIP events.

    python scripts/op_bench.py [wgrad] [fwd] [dgrad] [stem] [ew]      (default: all)
"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from syncvsr_amd import ops  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda:0")
N = 928
LAYERS = [("L1", 22, 64, 64), ("L2", 11, 128, 128), ("L3", 6, 256, 256), ("L4", 3, 512, 512)]


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("OPB_GRAPH", "1") == "1":
        # replay a captured graph so small kernels are not host-launch bound
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters * 1e3
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3     # us


def main():
    which = set(sys.argv[1:]) or {"wgrad", "fwd", "dgrad", "stem", "ew"}
    for kv in os.environ.get("OPB_TUNE", "").split(","):        # e.g. OPB_TUNE=wg_blocks=512,w3_blocks=256
        if "=" in kv:
            k, v = kv.split("=")
            ops.tune(k, int(v))
    tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("SVSR_"))
    print(f"== op_bench {tag}")
    for name, hw, ci, co in LAYERS:
        x = torch.randn(N, hw, hw, ci, device=dev).to(BF)
        dy = torch.randn(N, hw, hw, co, device=dev).to(BF)
        w = (torch.randn(co, 3, 3, ci, device=dev) / math.sqrt(9 * ci)).to(BF)
        wt = w.permute(3, 1, 2, 0).contiguous()
        flops = 2.0 * N * hw * hw * ci * co * 9
        if "fwd" in which:
            us = timeit(lambda: ops.conv2d_fwd(x, w, 3, 1, 1, want_stats=True))
            print(f"{name} conv3x3 fwd   {us:8.1f} us  {flops / us / 1e6:7.1f} TF")
        if "dgrad" in which:
            us = timeit(lambda: ops.conv2d_dgrad(dy, wt, 3, 1, 1, (hw, hw)))
            print(f"{name} conv3x3 dgrad {us:8.1f} us  {flops / us / 1e6:7.1f} TF")
        if "wgrad" in which:
            dw = torch.zeros(co, 3, 3, ci, device=dev)
            us = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, 3, 1, 1))
            print(f"{name} conv3x3 wgrad {us:8.1f} us  {flops / us / 1e6:7.1f} TF")
            ops.HALO_WGRAD = False
            us = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, 3, 1, 1))
            print(f"{name} conv3x3 wgrad(generic) {us:8.1f} us  {flops / us / 1e6:7.1f} TF")
            ops.HALO_WGRAD = True
    if "linear" in which:
        R = 960
        for nm, K, Nn in (("qkv", 512, 1536), ("attn_out", 512, 512), ("ffn1", 512, 2048), ("ffn2", 2048, 512), ("audio", 512, 2560)):
            x = torch.randn(R, K, device=dev).to(BF); w = (torch.randn(Nn, K, device=dev) / math.sqrt(K)).to(BF)
            b = torch.zeros(Nn, device=dev); dy = torch.randn(R, Nn, device=dev).to(BF)
            wt = w.t().contiguous().view(K, 1, Nn)
            fl = 2.0 * R * K * Nn
            us = timeit(lambda: ops.linear_fwd(x, w, b, rows=R, K=K, N=Nn, x_pitch=K), iters=20)
            print(f"linear {nm:9s} fwd   {us:7.1f} us {fl / us / 1e6:7.1f} TF")
            us = timeit(lambda: ops.linear_dgrad(dy, wt, rows=R, N=Nn, K=K, dy_pitch=Nn), iters=20)
            print(f"linear {nm:9s} dgrad {us:7.1f} us {fl / us / 1e6:7.1f} TF")
            dw = torch.zeros(Nn, K, device=dev)
            us = timeit(lambda: ops.linear_wgrad(x, dy, dw, rows=R, K=K, N=Nn, x_pitch=K, dy_pitch=Nn), iters=20)
            print(f"linear {nm:9s} wgrad {us:7.1f} us {fl / us / 1e6:7.1f} TF")
    if "lrs" in which:
        R = int(os.environ.get("OPB_ROWS", "2400"))
        for nm, K, Nn in (("qkv", 768, 2304), ("attn_out", 768, 768), ("ffn1", 768, 3072), ("ffn2", 3072, 768), ("pw1", 768, 1536),
                          ("audio", 768, 2560), ("ctc", 768, 5056)):
            x = torch.randn(R, K, device=dev).to(BF); w = (torch.randn(Nn, K, device=dev) / math.sqrt(K)).to(BF)
            b = torch.zeros(Nn, device=dev); dy = torch.randn(R, Nn, device=dev).to(BF)
            wt = w.t().contiguous().view(K, 1, Nn)
            fl = 2.0 * R * K * Nn
            us = timeit(lambda: ops.linear_fwd(x, w, b, rows=R, K=K, N=Nn, x_pitch=K), iters=20)
            print(f"lrs linear {nm:9s} fwd   {us:7.1f} us {fl / us / 1e6:7.1f} TF")
            us = timeit(lambda: ops.linear_dgrad(dy, wt, rows=R, N=Nn, K=K, dy_pitch=Nn), iters=20)
            print(f"lrs linear {nm:9s} dgrad {us:7.1f} us {fl / us / 1e6:7.1f} TF")
            dw = torch.zeros(Nn, K, device=dev)
            us = timeit(lambda: ops.linear_wgrad(x, dy, dw, rows=R, K=K, N=Nn, x_pitch=K, dy_pitch=Nn), iters=20)
            print(f"lrs linear {nm:9s} wgrad {us:7.1f} us {fl / us / 1e6:7.1f} TF")
    if "mha" in which:
        B, T, H = 16, int(os.environ.get("OPB_T", "150")), 12
        D = H * 64
        qkv = torch.randn(B * T, 3 * D, device=dev).to(BF)
        pe = torch.randn(2 * T - 1, D, device=dev).to(BF)
        u = torch.randn(D, device=dev) * 0.1; v = torch.randn(D, device=dev) * 0.1
        klen = torch.full((B,), T, dtype=torch.int32, device=dev)
        dctx = torch.randn(B * T, D, device=dev).to(BF)
        f = lambda: ops.mha_fwd(qkv, 3 * D, qkv[:, D:], qkv[:, 2 * D:], 3 * D, B=B, H=H, Lq=T, Lk=T, pe=pe, bias_u=u, bias_v=v, klen=klen)
        us = timeit(f)
        print(f"mha rel fwd  {us:8.1f} us  {2.0 * B * H * T * T * 64 * 3 / us / 1e6:6.1f} TF")
        ctx, probs = f()
        dqkv = torch.empty_like(qkv)
        us = timeit(lambda: ops.mha_bwd(dctx, qkv, 3 * D, qkv[:, D:], qkv[:, 2 * D:], 3 * D, probs, B=B, H=H, Lq=T, Lk=T, dq=dqkv, dq_pitch=3 * D,
                                        dk=dqkv[:, D:], dv=dqkv[:, 2 * D:], dkv_pitch=3 * D, pe=pe, bias_u=u, bias_v=v))
        print(f"mha rel bwd  {us:8.1f} us  {2.0 * B * H * T * T * 64 * 7 / us / 1e6:6.1f} TF")
    if "stem" in which:
        vid = torch.randn(32, 1, 29, 88, 88, device=dev)
        w = torch.randn(64 * 245, device=dev) * 0.05
        flops = 2.0 * N * 44 * 44 * 64 * 245
        us = timeit(lambda: ops.stem_conv_fwd(vid, w, want_stats=True))
        print(f"stem conv fwd   {us:8.1f} us  {flops / us / 1e6:7.1f} TF")
        dy = torch.randn(N, 44, 44, 64, device=dev).to(BF)
        dw = torch.zeros(64 * 245, device=dev)
        us = timeit(lambda: ops.stem_conv_wgrad(vid, dy, dw))
        print(f"stem conv wgrad {us:8.1f} us  {flops / us / 1e6:7.1f} TF")
        c = torch.randn(N, 44, 44, 64, device=dev).to(BF)
        mean = torch.zeros(64, device=dev); rstd = torch.ones(64, device=dev); g = torch.ones(64, device=dev); b = torch.zeros(64, device=dev)
        us = timeit(lambda: ops.stem_bn_gelu_pool_fwd(c, mean, rstd, g, b))
        print(f"stem bn+gelu+pool fwd {us:8.1f} us  {c.numel() * 2 * 1.25 / us / 1e6:6.2f} TB/s (algorithmic)")
        y, amax = ops.stem_bn_gelu_pool_fwd(c, mean, rstd, g, b)
        dp = torch.randn_like(y)
        coef = torch.empty(192, device=dev); dg = torch.zeros(64, device=dev); db = torch.zeros(64, device=dev)
        us = timeit(lambda: ops.stem_bn_gelu_pool_bwd(dp, amax, c, mean, rstd, g, b, coef, dg, db))
        print(f"stem bn+gelu+pool bwd {us:8.1f} us")
    if "ew" in which:
        for name, hw, ci, co in LAYERS:
            x = torch.randn(N, hw, hw, co, device=dev).to(BF)
            res = torch.randn_like(x)
            mean = torch.zeros(co, device=dev); rstd = torch.ones(co, device=dev); g = torch.ones(co, device=dev); b = torch.zeros(co, device=dev)
            us = timeit(lambda: ops.bn_act_fwd(x, res, mean, rstd, g, b, 1))
            print(f"{name} bn_act_fwd(+res) {us:7.1f} us  {x.numel() * 2 * 3 / us / 1e6:5.2f} TB/s")
            y = ops.bn_act_fwd(x, res, mean, rstd, g, b, 1)
            coef = torch.empty(3 * co, device=dev)
            dg = torch.zeros(co, device=dev); db = torch.zeros(co, device=dev)
            us = timeit(lambda: ops.bn_act_bwd(res, y, x, mean, rstd, g, coef, dg, db, 1, True))
            print(f"{name} bn_act_bwd(+res) {us:7.1f} us  {x.numel() * 2 * 8 / us / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main()
