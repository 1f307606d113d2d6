#!/usr/bin/env python
"""In-loop cycles per 64-deep K tile of k_igemm_p8 at the benchmark shapes (build variant p8stamp: SVSR_LIB_VARIANT=p8stamp): the number the
producer / consumer probe (scripts/probes/pc_probe.hip) is compared with."""
import ctypes, os, sys, math
os.environ["SVSR_LIB_VARIANT"] = "p8stamp"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from syncvsr_amd import _lib, ops
lib = ctypes.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
out2 = (ctypes.c_longlong * 8)()
GRIDS = [int(a) for a in sys.argv[1:]] or [0]          # p8_grid values to sweep (0 = one workgroup per CU): is the epilogue bound per CU or by the chip?
for name, (N, H, C) in {"layer2.conv": (928, 11, 128), "layer3.conv": (928, 6, 256), "layer4.conv": (928, 3, 512), "lrs layer2": (2560, 11, 128), "lrs layer3": (2560, 6, 256)}.items():
    x = (torch.randn(N, H, H, C, device=dev) * 0.5).to(BF16)
    w = (torch.randn(C, 3, 3, C, device=dev) / math.sqrt(9 * C)).to(BF16)
    dy = (torch.randn(N, H, H, C, device=dev) * 0.5).to(BF16)
    xb = torch.randn(N, H, H, C, device=dev).to(BF16)
    yb = torch.relu(torch.randn(N, H, H, C, device=dev)).to(BF16)
    add = (torch.randn(N, H, H, C, device=dev) * 0.5).to(BF16)
    mean, rstd, gamma, beta = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    runs = {"fwd+stats": lambda: ops.conv2d_fwd(x, w, 3, 1, 1, want_stats=True),
            "dgrad+add": lambda: ops.conv2d_dgrad(dy, w, 3, 1, 1, (H, H), addend=add),
            "dgrad+bn(res)": lambda: ops.conv2d_dgrad_bn(dy, w, 3, 1, 1, (H, H), add, yb, xb, mean, rstd, gamma, beta, 1)}
    keep = {}
    for what, fn in [(f"{k} grid={d}", (d, k, f)) for k, f in runs.items() for d in GRIDS]:
        ops.tune("p8_grid", fn[0])
        res = fn[2]()                     # bit-identity of the two epilogues: outputs and raw partial rows
        flat = [t.clone() for t in (res if isinstance(res, tuple) else (res,)) if torch.is_tensor(t)] + \
               [res[1][0][: res[1][1] * 2 * C].clone()] if isinstance(res, tuple) and isinstance(res[1], tuple) else [res.clone()]
        if fn[1] in keep:
            same = all(torch.equal(a, b) for a, b in zip(keep[fn[1]], flat))
            print(f"    {fn[1]}: same bits as the first grid: {same}")
        keep[fn[1]] = flat
        fn = fn[2]
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        lib.svsr_debug_p8_stamps(out2)
        reps = 10
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        lib.svsr_debug_p8_stamps(out2)
        print(f"{name:12s} {what:20s} {s.elapsed_time(e) / reps * 1e3:7.1f} us per launch; in-loop {out2[0] / max(1, out2[1]):6.0f} cycles per K tile ({out2[1] // reps} K tiles), "
              f"epilogue {out2[2] / max(1, out2[3]):7.0f} cycles per tile ({out2[3] // reps} tiles); phases per tile: head {out2[4] / max(1, out2[3]):6.0f}, row block 0 {out2[5] / max(1, out2[3]):6.0f}, row block 1 {out2[6] / max(1, out2[3]):6.0f}; per workgroup (of 256): loops {out2[0] / reps / 256 / 2.1e3:6.1f} us + epilogues {out2[2] / reps / 256 / 2.1e3:6.1f} us @2.1 GHz")
