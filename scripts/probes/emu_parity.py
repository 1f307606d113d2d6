#!/usr/bin/env python
"""HIP gradients against the oracle in both modes (fp32, and bf16-storage emulation: oracle.lrw_oracle.forward(emu=True)) on the full-size
LRW cases.  Prints the distribution the thresholds of tests/test_gpu_model.py are set from.  Run on the GPU box."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from golden_cases import build_case
from oracle import lrw_oracle as O
from syncvsr_amd.model import Model

dev = torch.device("cuda:0")
torch.set_num_threads(min(32, os.cpu_count() or 8))
for name in sys.argv[1:] or ["lrw_full_b2", "lrw_full_b32"]:
    cfg, sd, batch, training, gold = build_case(name)
    model = Model(cfg); model.load_state_dict(sd, strict=True); model.to(dev).train(True)
    out = model(*[t.to(dev) for t in batch]); out["loss_total"].backward(); torch.cuda.synchronize()
    hip = {n: p.grad.detach().float().cpu().flatten().clone() for n, p in model.named_parameters()}
    for emu in (False, True):
        t0 = time.time()
        osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
        keep = {}
        ref = O.forward(osd, cfg, *batch, training=True, keep=keep, emu=emu)
        ref["loss_total"].backward()
        rows = []
        for n, g in hip.items():
            r = osd[n].grad.flatten(); rn = r.norm().item()
            if rn > 1e-6 and not n.endswith("attention.self.key.bias"):      # (softmax is shift-invariant: that gradient is analytically zero)
                rows.append((float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-30)), float(g.norm() / rn), n))
        rows.sort()
        trunk = [r for r in rows if r[2].startswith(("resnet.", "stem3d."))]
        enc = [r for r in rows if not r[2].startswith(("resnet.", "stem3d."))]
        ratios = sorted(r[1] for r in rows)
        tr = sorted(r[1] for r in trunk); er = sorted(r[1] for r in enc)
        print(f"   ratio trunk [{tr[0]:.4f}, {tr[-1]:.4f}] enc [{er[0]:.4f}, {er[-1]:.4f}]; trunk cos p50 {trunk[len(trunk)//2][0]:.5f}; enc cos p50 {enc[len(enc)//2][0]:.5f}")
        def rel(a, b):
            a, b = a.detach().float().cpu().flatten(), b.detach().float().flatten()
            return ((a - b).norm() / (b.norm() + 1e-30)).item()
        print(f"{name} emu={emu} ({time.time() - t0:.0f}s): loss hip {out['loss_total'].item():.5f} oracle {ref['loss_total'].item():.5f} | "
              f"rel feats {rel(model._last['feats'], keep['feats']):.4f} logits_a {rel(model._last['logits_audio'], keep['logits_audio']):.4f} | "
              f"cos min {rows[0][0]:.5f} ({rows[0][2]}) median {rows[len(rows)//2][0]:.5f} | trunk min {trunk[0][0]:.5f} p10 {trunk[len(trunk)//10][0]:.5f} | enc min {enc[0][0]:.5f} | "
              f"ratio [{ratios[0]:.4f}, {ratios[-1]:.4f}]")
        print("   worst:", [(round(c, 4), round(r, 4), n) for c, r, n in rows[:6]])
