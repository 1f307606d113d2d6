#!/usr/bin/env python
"""Where does the HIP forward leave the bf16-storage emulation of the oracle?  Per-stage relative L2 error of the front-end tensors."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from golden_cases import build_case
from oracle import lrw_oracle as O
from syncvsr_amd import model as M

dev = torch.device("cuda:0")
torch.set_num_threads(min(32, os.cpu_count() or 8))
cfg, sd, batch, training, gold = build_case(sys.argv[1] if len(sys.argv) > 1 else "lrw_full_b2")
model = M.Model(cfg); model.load_state_dict(sd, strict=True); model.to(dev).train(True)
st = model.store(); st.refresh_shadows()
tape = {}
with torch.no_grad():
    feats = M._frontend_forward(model, st, tape, batch[0].to(dev).float().contiguous(), True)
torch.cuda.synchronize()
B, _, T = batch[0].shape[:3]


def rel(a, b):
    a, b = a.float().cpu().flatten(), b.float().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def nhwc(t):        # oracle [N,C,H,W] -> [N,H,W,C]
    return t.permute(0, 2, 3, 1).contiguous()


for emu in (False, True):
    keep = {}
    osd = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        O.forward_videos(batch[0], osd, True, {}, keep, emu)
    sc = keep["stem_conv"].transpose(1, 2).flatten(0, 1)          # [B,64,T,H,W] -> [B*T,64,H,W]
    print(f"emu={emu}: stem conv {rel(tape['stem']['c'], nhwc(sc)):.5f}", end=" ")
    print(f"stem out {rel(tape['resnet.layer1.0.conv1']['x'], nhwc(keep['stem_out'])):.5f}", end=" ")
    for pre in ("resnet.layer1.0", "resnet.layer1.1"):
        print(f"| {pre[7:]} c1 {rel(tape[pre + '.conv1']['c'], nhwc(keep[pre + '.conv1.c'])):.5f} y1 {rel(tape[pre + '.conv1']['y'], nhwc(keep[pre + '.bn1.y'])):.5f} "
              f"c2 {rel(tape[pre + '.conv2']['c'], nhwc(keep[pre + '.conv2.c'])):.5f} y {rel(tape[pre + '.conv2']['y'], nhwc(keep[pre + '.out'])):.5f}", end=" ")
        if True:
            hy, oy = tape[pre + '.conv2']['y'].float().cpu(), nhwc(keep[pre + '.out'])
            d = (hy - oy).abs()
            print(f"[y: frac differing {(d > 0).float().mean().item():.4f} max {d.max().item():.4f} frac >1% {(d > 0.01 * oy.abs().clamp(min=0.05)).float().mean().item():.5f} mask flips {((hy > 0) != (oy > 0)).float().mean().item():.5f}]", end=" ")
            hm = tape[pre + '.conv1']['mean'].float().cpu(); c = keep[pre + '.conv1.c']
            om = c.mean((0, 2, 3)); ors = torch.rsqrt(c.var((0, 2, 3), unbiased=False) + 1e-5)
            print(f"mean {((hm - om).abs().max() / om.abs().max()).item():.2e} rstd {((tape[pre + '.conv1']['rstd'].float().cpu() - ors).abs().max() / ors.abs().max()).item():.2e}", end=" ")
    for li in range(1, 5):
        print(f"layer{li} {rel(tape[f'resnet.layer{li}.1.conv2']['y'], nhwc(keep[f'layer{li}'])):.5f}", end=" ")
    print(f"feats {rel(feats, keep['layer4'].mean((2, 3))):.5f}")
