#!/bin/bash
# fused vs separate BatchNorm backward, each against the fp32 CPU oracle at B = 32: which one is closer?
cd tests
python - <<'PY' 2>&1 | tail -8
import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.dirname(os.getcwd()))
from golden_cases import build_case
from oracle import lrw_oracle as O
import syncvsr_amd
from syncvsr_amd import ops
from syncvsr_amd.model import Model
dev = torch.device("cuda:0")
cfg, sd, batch, training, gold = build_case("lrw_full_b32")
gb = [t.to(dev) for t in batch]
def grads(fused):
    ops.tune("bn_bwd_fused", int(fused))
    model = Model(cfg); model.load_state_dict(sd, strict=True); model.to(dev).train(True)
    out = model(*gb); out["loss_total"].backward(); torch.cuda.synchronize()
    return {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
ga, gbb = grads(True), grads(False)
osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
torch.set_num_threads(min(32, os.cpu_count() or 8))
ref = O.forward(osd, cfg, *batch, training=True)
ref["loss_total"].backward()
def stats(g):
    cs, rs = [], []
    for n, v in g.items():
        r = osd[n].grad
        if r is None or float(r.norm()) < 1e-10: continue
        cs.append(float(torch.nn.functional.cosine_similarity(v.flatten().double(), r.flatten().double(), dim=0)))
        rs.append(float(v.norm() / r.norm()))
    return min(cs), float(np.median(cs)), min(rs), max(rs)
print("fused    vs oracle: cos min %.4f median %.5f  norm ratio %.3f..%.3f" % stats(ga))
print("separate vs oracle: cos min %.4f median %.5f  norm ratio %.3f..%.3f" % stats(gbb))
for n in ("stem3d.0.weight", "resnet.layer1.0.conv1.weight", "resnet.layer2.0.conv1.weight"):
    r = osd[n].grad.flatten().double()
    c = lambda g: float(torch.nn.functional.cosine_similarity(g[n].flatten().double(), r, dim=0))
    print(n, "cos fused %.5f separate %.5f" % (c(ga), c(gbb)))
PY
