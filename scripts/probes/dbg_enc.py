import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import test_gpu_enc_fused as T
dev = torch.device("cuda:0")
model, out = T._tapes(dev, 3, 29, 2, True)
(h0, t0), (h1, t1) = out[False], out[True]
for i in range(2):
    a, b = t0[f"encoder.encoder.layer.{i}"], t1[f"encoder.encoder.layer.{i}"]
    for k in ("qkv", "probs", "ctx", "ao", "m1", "r1", "x1", "z", "hg", "f", "m2", "r2"):
        ne = (a[k] != b[k])
        print(i, k, int(ne.sum()), "of", ne.numel())
    if i == 1:
        ne = (a["hg"] != b["hg"]).nonzero()
        for r, c in ne[:6].tolist():
            print("  hg", r, c, float(a["hg"][r, c]), float(b["hg"][r, c]), "z", float(a["z"][r, c]), float(b["z"][r, c]))
        ne = (a["z"] != b["z"]).nonzero()
        print("  z diffs at", ne[:6].tolist())
        ne = (a["x1"] != b["x1"]).nonzero()
        print("  x1 diffs at", ne[:6].tolist())
