"""Where do the fused and the chained encoder backward differ?  (probe; run on the GPU box with PYTHONPATH=.)"""
import sys
import torch
sys.path.insert(0, "tests")
from test_gpu_enc_fused import _tapes
from syncvsr_amd import model as M, ops

dev = torch.device("cuda:0")
B, T, layers = int(__import__("os").environ.get("PB", "3")), 29, int(__import__("os").environ.get("PL", "2"))
model, out = _tapes(dev, B, T, layers, True)
h, tape = out[True]
st = model.store()
dh = (torch.randn(h.shape, generator=torch.Generator().manual_seed(5 + B)) * 1e-2).to(torch.bfloat16).to(dev)
cap = {"seq": []}
orig = {n: getattr(ops, n) for n in ("enc_bwd", "mha_bwd", "scale_bf16", "add_ln_bwd", "bias_act_bwd", "linear_dgrad")}
def wrap(name):
    def f(*a, **k):
        r = orig[name](*a, **k)
        if name == "enc_bwd":
            cap["recs"] = a[1]
        elif name == "mha_bwd":
            cap["seq"].append(("dqkv", k["dq"]))
        else:
            cap["seq"].append((name, r))
        return r
    return f
for n in orig:
    setattr(ops, n, wrap(n))
for fused in (False, True):
    ops.ENC_BWD_FUSED = fused
    st.zero_grad(); st.rebind_grads(); model._wg_group = None
    M._encoder_backward(model, st, tape, dh, B, T)
    torch.cuda.synchronize()
S = T + 1
# chain order per layer: add_ln_bwd (ds2), scale_bf16 (df), linear_dgrad (dhg), bias_act_bwd (dz), linear_dgrad (dx1), add_ln_bwd (ds1), scale_bf16 (dao),
# linear_dgrad (dctx), dqkv, linear_dgrad (dx)
names = ["ds2", "dhg", "dz", "dx1", "ds1", "dctx", "dqkv", "dx"]
seq = [t for _, t in cap["seq"]]
print([n for n, _ in cap["seq"]][:12])
for li, i in enumerate(reversed(range(layers))):
    rec = cap["recs"][i]
    for k, nm in enumerate(names):
        if nm not in rec:
            continue
        a = seq[li * 8 + k].float()
        b = rec[nm].float()
        a3, b3 = a.view(B, S, -1), b.view(B, S, -1)
        d = (a3 - b3).abs()
        bad = (d[0] > 0).nonzero()
        if bad.numel():
            r_, c_ = bad[0].tolist()
            print("     chain", float(a3[0, r_, c_]), "fused", float(b3[0, r_, c_]))
        print(f"layer {i} {nm:5s} seqs that differ", [s_ for s_ in range(B) if float((a3[s_] - b3[s_]).abs().max()) > 0][:8], "n diff", int((d > 0).sum()),
              "first", bad[:4].tolist())
a = seq[6].float().view(B, S, -1); b = cap["recs"][layers - 1]["dqkv"].float().view(B, S, -1)
d = (a - b).abs()
idx = (d > 0).nonzero()
print("first-layer dqkv differences (seq, row, col -> part, head, d):", [(int(s_), int(r_), int(c_), "qkv"[int(c_) // 512], int(c_) % 512 // 64, int(c_) % 64, float(a[s_, r_, c_]), float(b[s_, r_, c_])) for s_, r_, c_ in idx[:14]])
