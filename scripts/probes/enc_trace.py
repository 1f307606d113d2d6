#!/usr/bin/env python
"""Phase timeline of the fused encoder forward (svsr_enc_fwd): s_memtime stamps of workgroup 0 at the phase boundaries + launch time."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from syncvsr_amd import _lib, model as M, ops
from syncvsr_amd.config import default_lrw_config
from syncvsr_amd.init import init_state_dict

dev = torch.device("cuda:0")
B, T, layers = int(os.environ.get("ENC_B", "32")), 29, 6
cfg = default_lrw_config()
model = M.Model(cfg, seed=3); model.load_state_dict(init_state_dict(cfg, seed=11, perturb_norm=True)); model.to(dev).train(True)
st = model.store(); st.refresh_shadows()
feats = (torch.randn(B * T, 512, device=dev) * 0.7).to(torch.bfloat16)
model._advance_dropout(dev)
def run():
    tape = {}
    return M._encoder_forward(model, st, tape, feats, B, T)
for fused in (True, False):
    ops.ENC_FUSED = fused
    for _ in range(5): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    print("fused" if fused else "chain", round(s.elapsed_time(e) / 20 * 1e3, 1), "us per encoder forward (incl. embed)")
ops.ENC_FUSED = True
lib = _lib.load()
lib.svsr_debug_enc_trace(None, 0)
run()
n = 1 + 12 * layers
buf = (ctypes.c_int64 * 300)()
lib.svsr_debug_enc_trace(buf, 300)
t = list(buf)
f = [x for x in t[200:260] if x]
print("layer 1 fine stamps (ticks from its P1-gemm-done stamp):", [x - t[1 + 12] for x in f], "| [2] attention done at", t[2 + 12] - t[1 + 12], "| [6] barrier2 passed", t[6 + 12] - t[1 + 12], "[7] LN1 done", t[7 + 12] - t[1 + 12])
names = ["P1 gemm", "attention", "signal1", "wait1", "P2 gemm+epi", "wait2(+sig)", "LN1", "P3 gemm+epi", "wait3(+sig)", "P4 gemm+epi", "wait4(+sig)", "LN2"]
tot = {}
for l in range(layers):
    d = [t[1 + 12 * l + i] - t[12 * l + i] for i in range(12)]
    print(l, " ".join(f"{n}={x}" for n, x in zip(names, d)))
    for n_, x in zip(names, d): tot[n_] = tot.get(n_, 0) + x
print("sum over layers (ticks):", tot, "total", t[n - 1] - t[0])
