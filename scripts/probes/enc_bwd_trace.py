#!/usr/bin/env python
"""Phase timeline of the fused encoder backward (svsr_enc_bwd): s_memtime stamps (100 MHz ticks) of workgroup 0 at the phase boundaries + launch time."""
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
tape = {}
h = M._encoder_forward(model, st, tape, feats, B, T)
dh = (torch.randn(h.shape, device=dev) * 1e-2).to(torch.bfloat16)
model._ablate = None
M._ABLATE = frozenset({"lin_wgrad"})          # time the data path alone
def run():
    st.zero_grad(); model._wg_group = None
    return M._encoder_backward(model, st, tape, dh, B, T)
for fused in (True, False):
    ops.ENC_BWD_FUSED = fused
    for _ in range(5): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    print("fused" if fused else "chain", round(s.elapsed_time(e) / 20 * 1e3, 1), "us per encoder backward without weight gradients (incl. zero_grad, embed backward)")
ops.ENC_BWD_FUSED = True
lib = _lib.load()
lib.svsr_debug_enc_trace(None, 0)
run()
buf = (ctypes.c_int64 * 300)()
lib.svsr_debug_enc_trace(buf, 300)
t = list(buf)
names = ["wait4/start", "LN2 bwd", "B2 gemm+gelu'", "wait1(+sig)", "B4 gemm", "wait2(+sig)", "LN1 bwd", "B6 gemm+stage", "attn bwd", "wait3(+sig)", "B8 gemm"]
tot = {}
for l in range(layers):
    d = [t[1 + 11 * l + i] - t[11 * l + i] for i in range(11)]
    print(l, " ".join(f"{n}={x}" for n, x in zip(names, d)))
    for n_, x in zip(names, d): tot[n_] = tot.get(n_, 0) + x
print("sum over layers (ticks of 10 ns):", tot, "total", t[11 * layers] - t[0])
