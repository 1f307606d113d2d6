#!/bin/bash
python - <<'PY' 2>&1 | tail -12
import torch, time
import syncvsr_amd
from syncvsr_amd import ops
dev = torch.device("cuda:0")
vid = torch.randn(32, 1, 29, 88, 88, device=dev)
w = torch.randn(64, 1, 5, 7, 7, device=dev) * 0.05
def run(dbg, n=20):
    ops.tune("igemm_lds_pad", dbg)
    for _ in range(3): ops.stem_conv_fwd(vid, w, want_stats=True)
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): ops.stem_conv_fwd(vid, w, want_stats=True)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for dbg, name in ((0, "full"), (1, "no output stores"), (2, "no input fill after the first tile"), (4, "1/18 of the MFMA loop"), (7, "none of the three"), (8, "PIPE full"), (9, "PIPE no output stores"), (10, "PIPE no fetch after the first"), (12, "PIPE 1/18 MFMA"), (15, "PIPE none")):
    print(f"{name:40s} {run(dbg):8.1f} us")
PY
