#!/bin/bash
python - <<'PY' 2>&1 | tail -6
import torch
import syncvsr_amd
from syncvsr_amd import ops
from syncvsr_amd.config import default_lrw_config
from syncvsr_amd.init import synthetic_batch
from syncvsr_amd.model import Model
dev = torch.device("cuda:0")
cfg = default_lrw_config(); cfg.train.batch_size = 2
model = Model(cfg, seed=0).to(dev).train()
batch = [t.to(dev) for t in synthetic_batch(cfg, 2, seed=1)]
model(*batch); st = model.store()
ops.cast_bf16(st.flat, st.w16)
a = torch.zeros_like(st.w16t); b = torch.zeros_like(st.w16t)
ops.transpose_cast_multi(st.flat, a, st.table, st.n_entries)
ops.tune("transpose_from_bf16", 1); ops.transpose_shadows(st.flat, st.w16, b, st.table, st.n_entries)
torch.cuda.synchronize(); print("bf16-source refresh identical:", torch.equal(a, b), a.numel())
for flag in (0, 1, 0, 1):
    ops.tune("transpose_from_bf16", flag)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    for _ in range(3): ops.transpose_shadows(st.flat, st.w16, b, st.table, st.n_entries)
    s.record()
    for _ in range(20): ops.transpose_shadows(st.flat, st.w16, b, st.table, st.n_entries)
    e.record(); torch.cuda.synchronize(); print("from_bf16", flag, f"{s.elapsed_time(e) / 20 * 1e3:.1f} us")
PY
