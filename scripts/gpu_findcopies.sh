#!/bin/bash
# which host calls produce the per-step device-to-device copies / fills?  (torch profiler with stacks over one eager step)
python - <<'PY' 2>&1 | tail -60
import torch, collections
import syncvsr_amd
from syncvsr_amd.config import default_lrw_config
from syncvsr_amd.engine import TrainStep
from syncvsr_amd.init import synthetic_batch
from syncvsr_amd.model import Model
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
cfg = default_lrw_config(); cfg.train.batch_size = 32
model = Model(cfg, seed=0).to(dev).train()
batch = [t.to(dev) for t in synthetic_batch(cfg, 32, seed=1234)]
tr = TrainStep(model, cfg, use_graph=False)
for _ in range(3): tr.step(*batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(*batch)
    torch.cuda.synchronize()
cnt = collections.Counter(); stacks = collections.defaultdict(collections.Counter)
for ev in prof.events():
    n = ev.name
    if n.startswith("aten::") and any(k in n for k in ("copy_", "fill_", "zero_", "clone", "contiguous", "to", "_to_copy", "empty_like", "zeros")):
        st = [s for s in (ev.stack or []) if "syncvsr_amd" in s or "bench" in s]
        cnt[n] += 1
        stacks[n][st[0] if st else "?"] += 1
for n, c in cnt.most_common(12):
    print(n, c)
    for s, k in stacks[n].most_common(6): print("     ", k, s)
PY
