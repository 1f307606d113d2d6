"""Which Python lines issue device-to-device copies / ATen kernels during one eager LRW step?"""
import sys, torch
sys.path.insert(0, '.')
from torch.profiler import profile, ProfilerActivity
from syncvsr_amd.config import default_lrw_config
from syncvsr_amd.engine import TrainStep
from syncvsr_amd.init import synthetic_batch
from syncvsr_amd.model import Model
dev = torch.device('cuda:0')
cfg = default_lrw_config()
model = Model(cfg, seed=0).to(dev).train()
batch = [t.to(dev) for t in synthetic_batch(cfg, 32, seed=1)]
ts = TrainStep(model, cfg, use_graph=False)
for _ in range(3): ts.step(*batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    ts.step(*batch)
torch.cuda.synchronize()
from collections import Counter
c = Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name not in ("aten::empty", "aten::empty_like", "aten::view", "aten::slice", "aten::as_strided", "aten::empty_strided", "aten::select", "aten::reshape", "aten::detach", "aten::alias", "aten::_unsafe_view", "aten::permute", "aten::unflatten", "aten::to", "aten::lift_fresh"):
        st = [s for s in (e.stack or []) if "syncvsr_amd" in s or "bench" in s]
        c[(e.name, tuple(str(x) for x in e.input_shapes)[:2], st[0] if st else "?")] += 1
for k, v in c.most_common(40):
    print(v, k)
