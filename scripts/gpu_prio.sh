#!/bin/bash
# stream-priority experiment: main step on a high-priority stream, weight-gradient side stream at default priority
python - <<'PY' 2>&1 | tail -14
import time, torch
import syncvsr_amd
from syncvsr_amd.config import default_lrw_config
from syncvsr_amd.engine import TrainStep
from syncvsr_amd.init import synthetic_batch
from syncvsr_amd.model import Model
dev = torch.device("cuda:0")
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
cfg = default_lrw_config(); cfg.train.batch_size = 32
model = Model(cfg, seed=0).to(dev).train()
batch = [t.to(dev) for t in synthetic_batch(cfg, 32, seed=1234)]
tr = TrainStep(model, cfg, use_graph=False)
for _ in range(5): tr.step(*batch)
def run(n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.step(*batch)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
hi = torch.cuda.Stream(priority=-1)
lo = torch.cuda.Stream(priority=0)
print("hi prio", hi.priority, "lo prio", lo.priority)
def run_hi(n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(hi):
        for _ in range(n): tr.step(*batch)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    a = run()
    with torch.cuda.stream(hi):
        for _ in range(3): tr.step(*batch)
    b = run_hi()
    print(f"default streams {a:.3f} ms | main on high-priority stream {b:.3f} ms")
# the other way round: side stream high priority
model._side.stream = hi
for rep in range(2):
    print(f"side stream high priority, main default: {run():.3f} ms")
PY
