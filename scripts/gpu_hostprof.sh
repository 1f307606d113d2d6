#!/bin/bash
# host-side cost of the eager step: cProfile over 30 steps (GPU work is asynchronous; the profile shows who spends the enqueue time),
# plus per-phase host vs GPU clocks (is the GPU ever waiting for the host?)
mkdir -p gpurun_out
python - <<'PY' 2>&1 | tail -90
import cProfile, pstats, time, torch, io
import syncvsr_amd
from syncvsr_amd import ops
from syncvsr_amd.config import default_lrw_config
from syncvsr_amd.engine import TrainStep
from syncvsr_amd.init import synthetic_batch
from syncvsr_amd.model import Model
dev = torch.device("cuda:0")
cfg = default_lrw_config(); cfg.train.batch_size = 32
model = Model(cfg, seed=0).to(dev).train()
batch = [t.to(dev) for t in synthetic_batch(cfg, 32, seed=1234)]
tr = TrainStep(model, cfg, use_graph=False)
for _ in range(5): tr.step(*batch)
torch.cuda.synchronize()
# host enqueue time per step with an idle GPU queue at the start
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr.step(*batch); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("host enqueue ms / total ms (one step at a time):", [f"{a*1e3:.2f}/{b*1e3:.2f}" for a, b in ts[-4:]])
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(30): tr.step(*batch)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO(); st = pstats.Stats(pr, stream=s); st.sort_stats("tottime").print_stats(32); print(s.getvalue()[:9000])
PY
