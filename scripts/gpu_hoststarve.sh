#!/bin/bash
# is the GPU ever waiting for the host?  A spin kernel in front of every step lets the host run ahead; step time minus spin time = pure GPU time
python - <<'PY' 2>&1 | tail -12
import time, torch
import syncvsr_amd
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
def run(spin, n=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        if spin: torch.cuda._sleep(spin)
        tr.step(*batch)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def spin_only(spin, n=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): torch.cuda._sleep(spin)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    base = run(0)
    for spin in (2_000_000, 6_000_000, 12_000_000):
        s = spin_only(spin); t = run(spin)
        print(f"plain {base:.3f} ms | spin {s:.3f} ms: step+spin {t:.3f} -> step alone {t - s:.3f} ms")
PY
