#!/bin/bash
# where does the 1-rank collective path lose time?  One TrainStep per process (stream pool effects), three variants.
export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for variant in plain full noop_collectives noop_allreduce; do
MASTER_PORT=$((29571 + RANDOM % 200)) VARIANT=$variant python - <<'PY' 2>&1 | grep -E " ms|Error|error" | tail -3
import os, time, torch, torch.distributed as dist
import syncvsr_amd
from syncvsr_amd.config import default_lrw_config
from syncvsr_amd.engine import TrainStep
from syncvsr_amd.init import synthetic_batch
from syncvsr_amd.model import Model
v = os.environ["VARIANT"]
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, **({"device_id": dev} if os.environ.get("EAGER") == "1" else {}))
cfg = default_lrw_config(); cfg.train.batch_size = 32
model = Model(cfg, seed=0).to(dev).train()
batch = [t.to(dev) for t in synthetic_batch(cfg, 32, seed=1234)]
tr = TrainStep(model, cfg, use_graph=False, always_reduce=(v != "plain"), data_parallel=(v != "plain"))
if v == "noop_collectives":
    dist.all_reduce = lambda *a, **k: None; dist.broadcast = lambda *a, **k: None
if v == "noop_allreduce":
    dist.all_reduce = lambda *a, **k: None
for _ in range(6): tr.step(*batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): tr.step(*batch)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 40 * 1e3
print(f"{v:20s} {t:.3f} ms", flush=True)
dist.destroy_process_group()
PY
done
