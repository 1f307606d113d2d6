#!/bin/bash
# host enqueue time of one step, plain vs 1-rank RCCL path (is the collective path host-bound?)
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29581 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for variant in plain full full32 full64 full128; do
VARIANT=$variant python - <<'PY' 2>&1 | grep -E "steady|Error|error" | tail -3
import os, time, torch, torch.distributed as dist
import syncvsr_amd
from syncvsr_amd.config import default_lrw_config
from syncvsr_amd.engine import TrainStep
from syncvsr_amd.init import synthetic_batch
from syncvsr_amd.model import Model
v = os.environ["VARIANT"]
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfg = default_lrw_config(); cfg.train.batch_size = 32
model = Model(cfg, seed=0).to(dev).train()
batch = [t.to(dev) for t in synthetic_batch(cfg, 32, seed=1234)]
mb = {"full": 16.0, "full32": 32.0, "full64": 64.0, "full128": 128.0}.get(v, 16.0)
tr = TrainStep(model, cfg, use_graph=False, always_reduce=(v != "plain"), data_parallel=(v != "plain"), bucket_mb=mb)
for _ in range(6): tr.step(*batch)
hs, ts = [], []
for _ in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr.step(*batch); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    hs.append(t1 - t0); ts.append(t2 - t0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): tr.step(*batch)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 40 * 1e3
print(f"{v:8s} steady {t:.3f} ms | one step from idle: host enqueue {sorted(hs)[6]*1e3:.2f} ms, total {sorted(ts)[6]*1e3:.2f} ms", flush=True)
dist.destroy_process_group()
PY
done
