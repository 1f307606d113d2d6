"""Which launch of an LRW step waits for the foreign resident kernel?  Run under rocprofv3 --kernel-trace and list the long kernels / gaps."""
import sys, time, torch
sys.path.insert(0, '.')
from syncvsr_amd import _lib, ops
from syncvsr_amd.config import default_lrw_config
from syncvsr_amd.engine import TrainStep
from syncvsr_amd.init import synthetic_batch
from syncvsr_amd.model import Model
lib = _lib.load()
dev = torch.device("cuda:0")
cfg = default_lrw_config(); cfg.train.batch_size = 32
model = Model(cfg, seed=0).to(dev).train()
batch = [t.to(dev) for t in synthetic_batch(cfg, 32, seed=77)]
ts = TrainStep(model, cfg, native="--eager" not in sys.argv)
for _ in range(2):
    ts.step(*batch)
torch.cuda.synchronize()
side = torch.cuda.Stream()
lib.svsr_debug_occupy_start(int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32, 96 * 1024, side.cuda_stream)
time.sleep(0.05)
t0 = time.perf_counter()
out = ts.step(*batch)
torch.cuda.current_stream().synchronize()
print("step beside: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
if "--item" in sys.argv:
    t0 = time.perf_counter()
    v = float(out["loss_total"].item())
    print(".item() beside the co-tenant: %.1f ms" % ((time.perf_counter() - t0) * 1e3), v)
if "--sync" in sys.argv:
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    print("torch.cuda.synchronize() beside the co-tenant: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
lib.svsr_debug_occupy_stop()
torch.cuda.synchronize()
