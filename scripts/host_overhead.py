"""How long does the host take to ENQUEUE one eager LRW step (matters for N > 1, where steps are not graph-replayed)?"""
import sys, time, torch
sys.path.insert(0, '.')
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
n = 10
t0 = time.perf_counter()
for _ in range(n): ts.step(*batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/n:.2f} ms/step, total {1e3*(t2-t0)/n:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): ts.step(*batch)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
