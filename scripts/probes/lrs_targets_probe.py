"""LRS step with the decoder / CTC targets derived per step (raw labels, ~20 small torch launches on the step's stream) vs prepared once."""
import sys, time, types
sys.path.insert(0, ".")
import torch
import bench
from syncvsr_amd.engine import TrainStep

args = types.SimpleNamespace(dropout=0.1, frames=150, lrs_batch=16)
dev = torch.device("cuda:0")
model, cfg, batch, lrs_args, n_frames, label_len = bench.build_lrs(args, dev, 1, 0)
tr = TrainStep(model, cfg, native=True)
for _ in range(3):
    tr.step(*batch)
bufs = tr.input_buffers()
b_raw = tuple(b if b is not None else o for b, o in zip(bufs, batch))
tg = model.prepare_targets(batch[3].to(dev))
b_pre = (b_raw[0], b_raw[1], b_raw[2], tg)


def run(b, n=12):
    for _ in range(2):
        tr.step(*b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        tr.step(*b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for i in range(3):
    print(f"raw labels {run(b_raw):.3f} ms   prepared targets {run(b_pre):.3f} ms")
