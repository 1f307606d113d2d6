"""Golden for TimeMask (reference LRW/video/src/augment.py:120-141).  RUNS ONLY IN THE BUILD CONTAINER.
Runs the reference's TimeMask under fixed `random` seeds on a small clip and stores the outputs; tests/test_augment_cpu.py
replays syncvsr_amd.augment.TimeMask with the same seeds."""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/LRW/video/src")
from augment import TimeMask  # noqa: E402  (the reference's)

res = {}
g = torch.Generator().manual_seed(3)
clip = torch.randn(29, 4, 5, generator=g)
res["clip"] = clip.numpy()
for seed, kw in ((1, dict(T=15, n_mask=1)), (2, dict(T=15, n_mask=2)), (3, dict(T=40, n_mask=1, replace_with_zero=True)), (4, dict(T=15, n_mask=1))):
    random.seed(seed)
    res[f"out_{seed}"] = TimeMask(**kw)(clip).numpy()
np.savez_compressed(os.path.join(HERE, "timemask.npz"), **res)
print({k: v.shape for k, v in res.items()})
