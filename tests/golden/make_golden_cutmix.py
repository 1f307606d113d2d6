"""Golden for the sequential-frame CutMix (reference LRW/video/src/augment.py:11-118).  RUNS ONLY IN THE BUILD CONTAINER.
Feeds the reference CutMix tensors whose values encode (sample, frame) so its in-place splice chain can be read back, under a
fixed torch seed, and stores the outputs.  tests/test_augment_cpu.py replays syncvsr_amd.augment.CutMix with the same seed."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/LRW/video/src")
from augment import CutMix  # noqa: E402  (the reference's)

B, T, A, G, C = 6, 9, 4, 2, 11
res = {}
for seed in (7, 8, 9):
    videos = (torch.arange(B).view(B, 1, 1, 1, 1) * 100 + torch.arange(T).view(1, 1, T, 1, 1)).float().expand(B, 1, T, 2, 3).clone()
    audios = (torch.arange(B).view(B, 1, 1) * 1000 + torch.arange(T * A).view(1, T * A, 1) * 2 + torch.arange(G).view(1, 1, G)).clone()
    labels = torch.arange(B) % C
    word_mask = (torch.arange(T).view(1, T) >= torch.arange(B).view(B, 1)).long()
    torch.manual_seed(seed)
    v, a, l, w = CutMix(C, None)(videos, audios, labels, word_mask)
    res[f"videos_{seed}"] = v.numpy()
    res[f"audios_{seed}"] = a.numpy()
    res[f"labels_{seed}"] = l.float().numpy()
    res[f"word_mask_{seed}"] = w.float().numpy()
np.savez_compressed(os.path.join(HERE, "cutmix.npz"), **res)
print("saved", {k: v.shape for k, v in res.items() if k.endswith("_7")})
