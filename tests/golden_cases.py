"""Shared description of the committed golden cases (must mirror tests/golden/make_golden_lrw.py::CASES)."""
from __future__ import annotations

import os

import numpy as np
import torch

from syncvsr_amd.config import default_lrw_config
from syncvsr_amd.init import init_state_dict, synthetic_batch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = {
    "lrw_full_b2": (dict(), dict(batch=2, frames=29, size=88), 0, 1234, False, True),
    "lrw_tiny": (dict(model__bert__num_hidden_layers=2), dict(batch=2, frames=5, size=24), 1, 77, True, True),
    "lrw_tiny_soft_ls": (dict(model__bert__num_hidden_layers=2, train__label_smoothing=0.1, train__use_cutmix=True),
                         dict(batch=3, frames=4, size=24, soft_labels=True), 2, 78, True, True),
    "lrw_tiny_hard_ls": (dict(model__bert__num_hidden_layers=1, train__label_smoothing=0.1),
                         dict(batch=2, frames=3, size=16), 3, 79, True, True),
    "lrw_tiny_eval": (dict(model__bert__num_hidden_layers=2), dict(batch=2, frames=5, size=24), 1, 77, True, False),
}


def build_case(name: str):
    """-> (cfg, state_dict, batch, training, golden npz)"""
    over, bkw, wseed, dseed, perturb, training = CASES[name]
    cfg = default_lrw_config(**over)
    sd = init_state_dict(cfg, seed=wseed, perturb_norm=perturb)
    if not training:
        g = torch.Generator().manual_seed(5)
        for k in list(sd):
            if k.endswith("running_mean"):
                sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
            elif k.endswith("running_var"):
                sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
    batch = synthetic_batch(cfg, seed=dseed, **bkw)
    gold = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    return cfg, sd, batch, training, gold
