"""Shared description of the committed golden cases (tests/golden/make_golden_{lrw,lrs}.py import these tables)."""
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
    # BASELINE.json configs[1] at its full batch: the shape bench.py times (B = 32 -> 928 frames), so every kernel instantiation
    # the benchmark launches is the one this case runs.  Too heavy for the CPU suite: pinned on the GPU box (fp32 oracle vs golden).
    "lrw_full_b32": (dict(), dict(batch=32, frames=29, size=88), 0, 1234, False, True),
}

HEAVY_CASES = ("lrw_full_b32", "lrs_full_t150")      # excluded from the `-m "not gpu"` parametrisations (minutes of fp64 CPU work)


def sample_idx(n: int) -> torch.Tensor:
    """16 evenly spaced flat indices (the `sample.*` entries of the goldens).  float32 linspace as the first fixtures used;
    float64 once n - 1 is no longer exactly representable in float32 (the B = 32 stem tensor has 1.15e8 elements)."""
    return torch.linspace(0, n - 1, 16, dtype=torch.float32 if n <= (1 << 24) else torch.float64).long()


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


# ---------------------------------------------------------------------------------------------
# LRS (E2E) cases — mirrored by tests/golden/make_golden_lrs.py, which imports this table.
# name: (arg overrides, odim, batch kwargs, weight seed, data seed, perturb_norm, training)
# ---------------------------------------------------------------------------------------------
_LRS_TINY = dict(adim=128, aheads=2, eunits=256, elayers=2, ddim=128, dheads=2, dunits=256, dlayers=1)
LRS_CASES = {
    "lrs_tiny": (_LRS_TINY, 41, dict(batch=2, t_max=9, size=24, label_len=(2, 4)), 11, 91, True, True),
    "lrs_tiny_b3": (dict(_LRS_TINY, elayers=1, dlayers=2, cnn_module_kernel=7, lsm_weight=0.2, mtlalpha=0.3, audio_weight=2.0,
                         codec="wav2vec2"), 23, dict(batch=3, t_max=14, size=16, label_len=(1, 5), min_len_frac=0.3), 12, 92, True, True),
    "lrs_tiny_eval": (_LRS_TINY, 41, dict(batch=2, t_max=9, size=24, label_len=(2, 4)), 11, 91, True, False),
    "lrs_full_b2": (dict(), 5049, dict(batch=2, t_max=12, size=88, label_len=(3, 6)), 0, 1234, False, True),
    # adim != ddim (proj_decoder, e2e_asr_transformer.py:93-95) and the length-normalised attention loss
    # the shipped 252 M-parameter config at a realistic clip length (two ragged clips padded to 150 frames)
    "lrs_full_t150": (dict(), 5049, dict(batch=2, t_max=150, size=88, label_len=(10, 30), min_len_frac=0.6), 0, 1235, False, True),
    # transformer_input_layer conv3d-lrw: the word-level front-end (GELU stem, ReLU ResNet18) under the sentence-level encoder
    "lrs_tiny_lrwfe": (dict(_LRS_TINY, transformer_input_layer="conv3d-lrw"), 41, dict(batch=2, t_max=9, size=24, label_len=(2, 4)), 14, 94,
                       True, True),
    # mtlalpha = 0: no CTC branch at all (self.ctc = None, loss_ctc = 0; e2e_asr_transformer.py:127-132,205-208)
    "lrs_tiny_noctc": (dict(_LRS_TINY, mtlalpha=0.0), 41, dict(batch=2, t_max=9, size=24, label_len=(2, 4)), 15, 95, True, True),
    "lrs_tiny_proj": (dict(_LRS_TINY, ddim=192, dheads=3, dlayers=2, transformer_length_normalized_loss=True), 37,
                      dict(batch=3, t_max=10, size=16, label_len=(2, 5), min_len_frac=0.5), 13, 93, True, True),
}


# Inference (beam search) cases: (arg overrides, odim, clip frames, clip size, weight seed, data seed, head gain, eos bias, [(beam, ctc_weight)]).
# `head gain` multiplies decoder.output_layer / ctc.ctc_lo so the random-init posteriors are peaked and the n-best order is not a
# coin toss between near-ties.
LRS_INFER_CASES = {
    "lrs_infer_tiny": (dict(_LRS_TINY, dlayers=2), 41, 16, 24, 21, 191, 6.0, 2.5, [(5, 0.1), (30, 0.1), (4, 0.3)]),
    # the shipped model and the reference's own search settings (beam 40, CTC weight 0.1, 5,049 units: pre-beam of 60 candidates per hypothesis)
    "lrs_infer_full": (dict(), 5049, 36, 88, 22, 192, 8.0, 4.0, [(40, 0.1)]),
}


def build_lrs_infer_case(name: str, load_golden: bool = True):
    """-> (args, odim, state_dict (eval, perturbed running statistics), clip [T,1,H,W], runs [(beam, ctc_weight)], golden | None)"""
    from syncvsr_amd.lrs_init import default_lrs_args, lrs_init_state_dict

    over, odim, frames, size, wseed, dseed, gain, eos_bias, runs = LRS_INFER_CASES[name]
    args = default_lrs_args(**over)
    sd = lrs_init_state_dict(args, odim, seed=wseed, perturb_norm=True)
    g = torch.Generator().manual_seed(5)
    for k in list(sd):
        if k.endswith("running_mean"):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
    for k in ("decoder.output_layer.weight", "ctc.ctc_lo.weight"):
        sd[k] = sd[k] * gain
    sd["decoder.output_layer.bias"][odim - 1] += eos_bias      # lets hypotheses END before the length limit (end_detect path)
    clip = torch.randn(frames, 1, size, size, generator=torch.Generator().manual_seed(dseed))
    gold = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False) if load_golden else None
    return args, odim, sd, clip, runs, gold


def build_lrs_case(name: str, load_golden: bool = True):
    """-> (args, odim, state_dict, batch, training, golden npz | None)"""
    from syncvsr_amd.lrs_init import default_lrs_args, lrs_init_state_dict, lrs_synthetic_batch

    over, odim, bkw, wseed, dseed, perturb, training = LRS_CASES[name]
    args = default_lrs_args(**over)
    sd = lrs_init_state_dict(args, odim, seed=wseed, perturb_norm=perturb)
    if not training:
        g = torch.Generator().manual_seed(5)
        for k in list(sd):
            if k.endswith("running_mean"):
                sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
            elif k.endswith("running_var"):
                sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
    batch = lrs_synthetic_batch(args, odim=odim, seed=dseed, **bkw)
    gold = np.load(os.path.join(GOLDEN, f"{name}.npz")) if load_golden else None
    return args, odim, sd, batch, training, gold
