"""CPU: syncvsr_amd.augment.CutMix reproduces the reference's sequential in-place CutMix exactly under the same torch seed
(golden produced by tests/golden/make_golden_cutmix.py from the imported reference)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cutmix.npz")


@pytest.mark.parametrize("seed", [7, 8, 9])
def test_cutmix_matches_reference(seed):
    from syncvsr_amd.augment import CutMix

    gold = np.load(GOLD)
    B, T, A, G, C = 6, 9, 4, 2, 11
    videos = (torch.arange(B).view(B, 1, 1, 1, 1) * 100 + torch.arange(T).view(1, 1, T, 1, 1)).float().expand(B, 1, T, 2, 3).clone()
    audios = (torch.arange(B).view(B, 1, 1) * 1000 + torch.arange(T * A).view(1, T * A, 1) * 2 + torch.arange(G).view(1, 1, G)).clone()
    labels = torch.arange(B) % C
    word_mask = (torch.arange(T).view(1, T) >= torch.arange(B).view(B, 1)).long()
    torch.manual_seed(seed)
    v, a, l, w = CutMix(C)(videos, audios, labels, word_mask)
    assert np.array_equal(v.numpy(), gold[f"videos_{seed}"])
    assert np.array_equal(a.numpy(), gold[f"audios_{seed}"])
    np.testing.assert_allclose(l.numpy(), gold[f"labels_{seed}"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(w.numpy(), gold[f"word_mask_{seed}"], rtol=0, atol=1e-7)
    assert any((gold[f"videos_{s}"][:, 0, :, 0, 0] // 100 != np.arange(B)[:, None]).any() for s in (7, 8, 9)), "goldens must exercise a splice"


@pytest.mark.parametrize("seed,kw", [(1, dict(T=15, n_mask=1)), (2, dict(T=15, n_mask=2)), (3, dict(T=40, n_mask=1, replace_with_zero=True)),
                                     (4, dict(T=15, n_mask=1))])
def test_timemask_matches_reference(seed, kw):
    """syncvsr_amd.augment.TimeMask == the reference's TimeMask under the same `random` seed (golden from the imported reference)."""
    import random

    from syncvsr_amd.augment import TimeMask

    gold = np.load(os.path.join(os.path.dirname(GOLD), "timemask.npz"))
    clip = torch.from_numpy(gold["clip"])
    random.seed(seed)
    out = TimeMask(**kw)(clip)
    np.testing.assert_allclose(out.numpy(), gold[f"out_{seed}"], rtol=0, atol=1e-7)
    assert not np.array_equal(gold[f"out_{seed}"], gold["clip"]) or seed == 0


def test_clip_pipeline_draws_are_valid_windows():
    """DeviceClipPipeline.draw: RandomResizedCrop windows lie inside the stored frame with area in [0.6, 1] and ratio in
    [3/4, 4/3]; evaluation draws the centre crop without a flip."""
    from syncvsr_amd.augment import DeviceClipPipeline

    p = DeviceClipPipeline(88, train=True, seed=3).draw(500, 96, 112).numpy()
    top, left, h, w, flip = p.T
    assert (top >= 0).all() and (left >= 0).all() and (top + h <= 96).all() and (left + w <= 112).all()
    area = h * w / (96 * 112)
    assert area.min() >= 0.58 and area.max() <= 1.0 and 0.74 <= (w / h).min() and (w / h).max() <= 1.34
    assert 0.4 < flip.mean() < 0.6
    e = DeviceClipPipeline(88, train=False).draw(3, 96, 112).numpy()
    assert (e == np.array([4, 12, 88, 88, 0])).all()
