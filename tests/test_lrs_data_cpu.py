"""Length-bucketed data-parallel batching for LRS (BASELINE configs[4]; hook: reference LRS/video/datamodule/data_module.py:66-74)."""
import numpy as np
import pytest
import torch

from syncvsr_amd.lrs_data import LengthBucketBatchSampler, collate_pad, pad, reference_length_histogram


@pytest.mark.parametrize("world,drop_last", [(1, True), (8, True), (4, False)])
def test_bucket_sampler_equal_padded_length_across_ranks(world, drop_last):
    lengths = reference_length_histogram(5000, seed=3)
    assert lengths.min() >= 12 and lengths.max() <= 155 and abs(lengths.mean() - 84.7) < 4
    samplers = [LengthBucketBatchSampler(lengths, 16, world, r, width=16, seed=7, drop_last=drop_last) for r in range(world)]
    for s in samplers:
        s.set_epoch(2)
    per_rank = [list(s) for s in samplers]
    n_steps = len(per_rank[0])
    assert all(len(p) == n_steps == len(samplers[0]) for p in per_rank)           # every rank runs the same number of steps
    frames = samplers[0].padded_frames()
    assert frames == samplers[-1].padded_frames() and len(frames) == n_steps
    seen = []
    for step in range(n_steps):
        bound = frames[step]
        for r in range(world):
            idx = per_rank[r][step]
            assert len(idx) == 16
            assert bound - 16 < lengths[idx].max() <= bound or lengths[idx].max() <= bound   # all clips fit the step's padded length
            assert (np.ceil(lengths[idx] / 16) * 16 == bound).all()                           # ... and come from ONE bucket
            seen += idx
    if drop_last:
        assert len(seen) == len(set(seen))                                       # no clip twice, ranks disjoint
        assert len(seen) >= len(lengths) - 10 * world * 16                        # at most one partial global batch per bucket dropped
    else:
        assert set(seen) == set(range(len(lengths)))                             # every clip is visited
    assert samplers[0].padding_waste() < 0.12                                    # 16-frame buckets waste < 12 % of the frames
    # a different epoch reshuffles; the same epoch is reproducible
    again = list(LengthBucketBatchSampler(lengths, 16, world, 0, width=16, seed=7, drop_last=drop_last))
    samplers[0].set_epoch(0)
    assert list(samplers[0]) == again
    samplers[0].set_epoch(1)
    assert list(samplers[0]) != again


def test_collate_pad_layout_matches_reference_contract():
    g = torch.Generator().manual_seed(0)
    batch = [dict(input=torch.randn(t, 1, 8, 8, generator=g), target=torch.randint(1, 50, (l,), generator=g), audio=None)
             for t, l in ((12, 3), (31, 7), (20, 5))]
    out = collate_pad(batch)
    assert out["inputs"].shape == (3, 31, 1, 8, 8) and out["input_lengths"].tolist() == [12, 31, 20]
    assert out["targets"].shape == (3, 1, 7) and out["target_lengths"].tolist() == [3, 7, 5]       # [B, 1, L] padded with -1
    assert (out["targets"][0, 0, 3:] == -1).all() and (out["inputs"][0, 12:] == 0).all()
    assert "audios" not in out
    fixed = collate_pad(batch, pad_frames_to=32)
    assert fixed["inputs"].shape == (3, 32, 1, 8, 8) and torch.equal(fixed["inputs"][:, :31], out["inputs"])
    with pytest.raises(ValueError):
        pad([torch.zeros(5), torch.zeros(9)], pad_to=8)


def test_dataloader_integration():
    lengths = reference_length_histogram(300, seed=1)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(lengths)

        def __getitem__(self, i):
            return dict(input=torch.full((int(lengths[i]), 1, 4, 4), float(i)), target=torch.arange(1, 4))

    sampler = LengthBucketBatchSampler(lengths, 4, world_size=2, rank=1, width=32, seed=0)
    frames = sampler.padded_frames()
    dl = torch.utils.data.DataLoader(DS(), batch_sampler=sampler, collate_fn=lambda b: b)
    n = 0
    for step, b in enumerate(dl):
        out = collate_pad(b, pad_frames_to=frames[step])
        assert out["inputs"].shape[:2] == (4, frames[step])
        n += 1
    assert n == len(sampler)
