"""Length-bucketed data-parallel batching for the LRS (sentence-level) model — BASELINE.json configs[4].

The reference pads every batch to its longest clip (`collate_pad`, LRS/video/datamodule/data_module.py:12-43) and leaves the
place for a batch sampler commented out (`# batch_sampler=sampler`, data_module.py:66-74): with a plain DataLoader the padded
length differs from rank to rank, so under DistributedDataParallel every step waits for the rank that drew the longest clip,
and every new padded length is a new set of kernel shapes.  `LengthBucketBatchSampler` is the sampler for that hook:

  * clips are grouped into buckets of similar length (bucket = ceil(length / width) * width frames);
  * one training step takes `world_size * batch_size` clips from ONE bucket and deals them to the ranks, so every rank pads
    to the SAME number of frames in the same step (no stragglers) and only len(buckets) different shapes ever occur;
  * the order of steps and the clips inside a bucket are shuffled from (seed, epoch) identically on every rank — no
    communication — and every rank sees the same number of steps (a requirement of the gradient all-reduce).

`collate_pad` reproduces the reference's batch layout and can pad the frame axis to the bucket's bound instead of the longest
clip in the batch.  Pure host code (numpy / torch CPU); nothing here touches the GPU.
"""
from __future__ import annotations

from typing import Iterator, Optional, Sequence

import numpy as np
import torch


class LengthBucketBatchSampler:
    """`batch_sampler` for `torch.utils.data.DataLoader`: yields this rank's list of dataset indices per step.

    lengths     clip lengths in frames, one per dataset item (e.g. the reference's datamodule/video_length.npy, 12..155)
    batch_size  clips per rank and step (`batch_size: 16`, LRS/video/config/lrs3.yaml:1)
    width       bucket width in frames; the padded length of a step is a multiple of it
    max_frames  clips longer than this are treated as max_frames long (the dataset crops them, av_dataset.py:72-80)
    drop_last   drop the clips of a bucket that do not fill a whole global batch (True keeps every rank's batch full; False
                deals the remainder round-robin and pads short ranks by repeating clips of the same bucket)
    """

    def __init__(self, lengths: Sequence[int], batch_size: int, world_size: int = 1, rank: int = 0, width: int = 16,
                 max_frames: Optional[int] = None, seed: int = 0, shuffle: bool = True, drop_last: bool = True):
        if not 0 <= rank < world_size:
            raise ValueError("rank must be in [0, world_size)")
        if batch_size < 1 or width < 1:
            raise ValueError("batch_size and width must be positive")
        self.lengths = np.asarray(lengths, dtype=np.int64)
        if max_frames is not None:
            self.lengths = np.minimum(self.lengths, int(max_frames))
        self.batch_size, self.world, self.rank, self.width = int(batch_size), int(world_size), int(rank), int(width)
        self.seed, self.shuffle, self.drop_last = int(seed), bool(shuffle), bool(drop_last)
        self.epoch = 0
        self.bounds = ((self.lengths + self.width - 1) // self.width) * self.width       # padded length of each clip's bucket
        self._buckets = {int(b): np.nonzero(self.bounds == b)[0] for b in np.unique(self.bounds)}

    def set_epoch(self, epoch: int) -> None:
        """Same contract as DistributedSampler.set_epoch: call once per epoch on every rank with the same value."""
        self.epoch = int(epoch)

    def _steps(self) -> list[tuple[int, np.ndarray]]:
        """[(padded frames, global batch of world*batch indices)] in this epoch's order — identical on every rank."""
        rng = np.random.default_rng([self.seed, self.epoch])
        G = self.world * self.batch_size
        steps: list[tuple[int, np.ndarray]] = []
        for bound in sorted(self._buckets):
            idx = self._buckets[bound]
            if self.shuffle:
                idx = rng.permutation(idx)
            full = len(idx) // G
            for s in range(full):
                steps.append((bound, idx[s * G:(s + 1) * G]))
            rest = idx[full * G:]
            if len(rest) and not self.drop_last:
                fill = idx[rng.integers(0, len(idx), G - len(rest))] if self.shuffle else np.resize(idx, G - len(rest))
                steps.append((bound, np.concatenate([rest, fill])))
        if self.shuffle:
            order = rng.permutation(len(steps))
            steps = [steps[i] for i in order]
        return steps

    def __iter__(self) -> Iterator[list[int]]:
        for _, g in self._steps():
            yield [int(i) for i in g[self.rank::self.world]]          # dealt round-robin: neighbours in the shuffle go to different ranks

    def __len__(self) -> int:
        G = self.world * self.batch_size
        n = 0
        for idx in self._buckets.values():
            n += len(idx) // G + (1 if (len(idx) % G and not self.drop_last) else 0)
        return n

    def padded_frames(self) -> list[int]:
        """Padded clip length of every step of the current epoch (the same list on every rank)."""
        return [b for b, _ in self._steps()]

    def padding_waste(self) -> float:
        """Fraction of padded frames that are padding, over the current epoch."""
        real = padded = 0
        for bound, g in self._steps():
            real += int(self.lengths[g].sum())
            padded += bound * len(g)
        return 1.0 - real / max(padded, 1)


def pad(samples: list[torch.Tensor], pad_val: float = 0.0, pad_to: Optional[int] = None) -> tuple[torch.Tensor, list[int]]:
    """Stack variable-length samples along a new batch axis, padding axis 0 of each to the longest (or to `pad_to`).
    Layout as the reference's `pad` (data_module.py:12-32): 1-D samples (targets) come back as [B, 1, L]."""
    lengths = [len(s) for s in samples]
    size = max(lengths) if pad_to is None else int(pad_to)
    if size < max(lengths):
        raise ValueError(f"pad_to={size} is shorter than the longest sample ({max(lengths)})")
    out = samples[0].new_full([len(samples), size] + list(samples[0].shape[1:]), pad_val)
    for i, s in enumerate(samples):
        out[i, : len(s)] = s
    if samples[0].dim() == 1:
        out = out.unsqueeze(1)
    return out, lengths


def collate_pad(batch: list[dict], pad_frames_to: Optional[int] = None, frames_per_unit: Optional[dict] = None) -> dict:
    """The reference's `collate_pad` (data_module.py:34-43): {"inputs", "input_lengths", "targets", "target_lengths", ...} with
    targets padded by -1 and everything else by 0.  `pad_frames_to` (the sampler's bucket bound) pads "input" — and every key
    listed in `frames_per_unit` at its own rate, e.g. {"audio": 640} samples or {"audio": 4} tokens per video frame — to a fixed
    length, so that all ranks of a step produce the same shapes."""
    out = {}
    for key in batch[0].keys():
        vals = [s[key] for s in batch if s[key] is not None]
        if not vals:
            continue
        to = None
        if pad_frames_to is not None:
            if key == "input":
                to = pad_frames_to
            elif frames_per_unit and key in frames_per_unit:
                to = pad_frames_to * int(frames_per_unit[key])
        c, lens = pad(vals, -1 if key == "target" else 0.0, to)
        out[key + "s"] = c
        out[key + "_lengths"] = torch.tensor(lens)
    return out


def reference_length_histogram(n: int, seed: int = 0, lo: int = 12, hi: int = 155, mean: float = 84.7) -> np.ndarray:
    """Synthetic clip lengths with the range and mean of the reference's training set (video_length.npy: 31,982 clips,
    min 12 / max 155 / mean 84.7 frames) — for benchmarks and tests; the real file cannot travel."""
    rng = np.random.default_rng(seed)
    a = (mean - lo) / (hi - lo)
    x = rng.beta(2.0 * a / (1 - a) if a < 0.5 else 2.0, 2.0 if a < 0.5 else 2.0 * (1 - a) / a, n)
    return np.clip(np.round(lo + x * (hi - lo)), lo, hi).astype(np.int64)
