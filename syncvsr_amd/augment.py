"""Sequential-frame CutMix of the reference (LRW/video/src/augment.py:11-118) with a host-side plan and one device gather.

The reference loops over the batch in Python and splices frames IN PLACE, so a later sample can copy frames that an
earlier iteration already replaced.  Here the same random decisions are drawn on the host in the same order from torch's
CPU generator (so a seeded run reproduces the reference bit for bit), the in-place chain is resolved on the host into a
[B, T] source map (which original sample each frame finally comes from), and the clip / token / mask tensors are produced
by a single gather on the device — no per-sample launches, no device->host sync.

Quirks kept on purpose: the audio tokens are spliced with the FRAME indices (not frame*alignment) exactly as
augment.py:104-109 does; labels become [B, num_labels] probabilities and word masks become float.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F


class CutMixPlan:
    def __init__(self, frame_src: torch.Tensor, token_src: torch.Tensor, target_ids: torch.Tensor, mix: torch.Tensor):
        self.frame_src = frame_src      # int64 [B, T]   original sample whose frame t ends up in sample b
        self.token_src = token_src      # int64 [B, Ta]  same for the audio-token axis
        self.target_ids = target_ids    # int64 [B]
        self.mix = mix                  # float32 [B]    effective mix rate (0 where no cut happened)


def make_plan(batch_size: int, frames: int, token_steps: int, generator: Optional[torch.Generator] = None) -> CutMixPlan:
    """Draws exactly the random numbers CutMix.forward / mask_video draw, in the same order (augment.py:34-36,86,97-99)."""
    target_ids = torch.randint(0, batch_size, (batch_size,), generator=generator)
    target_rates = torch.rand(batch_size, generator=generator)
    frame_src = torch.arange(batch_size).unsqueeze(1).repeat(1, frames)
    token_src = torch.arange(batch_size).unsqueeze(1).repeat(1, token_steps)
    mix = torch.zeros(batch_size)
    for i in range(batch_size):
        rate = target_rates[i].item()
        cut_out_flag = torch.randint(0, 2, (1,), generator=generator)[0].item()
        if cut_out_flag != 1:
            continue
        cut = int(frames * rate)
        if cut <= 0:
            continue
        start = torch.randint(0, frames - cut, (1,), generator=generator).item()
        t = int(target_ids[i])
        # in-place semantics: the target's CURRENT frames (already spliced if t < i, or i itself) are copied
        frame_src[i, start : start + cut] = frame_src[t, start : start + cut].clone()
        hi = min(start + cut, token_steps)
        if start < token_steps:
            token_src[i, start:hi] = token_src[t, start:hi].clone()
        mix[i] = rate
    return CutMixPlan(frame_src, token_src, target_ids, mix)


class CutMix(torch.nn.Module):
    """Drop-in for the reference's ``CutMix(num_labels, wav2vec=None)`` on pre-tokenised audio."""

    def __init__(self, num_labels: int, wav2vec=None) -> None:
        super().__init__()
        if wav2vec is not None:
            raise NotImplementedError("on-the-fly wav2vec tokenisation is outside the hot path; pass audio tokens")
        self.num_labels = num_labels
        self.generator: Optional[torch.Generator] = None     # None = torch's global CPU generator, as the reference uses

    def forward(self, videos: torch.Tensor, audios: torch.Tensor, labels: torch.Tensor, word_mask: torch.Tensor):
        B, _, T = videos.shape[:3]
        plan = make_plan(B, T, audios.shape[1], self.generator)
        dev = videos.device
        fsrc = plan.frame_src.to(dev, non_blocking=True)
        tsrc = plan.token_src.to(dev, non_blocking=True)
        tidx = torch.arange(T, device=dev).unsqueeze(0).expand(B, T)
        mixed_videos = videos[fsrc, :, tidx].permute(0, 2, 1, 3, 4).contiguous()          # [B, 1, T, H, W]
        aidx = torch.arange(audios.shape[1], device=dev).unsqueeze(0).expand(B, -1)
        mixed_audios = audios[tsrc, aidx].contiguous()
        mix = plan.mix.to(dev)
        tgt = plan.target_ids.to(dev)
        one_hot = F.one_hot(labels, self.num_labels)
        # integer one-hots promoted by the float mix rate, as in augment.py:111-113; untouched samples stay integer there,
        # the concatenation makes the batch float — here everything is float32 from the start
        mixed_labels = (1.0 - mix).unsqueeze(1) * one_hot.float() + mix.unsqueeze(1) * one_hot[tgt].float()
        wm = word_mask.float()
        mixed_word_mask = (1.0 - mix).view(B, *([1] * (wm.dim() - 1))) * wm + mix.view(B, *([1] * (wm.dim() - 1))) * wm[tgt]
        return mixed_videos, mixed_audios, mixed_labels, mixed_word_mask


def draw_time_runs(n_frames: int, max_run, n_mask: int) -> list[tuple[int, int]]:
    """(start, length) of every masked run, drawn from Python's `random` with the two calls per mask the reference makes
    (LRW/video/src/augment.py:130-132): the run length first, then where it starts."""
    import random

    runs = []
    for _ in range(n_mask):
        length = random.randint(0, min(max_run, n_frames))
        runs.append((random.randint(0, n_frames - length), length))
    return runs


class TimeMask(torch.nn.Module):
    """The reference's TimeMask (LRW/video/src/augment.py:120-141; `train.use_timemask`, T = 0.6 * 25 frames, one mask):
    random runs of up to T frames of one clip [T, ...] are replaced by the clip's mean (or zero).  `draw_time_runs` makes the
    random decisions in the reference's order, so a seeded run reproduces it; every run is then one fill on the clip's device
    (CPU or HIP; no host sync: the mean stays a 0-d tensor)."""

    def __init__(self, T: float = 6400, n_mask: int = 2, replace_with_zero: bool = False):
        super().__init__()
        self.T, self.n_mask, self.replace_with_zero = T, n_mask, replace_with_zero

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        out = x.clone()
        for start, length in draw_time_runs(out.size(0), self.T, self.n_mask):
            # the mean is that of the clip as masked so far: a later run sees the earlier fills, as in the reference
            out[start : start + length] = 0 if self.replace_with_zero else out.mean()
        return out


class DeviceClipPipeline:
    """The reference's per-clip transform chain (LRW/video/src/data.py:150,157-171) on the device in one kernel: stored uint8 mouth
    crops [B, T, Hs, Ws] -> fp32 [B, 1, T, H, W] = Normalize(RandomResizedCrop | CenterCrop(flip(x / 255))) — the host only draws
    five integers per clip, and the uint8 clips cross PCIe at a quarter of the fp32 size.

    train=True:  RandomHorizontalFlip(0.5) and, with use_rrc, RandomResizedCrop(size, scale=(0.6, 1.0), ratio=(3/4, 4/3)) with
                 torchvision's rejection sampling (10 tries, then the central fallback); without use_rrc the frame passes through
                 at its stored size (nn.Identity, data.py:160).
    train=False: CenterCrop(size) (or Resize with use_val_resize).
    The draws come from a numpy generator (torchvision, whose RNG order the reference follows, is not available to pin them);
    the resize is bilinear without antialias."""

    def __init__(self, size: int, train: bool, use_rrc: bool = True, use_val_resize: bool = False, mean: float = 0.421,
                 std: float = 0.165, seed: int = 0):
        import numpy as np

        self.size, self.train, self.use_rrc, self.use_val_resize, self.mean, self.std = int(size), train, use_rrc, use_val_resize, mean, std
        self.rng = np.random.default_rng(seed)

    def draw(self, B: int, Hs: int, Ws: int) -> torch.Tensor:
        """int32 [B, 5] = (top, left, h, w, flip) per clip."""
        import math

        import numpy as np

        out = np.zeros((B, 5), dtype=np.int32)
        for b in range(B):
            if self.train and self.use_rrc:
                area = Hs * Ws
                for _ in range(10):
                    target = area * self.rng.uniform(0.6, 1.0)
                    ratio = math.exp(self.rng.uniform(math.log(3 / 4), math.log(4 / 3)))
                    w, h = int(round(math.sqrt(target * ratio))), int(round(math.sqrt(target / ratio)))
                    if 0 < w <= Ws and 0 < h <= Hs:
                        out[b, :4] = (self.rng.integers(0, Hs - h + 1), self.rng.integers(0, Ws - w + 1), h, w)
                        break
                else:       # torchvision's fallback (RandomResizedCrop.get_params): the central crop with the nearest admissible ratio
                    in_ratio = Ws / Hs
                    if in_ratio < 3 / 4:
                        w, h = Ws, int(round(Ws / (3 / 4)))
                    elif in_ratio > 4 / 3:
                        h = Hs
                        w = int(round(h * (4 / 3)))
                    else:
                        w, h = Ws, Hs
                    out[b, :4] = ((Hs - h) // 2, (Ws - w) // 2, h, w)
            elif (self.train and not self.use_rrc) or (not self.train and self.use_val_resize):
                out[b, :4] = (0, 0, Hs, Ws)
            else:
                out[b, :4] = ((Hs - self.size) // 2, (Ws - self.size) // 2, self.size, self.size)
            out[b, 4] = int(self.train and self.rng.random() < 0.5)
        return torch.from_numpy(out)

    def __call__(self, frames_u8: torch.Tensor, params: Optional[torch.Tensor] = None) -> torch.Tensor:
        from . import ops

        B, T, Hs, Ws = frames_u8.shape
        if params is None:
            params = self.draw(B, Hs, Ws)
        if params.device.type == "cpu":         # (device-resident windows are clamped into the frame by the kernel instead)
            top, left, h, w = (params[:, i] for i in range(4))
            if bool(((top < 0) | (left < 0) | (h < 1) | (w < 1) | (top + h > Hs) | (left + w > Ws)).any()):
                raise ValueError("crop window outside the stored frame")
        identity = self.train and not self.use_rrc          # nn.Identity: output size = stored size
        H, W = (Hs, Ws) if identity else (self.size, self.size)
        return ops.clip_prep(frames_u8.contiguous(), params.to(frames_u8.device, non_blocking=True), H, W, self.mean, self.std)
