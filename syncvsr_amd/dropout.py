"""Host-side twin of the library's counter-based dropout (csrc/common.h: `drop_key`, `drop_keep`) and the numbering of the
LRW / LRS models' dropout sites.  The product path only uses `lrw_sites` / `lrs_sites`; `keep_mask` exists so that parity tests (and the
oracle) can replay exactly the masks the kernels generate.

keep(i) <=> mix(i * 2654435761 + key) >= p * 2^32,   key = mix(seed * 0x9E3779B9 + site * 0x7F4A7C15 + 0x165667B1),
mix = murmur3's 32-bit finaliser; kept values are scaled by 1/(1-p).
"""
from __future__ import annotations

import numpy as np

_M = np.uint64(0xFFFFFFFF)


def _mix(h: np.ndarray) -> np.ndarray:
    h = h & _M
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M
    h ^= h >> np.uint64(16)
    return h


def keep_mask(seed: int, site: int, p: float, numel: int) -> np.ndarray:
    """bool [numel]: True where element i is kept."""
    key = _mix(np.array([(seed * 0x9E3779B9 + site * 0x7F4A7C15 + 0x165667B1) & 0xFFFFFFFF], dtype=np.uint64))[0]
    idx = np.arange(numel, dtype=np.uint64)
    h = _mix((idx * np.uint64(2654435761) + key) & _M)
    thresh = min(int(p * 4294967296.0), 4294967295)
    return h >= np.uint64(thresh)


def lrw_sites(layers: int) -> dict[str, int]:
    """Stable site ids of every nn.Dropout the LRW training forward executes with the `type: huggingface` encoder:
    `emb_dropout_bert` on cat(cls, feats) (reference LRW/video/src/lightning.py:45,150); HF BertEmbeddings.dropout,
    BertSelfAttention.dropout (attention probabilities), BertSelfOutput.dropout and BertOutput.dropout (lightning.py:92,152-156)."""
    names = ["emb.in", "emb.out"]
    for i in range(layers):
        names += [f"enc.{i}.{s}" for s in ("attn.probs", "attn.out", "ff.out")]
    return {n: k + 1 for k, n in enumerate(names)}


def lrw_xt_sites(depth: int) -> dict[str, int]:
    """Sites of the `type: x-transformers` encoder: `emb_dropout_bert` (lightning.py:45,150), Attention's dropout on the
    attention probabilities (`attn_dropout`) and FeedForward's dropout after the gate (`ff_dropout`) (lightning.py:95-105)."""
    names = ["emb.in"]
    for i in range(depth):
        names += [f"enc.{i}.attn.probs", f"enc.{i}.ff.hidden"]
    return {n: k + 1 for k, n in enumerate(names)}


def lrs_sites(elayers: int, dlayers: int) -> dict[str, int]:
    """Stable site ids of every nn.Dropout the LRS training forward executes (transformer/embedding.py:208-217,
    encoder_layer.py:97-137, positionwise_feed_forward.py:30, attention.py:80, ctc.py:97, decoder_layer.py:91-113)."""
    names = ["enc.embed.x", "enc.embed.pos", "ctc.in", "dec.embed"]
    for i in range(elayers):
        names += [f"enc.{i}.{s}" for s in ("ffm.hidden", "ffm.out", "attn.probs", "attn.out", "conv.out", "ff.hidden", "ff.out")]
    for i in range(dlayers):
        names += [f"dec.{i}.{s}" for s in ("self.probs", "self.out", "src.probs", "src.out", "ff.hidden", "ff.out")]
    return {n: k + 1 for k, n in enumerate(names)}
