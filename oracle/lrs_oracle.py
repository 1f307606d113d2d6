"""ORACLE — CPU restatement (plain torch fp32) of the SyncVSR LRS training hot path (``E2E.forward``).

TEST INFRASTRUCTURE ONLY.  Nothing under ``syncvsr_amd/`` may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and only as the checker.

Parity status: PINNED.  ``tests/golden/make_golden_lrs.py`` imports the reference's own ``E2E``
(``/root/reference/LRS/video/espnet/nets/pytorch_backend/e2e_asr_transformer.py`` with the import stubs of
SURVEY.md App. C) in the build container, loads the same seeded weights, and records losses, intermediate
activations and gradients in ``tests/golden/lrs_*.npz``; ``tests/test_oracle_golden.py`` checks this restatement
against those vectors.  Everything on this path is in-tree arithmetic of the reference (no third-party model code);
the only deliberate deviation is SURVEY §8(b): pre-computed audio tokens are passed in place of the raw waveform
(the frozen wav2vec quantiser's weights are not available offline), exactly as the golden generator overrides
``forward_audios``.

All dropout probabilities are 0 here (the goldens are generated with dropout 0 so train-mode BatchNorm can be
pinned without RNG).  Every function cites the reference lines it follows (paths relative to
``LRS/video/espnet/nets/pytorch_backend/``).
"""
from __future__ import annotations

import math
from typing import Any

import torch
import torch.nn.functional as F

from .lrw_oracle import batch_norm, layer_norm

Tensor = torch.Tensor
SD = dict[str, Tensor]
LN_EPS = 1e-12                       # transformer/layer_norm.py:19
NEG = -1e10                          # attention.py:73 (min_value)


class DropPlan:
    """Replays the HIP library's counter-based dropout masks (syncvsr_amd/dropout.py is the numpy twin of csrc/common.h's
    drop_keep) so a training forward WITH dropout can be compared element for element: the reference's nn.Dropout draws from
    torch's generator, which no other implementation can reproduce — sharing the mask is the only way to pin the arithmetic
    around it.  `probs_pitch` mirrors the row pitch of the library's probability buffer (keys padded to a multiple of 8)."""

    def __init__(self, seed: int, p: float, attn_p: float, sites: dict[str, int]):
        self.seed, self.p, self.attn_p, self.sites = int(seed), float(p), float(attn_p), sites

    def __call__(self, x: Tensor, site: str, attn: bool = False) -> Tensor:
        from syncvsr_amd.dropout import keep_mask

        p = self.attn_p if attn else self.p
        if p <= 0.0:
            return x
        if attn:
            B, H, Lq, Lk = x.shape
            pitch = (Lk + 7) // 8 * 8
            m = keep_mask(self.seed, self.sites[site], p, B * H * Lq * pitch).reshape(B, H, Lq, pitch)[..., :Lk]
        else:
            m = keep_mask(self.seed, self.sites[site], p, x.numel()).reshape(tuple(x.shape))
        return x * torch.from_numpy(m.copy()).to(x.dtype) / (1.0 - p)


def _dp(dp, x: Tensor, site: str, attn: bool = False) -> Tensor:
    return x if dp is None else dp(x, site, attn)


def swish(x: Tensor) -> Tensor:
    """transformer/convolution.py:78-83."""
    return x * torch.sigmoid(x)


def _ln(x: Tensor, sd: SD, p: str) -> Tensor:
    return layer_norm(x, sd[f"{p}.weight"], sd[f"{p}.bias"], LN_EPS)


def _lin(x: Tensor, sd: SD, p: str) -> Tensor:
    b = sd.get(f"{p}.bias")
    w = sd[f"{p}.weight"]
    return F.linear(x, w.flatten(1) if w.dim() == 3 else w, b)


# --------------------------------------------------------------------------------------------
# visual front-end: backbones/conv3d_extractor.py:40-48, backbones/modules/resnet.py:90-107,163-177
# --------------------------------------------------------------------------------------------
def frontend(x: Tensor, sd: SD, training: bool, stats_out: dict | None = None, keep: dict | None = None) -> Tensor:
    """x [B,T,1,H,W] -> [B,T,512].  Conv3d(5,7,7)/(1,2,2) -> BN3d -> act -> MaxPool(1,3,3)/(1,2,2) -> ResNet18(act) -> avgpool.
    `conv3d` (encoder.frontend.*): Swish everywhere; `conv3d-lrw` (encoder.stem3d / encoder.resnet, encoder.py:132-139,248-255):
    GELU stem and ReLU blocks, the word-level model's front-end."""
    lrw = "encoder.stem3d.0.weight" in sd
    stem, trunk = ("encoder.stem3d", "encoder.resnet") if lrw else ("encoder.frontend.frontend3D", "encoder.frontend.trunk")
    act_stem = (lambda t: 0.5 * t * (1.0 + torch.erf(t / math.sqrt(2.0)))) if lrw else swish
    act = torch.relu if lrw else swish
    B, T = x.shape[:2]
    h = F.conv3d(x.transpose(1, 2), sd[f"{stem}.0.weight"], None, stride=(1, 2, 2), padding=(2, 3, 3))
    h = act_stem(batch_norm(h, sd, f"{stem}.1", training, stats_out))
    h = F.max_pool3d(h, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    h = h.transpose(1, 2).reshape(B * T, 64, h.size(3), h.size(4))
    if keep is not None:
        keep["stem_out"] = h
    for li in range(1, 5):
        for bi in range(2):
            p = f"{trunk}.layer{li}.{bi}"
            stride = 2 if (bi == 0 and li > 1) else 1
            out = F.conv2d(h, sd[f"{p}.conv1.weight"], None, stride=stride, padding=1)
            out = act(batch_norm(out, sd, f"{p}.bn1", training, stats_out))
            out = F.conv2d(out, sd[f"{p}.conv2.weight"], None, stride=1, padding=1)
            out = batch_norm(out, sd, f"{p}.bn2", training, stats_out)
            if f"{p}.downsample.0.weight" in sd:
                res = F.conv2d(h, sd[f"{p}.downsample.0.weight"], None, stride=stride)
                res = batch_norm(res, sd, f"{p}.downsample.1", training, stats_out)
            else:
                res = h
            h = act(out + res)
    return h.mean((2, 3)).view(B, T, 512)


# --------------------------------------------------------------------------------------------
# positional encodings: transformer/embedding.py:54-76 (absolute), :168-217 (relative, "latest")
# --------------------------------------------------------------------------------------------
def sinusoid(positions: Tensor, d_model: int) -> Tensor:
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(positions.numel(), d_model)
    ang = positions.float().unsqueeze(1) * div
    pe[:, 0::2] = torch.sin(ang)
    pe[:, 1::2] = torch.cos(ang)
    return pe


def rel_pos_emb(T: int, d_model: int) -> Tensor:
    """[2T-1, d]: row r encodes relative position (T-1-r), i.e. +T-1 ... 0 ... -(T-1)  (embedding.py:180-216)."""
    return sinusoid(torch.arange(T - 1, -T, -1), d_model)


def abs_pos_emb(T: int, d_model: int) -> Tensor:
    return sinusoid(torch.arange(T), d_model)


# --------------------------------------------------------------------------------------------
# attention: transformer/attention.py:38-108 (MHA), :191-278 (rel-pos MHA, rel_shift :216-236)
# --------------------------------------------------------------------------------------------
def _attend(scores: Tensor, v: Tensor, mask: Tensor | None, dp=None, site: str = "") -> Tensor:
    """attention.py:59-88: mask==0 -> -1e10 before softmax and 0 after; dropout on the probabilities; context = P·V."""
    if mask is not None:
        m = mask.unsqueeze(1).eq(0)
        attn = torch.softmax(scores.masked_fill(m, NEG), dim=-1).masked_fill(m, 0.0)
    else:
        attn = torch.softmax(scores, dim=-1)
    attn = _dp(dp, attn, site, attn=True)
    ctx = torch.matmul(attn, v)                                # [B,H,Tq,dk]
    return ctx.transpose(1, 2).reshape(ctx.size(0), ctx.size(2), -1)


def mha(q_in: Tensor, kv_in: Tensor, mask: Tensor | None, sd: SD, p: str, heads: int, dp=None, site: str = "") -> Tensor:
    B, Tq, D = q_in.shape
    dk = D // heads
    q = _lin(q_in, sd, f"{p}.linear_q").view(B, Tq, heads, dk).transpose(1, 2)
    k = _lin(kv_in, sd, f"{p}.linear_k").view(B, -1, heads, dk).transpose(1, 2)
    v = _lin(kv_in, sd, f"{p}.linear_v").view(B, -1, heads, dk).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
    return _lin(_attend(scores, v, mask, dp, site), sd, f"{p}.linear_out")


def rel_mha(x: Tensor, pos: Tensor, mask: Tensor | None, sd: SD, p: str, heads: int, dp=None, site: str = "") -> Tensor:
    """scores[i,j] = ((q_i+u)·k_j + (q_i+v)·p[T-1+j-i]) / sqrt(dk)   — the closed form of rel_shift (attention.py:216-236)."""
    B, T, D = x.shape
    dk = D // heads
    q = _lin(x, sd, f"{p}.linear_q").view(B, T, heads, dk)
    k = _lin(x, sd, f"{p}.linear_k").view(B, T, heads, dk).transpose(1, 2)
    v = _lin(x, sd, f"{p}.linear_v").view(B, T, heads, dk).transpose(1, 2)
    pe = F.linear(pos, sd[f"{p}.linear_pos.weight"]).view(2 * T - 1, heads, dk).permute(1, 0, 2)     # [H,2T-1,dk]
    qu = (q + sd[f"{p}.pos_bias_u"]).transpose(1, 2)
    qv = (q + sd[f"{p}.pos_bias_v"]).transpose(1, 2)
    ac = torch.matmul(qu, k.transpose(-2, -1))
    bd_full = torch.matmul(qv, pe.transpose(-2, -1))           # [B,H,T,2T-1]
    idx = (T - 1) + torch.arange(T).view(1, T) - torch.arange(T).view(T, 1)                          # [T,T]
    bd = bd_full.gather(-1, idx.expand(B, heads, T, T))
    scores = (ac + bd) / math.sqrt(dk)
    return _lin(_attend(scores, v, mask, dp, site), sd, f"{p}.linear_out")


# --------------------------------------------------------------------------------------------
# Conformer blocks: transformer/convolution.py:56-75, positionwise_feed_forward.py:28-30, encoder_layer.py:76-150
# --------------------------------------------------------------------------------------------
def ffn(x: Tensor, sd: SD, p: str, dp=None, site: str = "") -> Tensor:
    return _lin(_dp(dp, torch.relu(_lin(x, sd, f"{p}.w_1")), f"{site}.hidden"), sd, f"{p}.w_2")


def conv_module(x: Tensor, sd: SD, p: str, training: bool, stats_out: dict | None = None) -> Tensor:
    """pointwise 1x1 (D->2D) -> GLU -> depthwise k (pad (k-1)/2) -> BatchNorm1d -> Swish -> pointwise 1x1.  The padding mask is ignored."""
    h = _lin(x, sd, f"{p}.pointwise_cov1")
    D = x.size(-1)
    h = h[..., :D] * torch.sigmoid(h[..., D:])
    w = sd[f"{p}.depthwise_conv.weight"]
    h = F.conv1d(h.transpose(1, 2), w, sd[f"{p}.depthwise_conv.bias"], padding=(w.size(-1) - 1) // 2, groups=D)
    h = swish(batch_norm(h, sd, f"{p}.norm", training, stats_out))
    return _lin(h.transpose(1, 2), sd, f"{p}.pointwise_cov2")


def encoder_layer(x: Tensor, pos: Tensor, mask: Tensor | None, sd: SD, p: str, heads: int, training: bool,
                  stats_out: dict | None = None, dp=None, s: str = "") -> Tensor:
    x = x + 0.5 * _dp(dp, ffn(_ln(x, sd, f"{p}.norm_ff_macaron"), sd, f"{p}.feed_forward_macaron", dp, f"{s}.ffm"), f"{s}.ffm.out")
    x = x + _dp(dp, rel_mha(_ln(x, sd, f"{p}.norm_mha"), pos, mask, sd, f"{p}.self_attn", heads, dp, f"{s}.attn.probs"), f"{s}.attn.out")
    x = x + _dp(dp, conv_module(_ln(x, sd, f"{p}.norm_conv"), sd, f"{p}.conv_module", training, stats_out), f"{s}.conv.out")
    x = x + 0.5 * _dp(dp, ffn(_ln(x, sd, f"{p}.norm_ff"), sd, f"{p}.feed_forward", dp, f"{s}.ff"), f"{s}.ff.out")
    return _ln(x, sd, f"{p}.norm_final")


def encoder(x: Tensor, mask: Tensor | None, sd: SD, args: Any, training: bool, stats_out: dict | None = None,
            keep: dict | None = None, dp=None) -> Tensor:
    """transformer/encoder.py:257-289 with input_layer conv3d, rel_mha, macaron, cnn module, normalize_before."""
    D = int(args.adim)
    feats = frontend(x, sd, training, stats_out, keep)
    if keep is not None:
        keep["feats"] = feats
    h = _dp(dp, _lin(feats, sd, "encoder.embed.0"), "enc.embed.x") * math.sqrt(D)       # dropout(x * xscale), embedding.py:208,217
    pos = rel_pos_emb(h.size(1), D).to(h.dtype)          # table computed in fp32 as the reference does
    if dp is not None:
        pos = _dp(dp, pos.to(torch.bfloat16).to(h.dtype), "enc.embed.pos")               # the HIP path keeps this table in bf16
    for i in range(int(args.elayers)):
        h = encoder_layer(h, pos, mask, sd, f"encoder.encoders.{i}", int(args.aheads), training, stats_out, dp, f"enc.{i}")
        if keep is not None:
            keep[f"enc{i}"] = h
    return _ln(h, sd, "encoder.after_norm")


# --------------------------------------------------------------------------------------------
# heads and losses: e2e_asr_transformer.py:193-227, ctc.py:65-151, decoder.py:122-151, label_smoothing_loss.py:41-63
# --------------------------------------------------------------------------------------------
def ctc_loss(h: Tensor, hlens: Tensor, ys: list[Tensor], sd: SD, dp=None) -> Tensor:
    """ctc_lo -> log_softmax -> CTC(sum, zero_infinity, blank 0) / B   (ctc.py:65-74).  The reference calls torch's builtin
    ``CTCLoss``; so does this restatement — ``ctc_nll_reference`` below spells the recursion out and is tested against it."""
    lp = _lin(_dp(dp, h, "ctc.in"), sd, "ctc.ctc_lo").transpose(0, 1).log_softmax(2)                # ctc.py:97
    olens = torch.tensor([len(y) for y in ys], dtype=torch.long)
    loss = F.ctc_loss(lp, torch.cat(ys), hlens.long(), olens, blank=0, reduction="sum", zero_infinity=True)
    return loss / h.size(0)


def ctc_nll_reference(lp: Tensor, y: Tensor) -> Tensor:
    """-log p(y | lp) by the textbook alpha recursion in log space; lp [T,V] log-probs, y [L] (blank = 0)."""
    T = lp.size(0)
    ext = torch.zeros(2 * len(y) + 1, dtype=torch.long)
    ext[1::2] = y
    S = len(ext)
    ninf = torch.tensor(float("-inf"), dtype=lp.dtype)
    alpha = torch.full((S,), float("-inf"), dtype=lp.dtype)
    alpha[0] = lp[0, 0]
    if S > 1:
        alpha[1] = lp[0, ext[1]]
    for t in range(1, T):
        new = torch.full((S,), float("-inf"), dtype=lp.dtype)
        for s in range(S):
            c = [alpha[s]]
            if s >= 1:
                c.append(alpha[s - 1])
            if s >= 2 and ext[s] != 0 and ext[s] != ext[s - 2]:
                c.append(alpha[s - 2])
            new[s] = torch.logsumexp(torch.stack(c), 0) + lp[t, ext[s]]
        alpha = new
    tail = torch.stack([alpha[S - 1], alpha[S - 2] if S > 1 else ninf])
    return -torch.logsumexp(tail, 0)


def add_sos_eos(ys: list[Tensor], sos: int, eos: int, ignore_id: int = -1) -> tuple[Tensor, Tensor]:
    """transformer/add_sos_eos.py:10-31: ys_in = [sos, y] padded with eos; ys_out = [y, eos] padded with ignore_id."""
    L = max(len(y) for y in ys) + 1
    ys_in = torch.full((len(ys), L), eos, dtype=torch.long)
    ys_out = torch.full((len(ys), L), ignore_id, dtype=torch.long)
    for b, y in enumerate(ys):
        ys_in[b, 0] = sos
        ys_in[b, 1 : len(y) + 1] = y
        ys_out[b, : len(y)] = y
        ys_out[b, len(y)] = eos
    return ys_in, ys_out


def decoder(ys_in: Tensor, memory: Tensor, memory_mask: Tensor, sd: SD, args: Any, keep: dict | None = None, dp=None) -> Tensor:
    """decoder.py:122-151 + decoder_layer.py:60-121 (pre-LN).  Self-attention mask = causal (ys_in is eos-padded, never -1:
    mask.py:41-51); source mask = encoder padding mask."""
    D = int(args.ddim)
    L = ys_in.size(1)
    x = F.embedding(ys_in, sd["decoder.embed.0.weight"]) * math.sqrt(D) + abs_pos_emb(L, D).to(memory.dtype)       # embedding.py:78-89
    x = _dp(dp, x, "dec.embed")
    causal = torch.tril(torch.ones(L, L, dtype=torch.bool)).unsqueeze(0)
    for i in range(int(args.dlayers)):
        p, s = f"decoder.decoders.{i}", f"dec.{i}"
        t = _ln(x, sd, f"{p}.norm1")
        x = x + _dp(dp, mha(t, t, causal, sd, f"{p}.self_attn", int(args.dheads), dp, f"{s}.self.probs"), f"{s}.self.out")
        x = x + _dp(dp, mha(_ln(x, sd, f"{p}.norm2"), memory, memory_mask, sd, f"{p}.src_attn", int(args.dheads), dp, f"{s}.src.probs"),
                    f"{s}.src.out")
        x = x + _dp(dp, ffn(_ln(x, sd, f"{p}.norm3"), sd, f"{p}.feed_forward", dp, f"{s}.ff"), f"{s}.ff.out")
        if keep is not None:
            keep[f"dec{i}"] = x
    return _lin(_ln(x, sd, "decoder.after_norm"), sd, "decoder.output_layer")


def label_smoothing_loss(pred: Tensor, target: Tensor, smoothing: float, normalize_length: bool = False, ignore_id: int = -1) -> Tensor:
    """KL(true_dist || softmax(pred)) summed over non-padded rows / batch size (label_smoothing_loss.py:41-63)."""
    B, L, V = pred.shape
    x = pred.reshape(-1, V)
    t = target.reshape(-1)
    ignore = t == ignore_id
    true = torch.full_like(x, smoothing / (V - 1))
    true.scatter_(1, t.masked_fill(ignore, 0).unsqueeze(1), 1.0 - smoothing)
    kl = torch.xlogy(true, true) - true * torch.log_softmax(x, dim=1)      # nn.KLDivLoss: 0 log 0 = 0
    denom = int((~ignore).sum()) if normalize_length else B
    return kl.masked_fill(ignore.unsqueeze(1), 0.0).sum() / denom


def th_accuracy(pred: Tensor, target: Tensor, ignore_id: int = -1) -> float:
    """nets_utils.py:303-323."""
    m = target != ignore_id
    return float((pred.argmax(-1)[m] == target[m]).sum()) / float(m.sum())


def forward(sd: SD, args: Any, x: Tensor, lengths: Tensor, audio_tokens: Tensor, label: Tensor, training: bool = True,
            stats_out: dict | None = None, keep: dict | None = None, dp: DropPlan | None = None) -> dict[str, Any]:
    """``E2E.forward`` (e2e_asr_transformer.py:187-227) with pre-computed audio tokens in the ``audios`` slot."""
    from syncvsr_amd.lrs_init import lrs_audio_dims          # parameter-free helper (codec string -> (A,G,V))
    odim = sd["decoder.output_layer.weight"].size(0)
    B, T = x.shape[:2]
    mask = (torch.arange(T).unsqueeze(0) < lengths.view(-1, 1)).unsqueeze(-2)                 # make_non_pad_mask, [B,1,T]
    if not training:
        dp = None
    h = encoder(x, mask, sd, args, training, stats_out, keep, dp)
    if keep is not None:
        keep["enc_out"] = h
    A, G, V = lrs_audio_dims(args)
    logits_audio = _lin(h, sd, "audio_classifier").float().unflatten(2, (-1, V))
    loss_audio = F.cross_entropy(logits_audio.flatten(0, 2), audio_tokens[:, : T * A].flatten())
    ys = [y[y != -1] for y in label.view(B, -1)]
    loss_ctc = ctc_loss(h, lengths, ys, sd, dp) if float(args.mtlalpha) > 0.0 else torch.zeros(())     # e2e_asr_transformer.py:205-208
    ys_in, ys_out = add_sos_eos(ys, odim - 1, odim - 1)
    memory = _lin(h, sd, "proj_decoder") if "proj_decoder.weight" in sd else h               # e2e_asr_transformer.py:209-210
    pred = decoder(ys_in, memory, mask, sd, args, keep, dp)
    if keep is not None:
        keep["pred"] = pred
    loss_att = label_smoothing_loss(pred.float(), ys_out, float(args.lsm_weight), bool(args.transformer_length_normalized_loss))
    alpha = float(args.mtlalpha)
    loss = alpha * loss_ctc + (1 - alpha) * loss_att + float(args.audio_weight) * loss_audio
    return {"loss": loss, "loss_ctc": loss_ctc, "loss_att": loss_att, "loss_audio": loss_audio,
            "acc": th_accuracy(pred, ys_out)}


# --------------------------------------------------------------------------------------------
# inference scorers (transformer/decoder.py:153-220, scorers/ctc.py:87-127, ctc_prefix_score.py:11-165) — CPU checkers for
# syncvsr_amd/lrs_infer.py; pinned by tests/golden/lrs_infer_tiny.npz (the reference's own BatchBeamSearch run)
# --------------------------------------------------------------------------------------------
CTC_LOGZERO = -1.0e10


def ctc_prefix_score(logp: Tensor, r_prev: Tensor, last: Tensor, ids: Tensor | None, out_len: int, blank: int, eos: int) -> tuple[Tensor, Tensor]:
    """logp [T, V]; r_prev [n, T, 2]; last [n]; ids [n, S] or None -> (r_new [n, S, T, 2], psi [n, S]).  Algorithm 2 of Watanabe et al.
    as ctc_prefix_score.py:118-165 evaluates it: r_n[t] = logaddexp(r_n[t-1], phi[t-1]) + x[t], r_b[t] = logaddexp(r_n[t-1], r_b[t-1])
    + x_blank[t], psi = logsumexp(r_n[start-1], phi[t-1] + x[t] for t >= start)."""
    T, V = logp.shape
    n = r_prev.shape[0]
    if ids is None:
        ids = torch.arange(V).unsqueeze(0).expand(n, V)
    S = ids.shape[1]
    x = logp[:, ids]                                   # [T, n, S]
    xb = logp[:, blank]                                # [T]
    r_sum = torch.logaddexp(r_prev[..., 0], r_prev[..., 1])           # [n, T]
    same = ids == last.view(n, 1)                      # [n, S]
    phi = torch.where(same.unsqueeze(0), r_prev[..., 1].t().unsqueeze(2), r_sum.t().unsqueeze(2))     # [T, n, S]
    start = max(out_len, 1)
    r = torch.full((T, 2, n, S), CTC_LOGZERO, dtype=logp.dtype)
    if out_len == 0:
        r[0, 0] = x[0]
    acc = [r[start - 1, 0]]
    for t in range(start, T):
        r[t, 0] = torch.logaddexp(r[t - 1, 0], phi[t - 1]) + x[t]
        r[t, 1] = torch.logaddexp(r[t - 1, 0], r[t - 1, 1]) + xb[t]
        acc.append(phi[t - 1] + x[t])
    psi = torch.logsumexp(torch.stack(acc), dim=0)
    psi = torch.where(ids == eos, r_sum[:, T - 1].unsqueeze(1), psi)
    psi = torch.where(ids == blank, torch.full_like(psi, CTC_LOGZERO), psi)
    return r.permute(2, 3, 0, 1).contiguous(), psi


class OracleDecoderScorer:
    """Decoder.batch_score restated without the cache (the last row of the causal decoder is the same)."""

    def __init__(self, sd: SD, args: Any):
        self.sd, self.args = sd, args

    def batch_init_state(self, x):
        return None

    def batch_score(self, ys: Tensor, states, xs: Tensor):
        n, T = xs.shape[:2]
        pred = decoder(ys, xs, torch.ones(n, 1, T, dtype=torch.bool), self.sd, self.args)
        return torch.log_softmax(pred[:, -1], dim=-1), [None] * n

    def select_states(self, states, prev, tok):
        return None


def make_oracle_ctc_scorer(sd: SD, eos: int):
    """The product's CTCPrefixScorer host logic (full-vocabulary scatter, eos / blank rules, state selection) with the two device
    calls replaced by this module's CPU restatements — so the CPU test exercises the shipped search + state plumbing."""
    from syncvsr_amd.lrs_infer import CTCPrefixScorer

    class _Scorer(CTCPrefixScorer):
        def ctc_log_softmax(self, x: Tensor) -> Tensor:
            return torch.log_softmax(_lin(x, sd, "ctc.ctc_lo"), dim=-1)

        def _prefix(self, logp, r_prev, last, ids, out_len):
            return ctc_prefix_score(logp, r_prev, last, ids, out_len, self.blank, self.eos)

    return _Scorer(None, eos)
