"""Inference surface of the LRS (sentence-level) model: the scorers and the batch beam search the reference's test loop drives.

    enc_feat, _ = model.encoder(sample.unsqueeze(0), None)          # LRS/video/lightning.py:114-118
    nbest = get_beam_search_decoder(model, token_list)(enc_feat.squeeze(0))        # lightning.py:119,237-279

Mirrors, by name and argument meaning, `E2E.scorers()` (e2e_asr_transformer.py:182-184), `Decoder.forward_one_step / score /
batch_score` (transformer/decoder.py:153-220), `CTCPrefixScorer` (scorers/ctc.py), `LengthBonus` (scorers/length_bonus.py),
`BatchBeamSearch` (batch_beam_search.py, beam_search.py:285-420), `end_detect` (e2e_asr_common.py:19-49) and
`get_beam_search_decoder` (lightning.py:237-279).  Design differences, all result-preserving:

  * the search state is batched tensors from start to end (token matrix, score vector, CTC forward variables [n, T, 2]) — the
    reference converts to and from per-hypothesis Python objects every step (batch_beam_search.py:230-275);
  * the CTC prefix recursion over the T frames runs in ONE HIP kernel per step, one thread per (hypothesis, candidate) pair
    (csrc/lrs_misc.hip k_ctc_prefix_score) instead of ~10 small torch kernels per frame (ctc_prefix_score.py:139-146);
  * the decoder scorer recomputes the prefix with the training kernels (one [n * L]-row batch keeps the MFMA tiles full) instead
    of caching per-layer outputs: the decoder's self-attention is causal, so the last row is identical either way.

The search itself is device-agnostic host logic over torch tensors; the two neural scorers need the HIP library.
"""
from __future__ import annotations

import math
from typing import Any, NamedTuple, Optional

import torch

from . import ops

LOGZERO = -1.0e10          # ctc_prefix_score.py:33
BF16 = torch.bfloat16


class Hypothesis(NamedTuple):
    """beam_search.py:17-33."""

    yseq: torch.Tensor
    score: float = 0.0
    scores: dict = {}
    states: dict = {}

    def asdict(self) -> dict:
        return dict(yseq=self.yseq.tolist(), score=float(self.score), scores={k: float(v) for k, v in self.scores.items()})


def end_detect(ended_hyps: list, i: int, M: int = 3, D_end: float = math.log(1 * math.exp(-10))) -> bool:
    """Eq. (50) of Watanabe et al. (e2e_asr_common.py:19-49): stop when, for each of the last M lengths, the best hypothesis that
    ended with that length scores more than |D_end| below the best ended hypothesis."""
    if not ended_hyps:
        return False
    best = max(float(h["score"]) for h in ended_hyps)
    count = 0
    for m in range(M):
        same = [float(h["score"]) for h in ended_hyps if len(h["yseq"]) == i - m]
        if same and max(same) - best < D_end:
            count += 1
    return count == M


# ----------------------------------------------------------------------------------------------------
# scorers
# ----------------------------------------------------------------------------------------------------
class LengthBonus:
    """scorers/length_bonus.py: +1 per emitted token (weighted by `penalty`)."""

    def __init__(self, n_vocab: int):
        self.n = int(n_vocab)

    def batch_init_state(self, x):
        return None

    def batch_score(self, ys, states, xs):
        return torch.ones((ys.shape[0], self.n), dtype=xs.dtype, device=xs.device), None

    def select_states(self, states, prev, tok):
        return None


class DecoderScorer:
    """The attention decoder as a full-vocabulary scorer (transformer/decoder.py:153-220), incremental like the reference's
    `forward_one_step(..., cache)` (decoder.py:153-186 + decoder_layer.py:67-103): a step computes ONE new row per hypothesis.

    State / cache: one tensor per decoder layer, [n, L, 3*ddim] = (layer output | self-attention key | self-attention value) of the
    L positions scored so far — a list of per-layer [n, L, .] tensors exactly like the reference's cache, so generic scorer plumbing
    (stack / index by hypothesis) works on it unchanged.  The reference caches the layer outputs only and re-projects keys and
    values of the whole prefix every step; keeping them too makes a step O(L) in the attention alone.  The source-attention keys /
    values of `memory` (recomputed per step in the reference, decoder_layer.py:106-113) are projected once per clip and layer."""

    def __init__(self, model):
        self.model = model
        self._mem = None          # (memory tensor, version, per-layer [T, 2D] of ONE row when all rows alias it | None, {n: [per-layer [n*T, 2D]]})

    # -- ScorerInterface / BatchScorerInterface ---------------------------------------------------
    def init_state(self, x):
        return None

    def batch_init_state(self, x):
        return None

    def select_state(self, state, i, new_id=None):
        return None if state is None else [c[i] for c in state]

    def select_states(self, states, prev, tok):
        return None if states is None else tuple(c[prev] for c in states)

    # -- source-attention keys / values -----------------------------------------------------------
    def _memory_kv(self, st, memory: torch.Tensor, n: int, T: int):
        m = self.model
        D = m.ddim
        c = self._mem
        try:                       # inference tensors (torch.inference_mode) carry no version counter: identity + weights generation decide alone
            ver = memory._version
        except RuntimeError:
            ver = -1
        gen = getattr(st, "generation", 0)          # bumped by every optimiser step / shadow refresh: cached keys / values follow the weights
        same = (c is not None and c["ptr"] == memory.data_ptr() and c["ver"] == ver and c["gen"] == gen and c["T"] == T and c["stride"] == memory.stride()
                and c["base"] is memory._base)
        if not same:
            c = self._mem = dict(ptr=memory.data_ptr(), ver=ver, gen=gen, T=T, stride=memory.stride(), base=memory._base, keep=memory, row=None, by_n={})
        if n in c["by_n"]:
            return c["by_n"][n]
        aliased = n > 1 and memory.stride(0) == 0           # x.unsqueeze(0).expand(n, T, D): every hypothesis attends to the same clip
        if aliased or n == 1:
            if c["row"] is None:
                mem = memory[0].to(BF16).contiguous()
                c["row"] = [_lin_kv(st, mem, f"decoder.decoders.{i}.src_attn.linear_k", T, D) for i in range(m.dlayers)]
            kv = [r.unsqueeze(0).expand(n, T, 2 * D).reshape(n * T, 2 * D).contiguous() if n > 1 else r for r in c["row"]]
        else:
            mem = memory.to(BF16).reshape(n * T, D).contiguous()
            kv = [_lin_kv(st, mem, f"decoder.decoders.{i}.src_attn.linear_k", n * T, D) for i in range(m.dlayers)]
        c["by_n"] = {n: kv}                                  # (the beam only shrinks or stays: one width at a time is enough)
        return kv

    def forward_one_step(self, tgt: torch.Tensor, tgt_mask, memory: torch.Tensor, memory_mask=None, cache=None):
        """tgt int64 [n, L], memory [n, T, ddim], cache: None or per-layer [n, L-1, 3*ddim] from the previous step ->
        (log-probabilities of the next token [n, odim], new cache: per-layer [n, L, 3*ddim]).  `tgt_mask` is the causal mask by
        construction (decoder.py:189,216).  Without a cache the whole prefix is computed (and the cache built); with one, only
        position L-1."""
        from .lrs_model import LrsTargets, _decoder_fwd

        m = self.model
        if m.training:
            raise RuntimeError("forward_one_step is an inference entry point: call model.eval() first")
        if memory.size(-1) != m.ddim:
            raise ValueError(f"memory is {memory.size(-1)} wide, the decoder expects ddim = {m.ddim} (the reference feeds the encoder "
                             "output to the decoder directly at inference, lightning.py:114-119, which needs adim == ddim)")
        st = m.store()
        m._side.join()                # (a TrainStep may have left the tail of its optimiser step on the side stream)
        if not st.shadow_fresh:
            st.refresh_shadows()
            self._mem = None                         # projected with the old weights
        n, L = tgt.shape
        T, D = memory.size(1), m.ddim
        if memory_mask is not None:
            ilen = memory_mask.reshape(n, -1).sum(-1).to(torch.int32).contiguous()
        else:
            ilen = torch.full((n,), T, dtype=torch.int32, device=memory.device)
        if cache is not None and (len(cache) != m.dlayers or any(c is None for c in cache)):
            cache = None
        if cache is not None and (cache[0].shape[0] != n or cache[0].shape[1] != L - 1 or cache[0].shape[2] != 3 * D):
            raise ValueError(f"cache entries are {tuple(cache[0].shape)}, expected ({n}, {L - 1}, {3 * D}): the cache must come from the "
                             "previous forward_one_step call for the same hypotheses")
        with torch.no_grad():
            if cache is None or L == 1:
                tg = LrsTargets(None, tgt.contiguous(), None)
                mem = memory.to(BF16).reshape(n * T, D).contiguous()
                tape: dict[str, Any] = {}
                pred = _decoder_fwd(m, st, tape, tg, mem, ilen, n, T)                # fp32 [n * L, odim padded to 64]
                logits = pred.view(n, L, -1)[:, -1, : m.odim]
                new_cache = tuple(torch.cat((tape[f"decoder.decoders.{i}"]["out"].view(n, L, D),
                                             tape[f"decoder.decoders.{i}"]["self"]["qkv"].view(n, L, 3 * D)[:, :, D:]), dim=2) for i in range(m.dlayers))
            else:
                logits, new_cache = self._step_cached(st, tgt, memory, ilen, cache, n, L, T)
        return torch.log_softmax(logits.float(), dim=-1), new_cache

    def _step_cached(self, st, tgt, memory, ilen, cache, n: int, L: int, T: int):
        """Position L-1 of every hypothesis on top of `cache` (decoder_layer.py:67-127 with tgt_q = tgt[:, -1:])."""
        from .lrs_model import _ffn_fwd, _lin, _ln

        m = self.model
        D, U, H = m.ddim, m.dunits, m.dheads
        memkv = self._memory_kv(st, memory, n, T)
        pe = m._pos_table("abs", L, memory.device)
        x = ops.embed_pos_fwd(tgt[:, -1:].contiguous(), st.p32("decoder.embed.0.weight"), pe[L - 1 : L].contiguous(), 1, D, math.sqrt(D))   # [n, D]
        new_cache = []
        for i in range(m.dlayers):
            p = f"decoder.decoders.{i}"
            c = cache[i]
            t1, _, _ = _ln(st, x, f"{p}.norm1")
            qkv = _lin(st, t1, f"{p}.self_attn.linear_q", n, D, 3 * D)                                     # this position's q | k | v
            kv = torch.cat((c[:, :, D:], qkv[:, D:].unsqueeze(1)), dim=1).view(n * L, 2 * D)               # keys / values 0..L-1
            ctx, _ = ops.mha_fwd(qkv, 3 * D, kv, kv[:, D:], 2 * D, B=n, H=H, Lq=1, Lk=L)                     # the last query sees every key
            x1 = _lin(st, ctx, f"{p}.self_attn.linear_out", n, D, D, addend=x)
            t2, _, _ = _ln(st, x1, f"{p}.norm2")
            q = _lin(st, t2, f"{p}.src_attn.linear_q", n, D, D)
            ctx2, _ = ops.mha_fwd(q, D, memkv[i], memkv[i][:, D:], 2 * D, B=n, H=H, Lq=1, Lk=T, klen=ilen)
            x2 = _lin(st, ctx2, f"{p}.src_attn.linear_out", n, D, D, addend=x1)
            x = _ffn_fwd(m, st, {}, "ff", x2, f"{p}.feed_forward", n, D, U, 1.0, f"{p}.norm3", f"dec.{i}.ff")
            new_cache.append(torch.cat((c, torch.cat((x, qkv[:, D:]), dim=1).unsqueeze(1)), dim=1))
        tn, _, _ = _ln(st, x, "decoder.after_norm")
        V = m.odim
        Vp = (V + 63) // 64 * 64
        pred = ops.linear_fwd(tn, st.s16("decoder.output_layer.weight"), st.p32("decoder.output_layer.bias"), rows=n, K=D, N=V, x_pitch=D,
                              out_f32=True, out_pitch=Vp)[0]
        return pred[:, :V], tuple(new_cache)

    def score(self, ys: torch.Tensor, state, x: torch.Tensor):
        logp, cache = self.forward_one_step(ys.unsqueeze(0), None, x.unsqueeze(0), cache=None if state is None else [c.unsqueeze(0) for c in state])
        return logp.squeeze(0), [c.squeeze(0) for c in cache]

    def batch_score(self, ys: torch.Tensor, states, xs: torch.Tensor):
        """states: None (first step) or the batched cache (per-layer [n, L-1, 3*ddim], as select_states returns it); a list of
        per-hypothesis states (the reference's calling convention, batch_beam_search.py) is stacked first."""
        if isinstance(states, list) and states and isinstance(states[0], (list, tuple)):
            states = [torch.stack([s[i] for s in states]) for i in range(len(states[0]))]
        elif isinstance(states, list):                      # [None] * n
            states = None
        return self.forward_one_step(ys, None, xs, cache=states)


def _lin_kv(st, mem, name: str, rows: int, D: int):
    from .lrs_model import _lin

    return _lin(st, mem, name, rows, D, 2 * D)


class CTCPrefixScorer:
    """CTC prefix scores of the candidate extensions (scorers/ctc.py:87-127 + ctc_prefix_score.py:11-165).  State of the running
    hypotheses: (r [n, T, 2] forward log-probabilities ending in non-blank / blank, s [n] log prefix probability)."""

    blank = 0

    def __init__(self, model, eos: int):
        self.model, self.eos = model, int(eos)
        self.logp: Optional[torch.Tensor] = None

    def ctc_log_softmax(self, x: torch.Tensor) -> torch.Tensor:
        """`CTC.log_softmax` (ctc.py:163-170): x [T, adim] -> fp32 [T, odim]."""
        m = self.model
        st = m.store()
        m._side.join()                # (a TrainStep may have left the tail of its optimiser step on the side stream)
        if not st.shadow_fresh:
            st.refresh_shadows()
        T = x.size(0)
        with torch.no_grad():
            logits = ops.linear_fwd(x.to(BF16).contiguous(), st.s16("ctc.ctc_lo.weight"), st.p32("ctc.ctc_lo.bias"), rows=T, K=m.adim, N=m.odim,
                                    x_pitch=m.adim, out_f32=True, out_pitch=(m.odim + 63) // 64 * 64)[0]
        return torch.log_softmax(logits[:, : m.odim].float(), dim=-1).contiguous()

    def batch_init_state(self, x: torch.Tensor):
        self.logp = self.ctc_log_softmax(x)
        return None

    def _prefix(self, logp, r_prev, last, ids, out_len):
        return ops.ctc_prefix_score(logp, r_prev, last, ids, out_len, self.blank, self.eos)

    def batch_score_partial(self, y: torch.Tensor, ids: Optional[torch.Tensor], state, x: torch.Tensor):
        """y int64 [n, L] (with <sos>), ids int64 [n, S] or None -> (scores [n, odim]: log psi(prefix + c) - log psi(prefix), with
        -1e10 for labels outside ids and for blank, pending state for select_states)."""
        logp = self.logp
        T, V = logp.shape
        n = y.shape[0]
        if state is None:
            r_prev = torch.full((T, 2), LOGZERO, dtype=logp.dtype, device=logp.device)
            r_prev[:, 1] = torch.cumsum(logp[:, self.blank], 0)
            r_prev = r_prev.unsqueeze(0).expand(n, T, 2).contiguous()
            s_prev = torch.zeros(n, dtype=logp.dtype, device=logp.device)
        else:
            r_prev, s_prev = state
        ids_c = None if ids is None else ids.contiguous()
        r_new, psi = self._prefix(logp, r_prev.contiguous(), y[:, -1].contiguous(), ids_c, y.shape[1] - 1)
        if ids_c is None:
            full = psi.clone()
        else:
            full = torch.full((n, V), LOGZERO, dtype=logp.dtype, device=logp.device).scatter_(1, ids_c, psi)
        full[:, self.eos] = torch.logaddexp(r_prev[:, T - 1, 0], r_prev[:, T - 1, 1])
        full[:, self.blank] = LOGZERO
        return full - s_prev.unsqueeze(1), (r_new, full, ids_c)

    def select_states(self, pending, prev: torch.Tensor, tok: torch.Tensor):
        """State of the extensions (prev[i], tok[i])."""
        r_new, full, ids = pending
        if ids is None:
            j = tok
        else:
            idmap = torch.full(full.shape, 0, dtype=torch.int64, device=full.device).scatter_(
                1, ids, torch.arange(ids.shape[1], device=full.device).expand_as(ids))
            j = idmap[prev, tok]
        return r_new[prev, j].contiguous(), full[prev, tok].contiguous()


# ----------------------------------------------------------------------------------------------------
# batch beam search
# ----------------------------------------------------------------------------------------------------
class BatchBeamSearch:
    """beam_search.py:36-113 (constructor contract) + batch_beam_search.py (one vectorised step per output position)."""

    def __init__(self, beam_size: int, vocab_size: int, weights: dict, scorers: dict, sos: int, eos: int, token_list=None,
                 pre_beam_ratio: float = 1.5, pre_beam_score_key: Optional[str] = None):
        self.weights = weights
        self.scorers, self.full_scorers, self.part_scorers = {}, {}, {}
        for k, v in scorers.items():
            if weights.get(k, 0) == 0 or v is None:          # beam_search.py:73-76
                continue
            self.scorers[k] = v
            (self.part_scorers if hasattr(v, "batch_score_partial") else self.full_scorers)[k] = v
        self.sos, self.eos, self.token_list = int(sos), int(eos), token_list
        self.beam_size, self.n_vocab = int(beam_size), int(vocab_size)
        self.pre_beam_size = int(pre_beam_ratio * beam_size)
        if pre_beam_score_key is not None and pre_beam_score_key != "full" and pre_beam_score_key not in self.full_scorers:
            raise KeyError(f"{pre_beam_score_key} is not found in {self.full_scorers}")
        self.pre_beam_score_key = pre_beam_score_key
        self.do_pre_beam = pre_beam_score_key is not None and self.pre_beam_size < self.n_vocab and len(self.part_scorers) > 0

    # one step: running = dict(yseq [n, L], score [n], scores {k: [n]}, states {k: batched state})
    def _search(self, run: dict, x: torch.Tensor) -> dict:
        yseq = run["yseq"]
        n, V = yseq.shape[0], self.n_vocab
        xs = x.unsqueeze(0).expand(n, *x.shape)
        weighted = torch.zeros((n, V), dtype=x.dtype, device=x.device)
        sc, st = {}, {}
        for k, d in self.full_scorers.items():
            sc[k], st[k] = d.batch_score(yseq, run["states"][k], xs)
            weighted += self.weights[k] * sc[k].to(x.dtype)
        part_ids = None
        if self.do_pre_beam:
            pre = weighted if self.pre_beam_score_key == "full" else sc[self.pre_beam_score_key]
            part_ids = torch.topk(pre, self.pre_beam_size, dim=-1)[1]
        for k, d in self.part_scorers.items():
            sc[k], st[k] = d.batch_score_partial(yseq, part_ids, run["states"][k], x)
            weighted += self.weights[k] * sc[k].to(x.dtype)
        weighted += run["score"].to(x.dtype).unsqueeze(1)
        top = weighted.view(-1).topk(min(self.beam_size, n * V))[1]
        prev, tok = torch.div(top, V, rounding_mode="trunc"), top % V
        return dict(
            yseq=torch.cat((yseq[prev], tok.unsqueeze(1)), dim=1),
            score=weighted[prev, tok],
            scores={k: run["scores"][k][prev] + sc[k][prev, tok].to(x.dtype) for k in self.scorers},
            states={k: self.scorers[k].select_states(st[k], prev, tok) for k in self.scorers},
        )

    @staticmethod
    def _take(states, keep: torch.Tensor):
        if states is None:
            return None
        if isinstance(states, tuple):
            return tuple(s[keep] for s in states)
        return states[keep]

    def forward(self, x: torch.Tensor, maxlenratio: float = 0.0, minlenratio: float = 0.0) -> list:
        """x: encoder output of ONE clip [T, D] -> ended hypotheses, best first (beam_search.py:333-405)."""
        if maxlenratio == 0:
            maxlen = x.shape[0]
        elif maxlenratio < 0:
            maxlen = -1 * int(maxlenratio)
        else:
            maxlen = max(1, int(maxlenratio * x.size(0)))
        run = dict(yseq=torch.tensor([[self.sos]], dtype=torch.int64, device=x.device), score=torch.zeros(1, dtype=x.dtype, device=x.device),
                   scores={k: torch.zeros(1, dtype=x.dtype, device=x.device) for k in self.scorers},
                   states={k: d.batch_init_state(x) for k, d in self.scorers.items()})
        ended: list[Hypothesis] = []
        for i in range(maxlen):
            run = self._search(run, x)
            n = run["yseq"].shape[0]
            if i == maxlen - 1:          # batch_beam_search.py:318-334: close every running hypothesis at the length limit
                run["yseq"] = torch.cat((run["yseq"], torch.full((n, 1), self.eos, dtype=torch.int64, device=x.device)), dim=1)
            is_eos = run["yseq"][:, -1] == self.eos
            scores_cpu = run["score"].tolist()
            for b in torch.nonzero(is_eos).view(-1).tolist():
                ended.append(Hypothesis(yseq=run["yseq"][b], score=scores_cpu[b], scores={k: float(v[b]) for k, v in run["scores"].items()}))
            keep = torch.nonzero(~is_eos).view(-1)
            run = dict(yseq=run["yseq"][keep], score=run["score"][keep], scores={k: v[keep] for k, v in run["scores"].items()},
                       states={k: self._take(v, keep) for k, v in run["states"].items()})
            if maxlenratio == 0.0 and end_detect([dict(score=h.score, yseq=h.yseq) for h in ended], i):
                break
            if keep.numel() == 0:
                break
        nbest = sorted(ended, key=lambda h: h.score, reverse=True)
        if not nbest:                      # beam_search.py:383-392
            return [] if minlenratio < 0.1 else self.forward(x, maxlenratio, max(0.0, minlenratio - 0.1))
        return nbest

    __call__ = forward


def get_beam_search_decoder(model, token_list, rnnlm=None, rnnlm_conf=None, penalty=0, ctc_weight: float = 0.1, lm_weight: float = 0.0,
                            beam_size: int = 40, scorers: Optional[dict] = None) -> BatchBeamSearch:
    """LRS/video/lightning.py:237-279.  Language-model rescoring (`rnnlm`) is not part of this package: the reference's default
    passes none (lm_weight 0.0)."""
    if rnnlm:
        raise NotImplementedError("language-model scorers are outside this package (the reference's test loop passes rnnlm=None)")
    sos = eos = model.odim - 1
    scorers = dict(scorers) if scorers is not None else model.scorers()
    scorers["lm"] = None
    scorers["length_bonus"] = LengthBonus(len(token_list))
    weights = {"decoder": 1.0 - ctc_weight, "ctc": ctc_weight, "lm": lm_weight, "length_bonus": penalty}
    return BatchBeamSearch(beam_size=beam_size, vocab_size=len(token_list), weights=weights, scorers=scorers, sos=sos, eos=eos,
                           token_list=token_list, pre_beam_score_key=None if ctc_weight == 1.0 else "decoder")
