"""LRS inference surface on CPU: the shipped beam search (syncvsr_amd/lrs_infer.py BatchBeamSearch, CTCPrefixScorer host logic,
end detection) driven by the oracle's CPU scorers must reproduce what the reference's own BatchBeamSearch produced for the same
seeded weights and clip (tests/golden/lrs_infer_tiny.npz, made by tests/golden/make_golden_lrs_infer.py) — call by call for the
first steps of both scorers, and the final n-best lists with their scores."""
import numpy as np
import torch

from golden_cases import build_lrs_infer_case


def _run(sd, args, odim, enc_feat, beam, ctcw, record):
    from oracle import lrs_oracle as O
    from syncvsr_amd.lrs_infer import get_beam_search_decoder

    dec, ctc = O.OracleDecoderScorer(sd, args), O.make_oracle_ctc_scorer(sd, odim - 1)
    o_dec, o_ctc = dec.batch_score, ctc.batch_score_partial

    def rec_dec(ys, states, xs):
        out = o_dec(ys, states, xs)
        record["dec"].append((ys.clone(), out[0].clone()))
        return out

    def rec_ctc(y, ids, state, x):
        out = o_ctc(y, ids, state, x)
        record["ctc"].append((y.clone(), None if ids is None else ids.clone(), out[0].clone()))
        return out

    dec.batch_score, ctc.batch_score_partial = rec_dec, rec_ctc

    class _M:
        pass

    m = _M()
    m.odim = odim
    bs = get_beam_search_decoder(m, [f"t{i}" for i in range(odim)], ctc_weight=ctcw, beam_size=beam, scorers=dict(decoder=dec, ctc=ctc))
    return bs(enc_feat)


def test_beam_search_reproduces_the_reference_nbest():
    args, odim, sd, clip, runs, gold = build_lrs_infer_case("lrs_infer_tiny")
    sd = {k: v.double() if v.is_floating_point() else v for k, v in sd.items()}
    enc = torch.from_numpy(gold["enc_feat"]).double()
    for r, (beam, ctcw) in enumerate(runs):
        assert int(gold[f"run{r}.beam"]) == beam
        rec = {"dec": [], "ctc": []}
        nbest = _run(sd, args, odim, enc, beam, ctcw, rec)
        assert len(nbest) == int(gold[f"run{r}.n_ended"])
        # the scorers were called with the same prefixes / candidates and answered the same, step by step
        for j in range(4):
            if f"run{r}.dec{j}.ys" not in gold:
                break
            ys, logp = rec["dec"][j]
            assert np.array_equal(ys.numpy(), gold[f"run{r}.dec{j}.ys"]), (r, j)
            np.testing.assert_allclose(logp.numpy(), gold[f"run{r}.dec{j}.logp"], atol=2e-5, rtol=1e-5)
            y, ids, sc = rec["ctc"][j]
            assert np.array_equal(y.numpy(), gold[f"run{r}.ctc{j}.y"]), (r, j)
            if f"run{r}.ctc{j}.ids" in gold:
                assert np.array_equal(ids.numpy(), gold[f"run{r}.ctc{j}.ids"]), (r, j)
            g = gold[f"run{r}.ctc{j}.score"]
            live = g > -1e9                                        # labels outside the pre-beam carry -1e10 (minus the prefix score)
            assert np.array_equal(live, sc.numpy() > -1e9)
            np.testing.assert_allclose(sc.numpy()[live], g[live], atol=2e-5, rtol=1e-5)
        gy = gold[f"run{r}.yseq"]
        for i in range(gy.shape[0]):
            want = gy[i][gy[i] >= 0]
            assert nbest[i].yseq.tolist() == want.tolist(), (r, i, nbest[i].yseq.tolist(), want.tolist())
            assert abs(nbest[i].score - gold[f"run{r}.score"][i]) < 1e-4
            assert abs(nbest[i].scores["decoder"] - gold[f"run{r}.score_decoder"][i]) < 1e-4
            assert abs(nbest[i].scores["ctc"] - gold[f"run{r}.score_ctc"][i]) < 1e-4
        assert nbest[0].asdict()["yseq"][0] == odim - 1 and nbest[0].asdict()["yseq"][-1] == odim - 1


def test_ctc_prefix_restatement_against_brute_force():
    """psi(prefix + c) must equal the summed probability of every CTC path whose collapsed labelling starts with prefix + c."""
    import itertools

    from oracle import lrs_oracle as O

    T, V = 5, 4
    logp = torch.log_softmax(torch.randn(T, V, generator=torch.Generator().manual_seed(3)).double(), dim=-1)
    p = logp.exp()

    def collapse(path):
        out, prev = [], None
        for s in path:
            if s != prev and s != 0:
                out.append(s)
            prev = s
        return out

    def prefix_prob(prefix):          # P(labelling starts with prefix): sum over paths of the first time the prefix is completed
        tot = 0.0
        for t_end in range(T):
            for path in itertools.product(range(V), repeat=t_end + 1):
                if collapse(path) == prefix and (t_end == 0 or collapse(path[:-1]) != prefix):
                    tot += float(torch.prod(torch.stack([p[t, s] for t, s in enumerate(path)])))
        return tot

    # step 0: empty prefix -> single labels; then extend the prefix [2]
    r0 = torch.full((1, T, 2), O.CTC_LOGZERO, dtype=torch.float64)
    r0[0, :, 1] = torch.cumsum(logp[:, 0], 0)
    r1, psi1 = O.ctc_prefix_score(logp, r0, torch.tensor([V - 1]), None, 0, 0, V - 1)
    for c in (1, 2):
        assert abs(float(psi1[0, c].exp()) - prefix_prob([c])) < 1e-9
    r2, psi2 = O.ctc_prefix_score(logp, r1[:, 2].contiguous(), torch.tensor([2]), None, 1, 0, V - 1)
    for c in (1, 2):
        assert abs(float(psi2[0, c].exp()) - prefix_prob([2, c])) < 1e-9
