"""Golden vectors for the LRS inference surface (encoder -> decoder / CTC-prefix scorers -> batch beam search).  RUNS ONLY IN
THE BUILD CONTAINER: imports the reference itself (`/root/reference/LRS/video/espnet/...`, import stub of SURVEY.md App. C) and
records what ITS `E2E.encoder`, `Decoder.batch_score`, `CTCPrefixScorer.batch_score_partial` and `BatchBeamSearch` compute for
seeded weights and a seeded clip, exactly as `ModelModule.test_step` / `get_beam_search_decoder` drive them
(LRS/video/lightning.py:114-129,237-279).  Only numbers are stored; weights are regenerated from the seed.

    python tests/golden/make_golden_lrs_infer.py [case ...]
"""
from __future__ import annotations

import os
import sys
from argparse import Namespace

import numpy as np
import torch
import torch.nn as nn
import transformers  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

from golden_cases import LRS_INFER_CASES, build_lrs_infer_case  # noqa: E402
from make_golden_lrs import import_reference  # noqa: E402
from syncvsr_amd.lrs_init import lrs_audio_dims  # noqa: E402


def run_case(E2E, name: str) -> dict[str, np.ndarray]:
    from espnet.nets.batch_beam_search import BatchBeamSearch
    from espnet.nets.scorers.length_bonus import LengthBonus

    args, odim, sd, clip, runs, _ = build_lrs_infer_case(name, load_golden=False)
    ns = Namespace(**{k: v for k, v in args.items() if k != "codec"}, codec=None)
    torch.manual_seed(0)
    model = E2E(odim, ns)
    A, G, V = lrs_audio_dims(args)
    model.audio_classifier = nn.Linear(int(args.adim), A * G * V)
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    model.eval().double()
    res: dict[str, np.ndarray] = {}
    with torch.no_grad():
        enc_feat, _ = model.encoder(clip.double().unsqueeze(0), None)          # lightning.py:115-118
        enc_feat = enc_feat.squeeze(0)
        res["enc_feat"] = enc_feat.float().numpy()
        res["ctc_logp"] = model.ctc.log_softmax(enc_feat.unsqueeze(0)).squeeze(0).float().numpy()
        token_list = [f"t{i}" for i in range(odim)]
        for r, (beam, ctcw) in enumerate(runs):
            scorers = model.scorers()                                          # e2e_asr_transformer.py:182-184
            scorers["lm"] = None
            scorers["length_bonus"] = LengthBonus(len(token_list))
            weights = {"decoder": 1.0 - ctcw, "ctc": ctcw, "lm": 0.0, "length_bonus": 0.0}     # lightning.py:263-268 (penalty 0)
            bs = BatchBeamSearch(beam_size=beam, vocab_size=len(token_list), weights=weights, scorers=scorers, sos=odim - 1, eos=odim - 1,
                                 token_list=token_list, pre_beam_score_key=None if ctcw == 1.0 else "decoder")
            # record the first calls of both scorers (inputs and outputs) while the reference's own search runs
            calls: dict[str, list] = {"dec": [], "ctc": []}
            dec, ctc = scorers["decoder"], scorers["ctc"]
            orig_dec, orig_ctc = dec.batch_score, ctc.batch_score_partial

            def rec_dec(ys, states, xs, _o=orig_dec):
                out = _o(ys, states, xs)
                if len(calls["dec"]) < 4:
                    calls["dec"].append((ys.clone(), out[0].clone()))
                return out

            def rec_ctc(y, ids, state, x, _o=orig_ctc):
                out = _o(y, ids, state, x)
                if len(calls["ctc"]) < 4:
                    calls["ctc"].append((y.clone(), None if ids is None else ids.clone(), out[0].clone()))
                return out

            dec.batch_score, ctc.batch_score_partial = rec_dec, rec_ctc
            nbest = bs(enc_feat)
            dec.batch_score, ctc.batch_score_partial = orig_dec, orig_ctc
            n = min(len(nbest), 10)
            L = max(len(h.yseq) for h in nbest[:n])
            ys = np.full((n, L), -1, dtype=np.int64)
            for i, h in enumerate(nbest[:n]):
                ys[i, : len(h.yseq)] = h.yseq.numpy()
            res[f"run{r}.beam"] = np.int64(beam)
            res[f"run{r}.ctc_weight"] = np.float64(ctcw)
            res[f"run{r}.n_ended"] = np.int64(len(nbest))
            res[f"run{r}.yseq"] = ys
            res[f"run{r}.score"] = np.array([float(h.score) for h in nbest[:n]])
            res[f"run{r}.score_decoder"] = np.array([float(h.scores["decoder"]) for h in nbest[:n]])
            res[f"run{r}.score_ctc"] = np.array([float(h.scores["ctc"]) for h in nbest[:n]])
            for j, (ys_in, logp) in enumerate(calls["dec"]):
                res[f"run{r}.dec{j}.ys"] = ys_in.numpy()
                res[f"run{r}.dec{j}.logp"] = logp.float().numpy()
            for j, (y, ids, sc) in enumerate(calls["ctc"]):
                res[f"run{r}.ctc{j}.y"] = y.numpy()
                if ids is not None:
                    res[f"run{r}.ctc{j}.ids"] = ids.numpy()
                res[f"run{r}.ctc{j}.score"] = sc.double().numpy()
            print(f"{name} run{r} beam={beam} ctc={ctcw}: ended={len(nbest)} best={nbest[0].yseq.tolist()} score={float(nbest[0].score):.4f} "
                  f"2nd={float(nbest[1].score) if len(nbest) > 1 else float('nan'):.4f}")
    if odim > 1000:           # full-size vocabulary: keep the fixture small — two rows of the first two calls, CTC posteriors as arg-max + max
        lp = res.pop("ctc_logp")
        res["ctc_logp_argmax"], res["ctc_logp_max"] = lp.argmax(-1), lp.max(-1)
        for k in list(res):
            part = k.split(".")[1] if "." in k else ""
            if part[:3] in ("dec", "ctc") and part[3:].isdigit():
                j = int(part[3:])
                if j >= 2:
                    del res[k]
                elif k.endswith((".logp", ".score")):
                    res[k] = res[k][:2]
    return res


def main() -> None:
    E2E = import_reference()
    torch.set_num_threads(8)
    for name in (sys.argv[1:] or list(LRS_INFER_CASES)):
        res = run_case(E2E, name)
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **res)
        print(f"-> {path} ({os.path.getsize(path) / 1024:.0f} KB)")


if __name__ == "__main__":
    main()
