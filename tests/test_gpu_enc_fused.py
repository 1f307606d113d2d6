"""Fused encoder forward (csrc/enc_fused.hip, svsr_enc_fwd) against the per-layer launch chain it replaces (-m gpu).

The fused kernel keeps the chain's arithmetic to the last bit: the same bf16 rounding points, LayerNorm code and counter-based dropout
masks, and the same ORDER of fp32 additions inside every contraction (one accumulator walking k upwards; two K halves for the 2048-deep
output GEMM; the attention kernel's partial-sum order for the softmax denominators and its two 16-key slices for P.V).  So every
tensor a layer keeps for the backward must be EQUAL to the chain's, bit for bit, dropout on or off.
Reference: HF BertLayer as reached from LRW/video/src/lightning.py:92,152-156.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _tapes(dev, B, T, layers, training, seed=3):
    from syncvsr_amd import model as M
    from syncvsr_amd import ops
    from syncvsr_amd.config import default_lrw_config
    from syncvsr_amd.init import init_state_dict

    cfg = default_lrw_config(model__bert__num_hidden_layers=layers)
    sd = init_state_dict(cfg, seed=11, perturb_norm=True)
    model = M.Model(cfg, seed=seed)
    model.load_state_dict(sd)
    model.to(dev).train(training)
    st = model.store()
    st.refresh_shadows()
    feats = (torch.randn(B * T, 512, generator=torch.Generator().manual_seed(17 + B + T)) * 0.7).to(torch.bfloat16).to(dev)
    out = {}
    for fused in (False, True):
        ops.ENC_FUSED = fused
        try:
            model.reseed_dropout(seed)
            if training:
                model._advance_dropout(dev)
            tape = {}
            h = M._encoder_forward(model, st, tape, feats, B, T)
            torch.cuda.synchronize()
            out[fused] = (h, tape)
        finally:
            ops.ENC_FUSED = True
    return model, out


def _rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm().clamp_min(1e-20))


@pytest.mark.parametrize("B,T,layers,training", [(2, 7, 2, False), (3, 29, 2, True), (32, 29, 6, True), (5, 31, 1, True), (40, 11, 1, False)])
def test_fused_encoder_forward_equals_launch_chain(dev, B, T, layers, training):
    from syncvsr_amd import ops

    model, out = _tapes(dev, B, T, layers, training)
    (h0, t0), (h1, t1) = out[False], out[True]
    S = T + 1
    ws = ops._ENC_WS[next(iter(ops._ENC_WS))]
    assert int(ws.view(torch.int32)[B].item()) == 0, "a bounded cluster wait gave up"
    assert int(ws.view(torch.int32)[:B].min().item()) == 4 * 8 * layers and int(ws.view(torch.int32)[:B].max().item()) == 4 * 8 * layers
    diff = {}
    for i in range(layers):
        a, b = t0[f"encoder.encoder.layer.{i}"], t1[f"encoder.encoder.layer.{i}"]
        for k in ("qkv", "ctx", "ao", "x1", "z", "hg", "f", "m1", "r1", "m2", "r2"):
            if not torch.equal(a[k], b[k]):
                diff[f"{i}.{k}"] = (float((a[k].float() - b[k].float()).abs().max()), float((a[k] != b[k]).float().mean()))
        if not torch.equal(a["probs"][..., :S], b["probs"][..., :S]):
            diff[f"{i}.probs"] = float((a["probs"][..., :S].float() - b["probs"][..., :S].float()).abs().max())
    if not torch.equal(h0, h1):
        diff["h"] = float((h0.float() - h1.float()).abs().max())
    assert not diff, diff


def test_fused_encoder_is_bit_reproducible(dev):
    _, o1 = _tapes(dev, 32, 29, 6, True)
    _, o2 = _tapes(dev, 32, 29, 6, True)
    assert torch.equal(o1[True][0], o2[True][0])
    for k in ("qkv", "probs", "ctx", "ao", "x1", "z", "hg", "f"):
        assert torch.equal(o1[True][1]["encoder.encoder.layer.5"][k], o2[True][1]["encoder.encoder.layer.5"][k]), k


def _backward_pair(dev, B, T, layers, training, seed=3):
    """The encoder's backward from the same tape and the same upstream gradient: the per-layer launch chain, then the fused launch."""
    from syncvsr_amd import model as M
    from syncvsr_amd import ops

    model, out = _tapes(dev, B, T, layers, training, seed)
    h, tape = out[True]
    st = model.store()
    dh = (torch.randn(h.shape, generator=torch.Generator().manual_seed(5 + B)) * 1e-2).to(torch.bfloat16).to(dev)
    res = {}
    for fused in (False, True):
        ops.ENC_BWD_FUSED = fused
        try:
            st.zero_grad()
            st.rebind_grads()
            model._wg_group = None
            dfeats = M._encoder_backward(model, st, tape, dh, B, T)
            M._flush_deferred(model)
            model._side.join()
            torch.cuda.synchronize()
            res[fused] = (dfeats.clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if n.startswith(("encoder.", "cls_token"))})
        finally:
            ops.ENC_BWD_FUSED = True
    ws = ops._ENC_BWD_WS[next(iter(ops._ENC_BWD_WS))]
    assert int(ws.view(torch.int32)[B].item()) == 0, "a bounded cluster wait gave up"
    return res


@pytest.mark.parametrize("B,T,layers,training", [(2, 7, 2, False), (3, 29, 2, True), (32, 29, 6, True), (5, 31, 1, True), (40, 11, 1, False)])
def test_fused_encoder_backward_matches_launch_chain(dev, B, T, layers, training):
    """The fused backward (svsr_enc_bwd) keeps the chain's rounding points, dropout masks and K-half split; what differs is the order of a few
    fp32 additions (the LayerNorm parameter sums: one partial row per sequence instead of one per four rows; the attention backward's row
    sums).  Input gradient and every parameter gradient of the encoder: relative L2 <= 2e-3 against the chain (measured: see the print)."""
    res = _backward_pair(dev, B, T, layers, training)
    (d0, g0), (d1, g1) = res[False], res[True]
    worst = [(_rel(d1, d0), "dfeats")]
    for n in g0:
        if float(g0[n].float().norm()) == 0.0:
            assert float(g1[n].float().norm()) == 0.0, n
            continue
        if n.endswith("attention.self.key.bias"):
            # zero in exact arithmetic (a key bias shifts every score of a query alike): rounding noise on both sides, measured against the
            # query bias's gradient instead of compared
            qn = n.replace(".key.", ".query.")
            noise = float((g1[n].float() - g0[n].float()).norm() / g0[qn].float().norm())
            assert noise <= 2e-3, (n, noise)
            continue
        worst.append((_rel(g1[n], g0[n]), n))
    worst.sort(reverse=True)
    print("largest relative L2 differences:", "  ".join(f"{r:.2e} {n}" for r, n in worst[:8]))
    assert worst[0][0] <= 2e-3, "  ".join(f"{r:.2e} {n}" for r, n in worst[:8])


def test_fused_encoder_backward_is_bit_reproducible(dev):
    a = _backward_pair(dev, 32, 29, 6, True)[True]
    b = _backward_pair(dev, 32, 29, 6, True)[True]
    assert torch.equal(a[0], b[0])
    for n in a[1]:
        assert torch.equal(a[1][n], b[1][n]), n
