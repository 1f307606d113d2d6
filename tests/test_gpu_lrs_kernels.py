"""GPU parity of the LRS-specific C-ABI entry points against plain torch / the LRS oracle's own functions."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _dev():
    return torch.device("cuda:0")


def _r(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def _rel_err(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.mark.parametrize("flash", [False, True])
@pytest.mark.parametrize("B,H,T,lens", [(2, 2, 9, [9, 6]), (3, 3, 50, [50, 33, 41]), (2, 12, 150, [150, 97]), (2, 2, 300, [300, 77])])
def test_rel_mha_fwd_bwd(B, H, T, lens, flash):
    """Conformer rel-pos attention (attention.py:191-278) on bf16-rounded inputs vs the oracle's closed form in fp32; flash: the streamed-key
    kernels (mha_flash.h: online softmax, no stored probabilities; 300 frames = two workgroups of five query tiles per (clip, head))."""
    from syncvsr_amd import ops
    dev = _dev()
    D = H * 64
    qkv = _r(B * T, 3 * D, seed=1).to(BF)
    pe = _r(2 * T - 1, D, seed=2).to(BF)
    u = _r(H, 64, seed=3, scale=0.5)
    v = _r(H, 64, seed=4, scale=0.5)
    dctx = _r(B * T, D, seed=5).to(BF)
    klen = torch.tensor(lens, dtype=torch.int32)

    # reference in fp32 on the rounded inputs
    qf = qkv.float().view(B, T, 3, H, 64).requires_grad_(True)
    pef = pe.float().view(2 * T - 1, H, 64).requires_grad_(True)
    uf, vf = u.clone().requires_grad_(True), v.clone().requires_grad_(True)
    q, k, val = qf[:, :, 0], qf[:, :, 1].transpose(1, 2), qf[:, :, 2].transpose(1, 2)
    qu = (q + uf).detach().to(BF).float() + ((q + uf) - (q + uf).detach())            # bf16-rounded forward value, identity gradient
    qv = (q + vf).detach().to(BF).float() + ((q + vf) - (q + vf).detach())
    ac = torch.matmul(qu.transpose(1, 2), k.transpose(-2, -1))
    bd_full = torch.matmul(qv.transpose(1, 2), pef.permute(1, 2, 0))
    idx = (T - 1) + torch.arange(T).view(1, T) - torch.arange(T).view(T, 1)
    bd = bd_full.gather(-1, idx.expand(B, H, T, T))
    scores = (ac + bd) / 8.0
    mask = (torch.arange(T).view(1, T) < klen.view(B, 1)).view(B, 1, 1, T)
    attn = torch.softmax(scores.masked_fill(~mask, -1e10), -1).masked_fill(~mask, 0.0)
    ctx_ref = torch.matmul(attn, val).transpose(1, 2).reshape(B * T, D)
    ctx_ref.backward(dctx.float())

    qkv_d, pe_d = qkv.to(dev), pe.to(dev)
    ctx, probs = ops.mha_fwd(qkv_d, 3 * D, qkv_d[:, D:], qkv_d[:, 2 * D:], 3 * D, B=B, H=H, Lq=T, Lk=T, pe=pe_d, bias_u=u.to(dev).contiguous(),
                             bias_v=v.to(dev).contiguous(), klen=klen.to(dev), flash=flash)
    assert _rel_err(ctx.float().cpu(), ctx_ref.detach()) < 1.5e-2
    if flash:
        assert isinstance(probs, ops.MhaLse)
        lse_ref = torch.logsumexp(scores.masked_fill(~mask, -float("inf")), -1).detach()
        assert _rel_err(probs.lse.cpu().view(B, H, T), lse_ref) < 2e-3
    else:
        assert _rel_err(probs[:, :, :T].float().cpu().view(B, H, T, T), attn.detach()) < 1.5e-2
    dqkv = torch.empty_like(qkv_d)
    dq_ac, dq_bd, dpe = ops.mha_bwd(dctx.to(dev), qkv_d, 3 * D, qkv_d[:, D:], qkv_d[:, 2 * D:], 3 * D, probs, B=B, H=H, Lq=T, Lk=T,
                                    dq=dqkv, dq_pitch=3 * D, dk=dqkv[:, D:], dv=dqkv[:, 2 * D:], dkv_pitch=3 * D, pe=pe_d,
                                    bias_u=u.to(dev).contiguous(), bias_v=v.to(dev).contiguous())
    g = qf.grad.view(B * T, 3 * D)
    got = dqkv.float().cpu()
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert _rel_err(got[:, sl], g[:, sl]) < 2.5e-2, name
    assert _rel_err(dpe.float().cpu(), pef.grad.reshape(2 * T - 1, D)) < 2.5e-2
    assert _rel_err(dq_ac.float().sum(0).cpu(), uf.grad.flatten()) < 3e-2
    assert _rel_err(dq_bd.float().sum(0).cpu(), vf.grad.flatten()) < 3e-2


@pytest.mark.parametrize("flash", [False, True])
@pytest.mark.parametrize("causal,Lq,Lk", [(True, 7, 7), (False, 11, 40), (True, 41, 41), (False, 41, 150), (True, 290, 290), (False, 70, 333)])
def test_plain_mha_fwd_bwd(causal, Lq, Lk, flash):
    """Decoder self- (causal) and source- (key padding) attention, attention.py:38-108."""
    from syncvsr_amd import ops
    dev = _dev()
    B, H = 3, 2
    D = H * 64
    q = _r(B * Lq, D, seed=1).to(BF)
    kv = _r(B * Lk, 2 * D, seed=2).to(BF)
    dctx = _r(B * Lq, D, seed=3).to(BF)
    klen = None if causal else torch.tensor([Lk, max(1, Lk // 2), max(1, Lk - 3)], dtype=torch.int32)
    qf = q.float().view(B, Lq, H, 64).requires_grad_(True)
    kvf = kv.float().view(B, Lk, 2, H, 64).requires_grad_(True)
    scores = torch.matmul(qf.transpose(1, 2), kvf[:, :, 0].permute(0, 2, 3, 1)) / 8.0
    if causal:
        mask = torch.tril(torch.ones(Lq, Lk, dtype=torch.bool)).view(1, 1, Lq, Lk)
    else:
        mask = (torch.arange(Lk).view(1, Lk) < klen.view(B, 1)).view(B, 1, 1, Lk)
    attn = torch.softmax(scores.masked_fill(~mask, -1e10), -1).masked_fill(~mask, 0.0)
    ctx_ref = torch.matmul(attn, kvf[:, :, 1].transpose(1, 2)).transpose(1, 2).reshape(B * Lq, D)
    ctx_ref.backward(dctx.float())
    q_d, kv_d = q.to(dev), kv.to(dev)
    ctx, probs = ops.mha_fwd(q_d, D, kv_d, kv_d[:, D:], 2 * D, B=B, H=H, Lq=Lq, Lk=Lk, klen=None if klen is None else klen.to(dev), causal=causal, flash=flash)
    assert _rel_err(ctx.float().cpu(), ctx_ref.detach()) < 1.5e-2
    dq = torch.empty_like(q_d)
    dkv = torch.empty_like(kv_d)
    ops.mha_bwd(dctx.to(dev), q_d, D, kv_d, kv_d[:, D:], 2 * D, probs, B=B, H=H, Lq=Lq, Lk=Lk, dq=dq, dq_pitch=D, dk=dkv, dv=dkv[:, D:],
                dkv_pitch=2 * D)
    assert _rel_err(dq.float().cpu(), qf.grad.reshape(B * Lq, D)) < 2.5e-2
    assert _rel_err(dkv.float().cpu(), kvf.grad.reshape(B * Lk, 2 * D)) < 2.5e-2


@pytest.mark.parametrize("rel", [True, False])
def test_flash_attention_draws_the_same_dropout_masks_as_the_per_tile_kernels(rel):
    """Attention dropout (attention.py:80) is decided by hash(seed, site, index into the [B*H][Lq][ldp] probability tensor) in both
    implementations (the oracles replay exactly these masks): with dropout 0.3 the streamed-key kernels must reproduce the per-tile kernels'
    outputs and gradients to bf16 rounding — a different mask would change ctx by O(1)."""
    from syncvsr_amd import ops
    dev = _dev()
    B, H, T = 2, 3, 75
    D = H * 64
    qkv = _r(B * T, 3 * D, seed=1).to(BF).to(dev)
    pe = _r(2 * T - 1, D, seed=2).to(BF).to(dev) if rel else None
    u = _r(H, 64, seed=3, scale=0.5).to(dev).contiguous() if rel else None
    v = _r(H, 64, seed=4, scale=0.5).to(dev).contiguous() if rel else None
    dctx = _r(B * T, D, seed=5).to(BF).to(dev)
    klen = torch.tensor([T, 40], dtype=torch.int32, device=dev)
    seed = torch.tensor([777], dtype=torch.int32, device=dev)
    drop = (seed, 5, 0.3)
    res = {}
    for flash in (False, True):
        ctx, keep = ops.mha_fwd(qkv, 3 * D, qkv[:, D:], qkv[:, 2 * D:], 3 * D, B=B, H=H, Lq=T, Lk=T, pe=pe, bias_u=u, bias_v=v, klen=klen,
                                causal=not rel, drop=drop, flash=flash)
        dqkv = torch.zeros_like(qkv)
        aux = ops.mha_bwd(dctx, qkv, 3 * D, qkv[:, D:], qkv[:, 2 * D:], 3 * D, keep, B=B, H=H, Lq=T, Lk=T, dq=dqkv, dq_pitch=3 * D, dk=dqkv[:, D:],
                          dv=dqkv[:, 2 * D:], dkv_pitch=3 * D, pe=pe, bias_u=u, bias_v=v, drop=drop)
        res[flash] = (ctx.float().cpu(), dqkv.float().cpu(), None if not rel else aux[2].float().cpu())
    assert _rel_err(res[True][0], res[False][0]) < 1e-2
    assert _rel_err(res[True][1], res[False][1]) < 2e-2
    if rel:
        assert _rel_err(res[True][2], res[False][2]) < 2e-2


@pytest.mark.parametrize("T", [75, 160])
def test_flash_attention_backward_in_parts_and_with_the_table_made_ahead_equals_the_single_call(T):
    """svsr_mha_flash_bwd_parts (query + key passes, then the position-table pass as a call of its own — on ANOTHER stream behind an event, as the
    sentence-level model issues it) and svsr_mha_pe_transpose (the transposed position table made ahead of time) against svsr_mha_flash_bwd:
    dq / dk / dv, dq_ac, dq_bd and dpe bit for bit."""
    from syncvsr_amd import ops
    dev = _dev()
    B, H = 2, 3
    D = H * 64
    qkv = _r(B * T, 3 * D, seed=1).to(BF).to(dev)
    pe = _r(2 * T - 1, D, seed=2).to(BF).to(dev)
    u = _r(H, 64, seed=3, scale=0.5).to(dev).contiguous()
    v = _r(H, 64, seed=4, scale=0.5).to(dev).contiguous()
    dctx = _r(B * T, D, seed=5).to(BF).to(dev)
    klen = torch.tensor([T, T - 35], dtype=torch.int32, device=dev)
    seed = torch.tensor([777], dtype=torch.int32, device=dev)
    drop = (seed, 5, 0.2)
    ctx, rec = ops.mha_fwd(qkv, 3 * D, qkv[:, D:], qkv[:, 2 * D:], 3 * D, B=B, H=H, Lq=T, Lk=T, pe=pe, bias_u=u, bias_v=v, klen=klen, drop=drop, flash=True)

    def bwd(**kw):
        dqkv = torch.zeros_like(qkv)
        out = ops.mha_bwd(dctx, qkv, 3 * D, qkv[:, D:], qkv[:, 2 * D:], 3 * D, rec, B=B, H=H, Lq=T, Lk=T, dq=dqkv, dq_pitch=3 * D, dk=dqkv[:, D:],
                          dv=dqkv[:, 2 * D:], dkv_pitch=3 * D, pe=pe, bias_u=u, bias_v=v, drop=drop, **kw)
        return dqkv, out

    dq0, (ac0, bd0, dpe0) = bwd()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    pet = ops.mha_pe_transpose(pe, H, T)
    dq1, out1 = bwd(pe_later=True, pet=pet)
    assert len(out1) == 4
    ops.stream_wait(side, torch.cuda.current_stream())
    ops.STREAM_OVERRIDE = side.cuda_stream
    try:
        out1[3][0]()                         # the position-table pass, on the other stream
    finally:
        ops.STREAM_OVERRIDE = None
    torch.cuda.synchronize()
    for a, b, name in ((dq1, dq0, "dqkv"), (out1[0], ac0, "dq_ac"), (out1[1], bd0, "dq_bd"), (out1[2], dpe0, "dpe")):
        assert torch.equal(a, b), name


@pytest.mark.parametrize("B,T,D,K", [(2, 9, 128, 31), (2, 150, 768, 31)])
def test_glu_dwconv_backward_in_parts_equals_the_single_call(B, T, D, K):
    """svsr_glu_dwconv_bwd_parts: the pass over the activations, then the sum of the weight gradient's partial rows as a call of its own."""
    from syncvsr_amd import ops
    dev = _dev()
    u = _r(B * T, 2 * D, seed=1).to(BF).to(dev)
    w = _r(D, K, seed=2, scale=1 / math.sqrt(K)).to(dev)
    dc = _r(B * T, D, seed=4).to(BF).to(dev)
    res = []
    for later in (False, True):
        dw, db = torch.zeros(D, K, device=dev), torch.zeros(D, device=dev)
        out = ops.glu_dwconv_bwd(dc, u, w, dw, db, B, T, D, K, reduce_later=later)
        if later:
            du, (fn, keep) = out
            assert float(dw.abs().sum()) == 0.0          # nothing summed yet
            fn()
        else:
            du = out
        torch.cuda.synchronize()
        res.append((du, dw, db))
    for a, b, name in zip(res[1], res[0], ("du", "dw", "dbias")):
        assert torch.equal(a, b), name


@pytest.mark.parametrize("B,T,D,K", [(2, 9, 128, 31), (3, 70, 128, 7), (2, 150, 768, 31)])
def test_glu_dwconv(B, T, D, K):
    from syncvsr_amd import ops
    dev = _dev()
    u = _r(B * T, 2 * D, seed=1).to(BF)
    w = _r(D, K, seed=2, scale=1 / math.sqrt(K))
    bias = _r(D, seed=3, scale=0.1)
    dc = _r(B * T, D, seed=4).to(BF)
    uf = u.float().view(B, T, 2 * D).requires_grad_(True)
    wf, bf_ = w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    g = uf[..., :D] * torch.sigmoid(uf[..., D:])
    c_ref = F.conv1d(g.transpose(1, 2), wf.view(D, 1, K), bf_, padding=(K - 1) // 2, groups=D).transpose(1, 2).reshape(B * T, D)
    c_ref.backward(dc.float())
    c, stats = ops.glu_dwconv_fwd(u.to(dev), w.to(dev), bias.to(dev), True, B, T, D, K)
    assert _rel_err(c.float().cpu(), c_ref.detach()) < 6e-3
    st = stats[0][: stats[1] * 2 * D].view(stats[1], 2, D).double().sum(0).float().cpu()
    assert _rel_err(st[0], c_ref.detach().sum(0)) < 1e-3 + 1e-3
    assert _rel_err(st[1], (c_ref.detach() ** 2).sum(0)) < 2e-3
    dw = torch.zeros(D, K, device=dev)
    db = torch.zeros(D, device=dev)
    du = ops.glu_dwconv_bwd(dc.to(dev), u.to(dev), w.to(dev), dw, db, B, T, D, K)
    assert _rel_err(du.float().cpu(), uf.grad.reshape(B * T, 2 * D)) < 8e-3
    assert _rel_err(dw.cpu(), wf.grad) < 5e-3
    assert _rel_err(db.cpu(), bf_.grad) < 5e-3


def test_lrs_targets_equal_add_sos_eos():
    """svsr_lrs_targets (one launch) against the oracle's add_sos_eos (reference add_sos_eos.py:10-31) and the CTC label form: rows of every
    length from 1 to L (a full row has no padding), 300-token rows (more than one pass of the workgroup)."""
    from oracle import lrs_oracle as O
    from syncvsr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    odim, ignore = 5049, -1
    for B, L in ((7, 7), (16, 41), (3, 300)):
        lens = [L, 1] + [int(x) for x in torch.randint(1, L + 1, (B - 2,), generator=g)]
        label = torch.full((B, L), ignore, dtype=torch.int64)
        for b, n in enumerate(lens):
            label[b, :n] = torch.randint(1, odim, (n,), generator=g)
        labels, ys_in, ys_out = ops.lrs_targets(label.to(dev), odim, ignore, odim - 1)
        ref_in, ref_out = O.add_sos_eos([y[y != ignore] for y in label], odim - 1, odim - 1)
        assert torch.equal(ys_in.cpu(), ref_in) and torch.equal(ys_out.cpu(), ref_out)
        assert torch.equal(labels.cpu(), label)
        # ignore_id INSIDE a row is dropped, as the reference's `y[y != ignore_id]` drops it (add_sos_eos.py:26-27, ctc.py's own filter)
        holes = label.clone()
        holes[torch.rand(B, L, generator=g) < 0.3] = ignore
        holes[0, 0] = 7                     # (at least one live token per row keeps the reference's pad_list happy)
        holes[:, -1] = torch.where(holes[:, -1] == ignore, torch.ones_like(holes[:, -1]), holes[:, -1])
        labels, ys_in, ys_out = ops.lrs_targets(holes.to(dev), odim, ignore, odim - 1)
        live = [y[y != ignore] for y in holes]
        ref_in, ref_out = O.add_sos_eos(live, odim - 1, odim - 1)
        pad = L + 1 - ref_in.shape[1]       # (the reference pads to the longest LIVE row; the kernel to L + 1)
        ref_in = torch.nn.functional.pad(ref_in, (0, pad), value=odim - 1)
        ref_out = torch.nn.functional.pad(ref_out, (0, pad), value=ignore)
        assert torch.equal(ys_in.cpu(), ref_in) and torch.equal(ys_out.cpu(), ref_out)
        want = torch.full((B, L), -1, dtype=torch.int64)
        for b, y in enumerate(live):
            want[b, : len(y)] = y
        assert torch.equal(labels.cpu(), want)
    # a label outside [1, odim) does not trap: it becomes eos and a sticky error word is set
    assert not ops.lrs_target_errors(reset=True)
    bad = torch.tensor([[3, odim + 5, 4, ignore], [0, 2, ignore, ignore]], dtype=torch.int64)
    labels, ys_in, ys_out = ops.lrs_targets(bad.to(dev), odim, ignore, odim - 1)
    assert labels.cpu().tolist() == [[3, odim - 1, 4, -1], [odim - 1, 2, -1, -1]]
    assert ops.lrs_target_errors(reset=True) and not ops.lrs_target_errors(reset=True)


@pytest.mark.parametrize("B,T,V,lens,ylens", [(2, 7, 11, [7, 5], [3, 2]), (3, 40, 41, [40, 25, 31], [10, 4, 12]), (4, 150, 5049, [150, 90, 120, 6], [40, 12, 25, 8])])
def test_ctc(B, T, V, lens, ylens):
    """Matches torch.nn.CTCLoss(reduction='sum', zero_infinity=True)/B on log_softmax, ctc.py:65-74; last case has an infeasible item."""
    from syncvsr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(7)
    Vp = (V + 63) // 64 * 64
    z = torch.randn(B * T, Vp, generator=g)
    Lmax = max(ylens)
    labels = torch.full((B, Lmax), -1, dtype=torch.long)
    ys = []
    for b, n in enumerate(ylens):
        y = torch.randint(1, V, (n,), generator=g)
        if n > 2:
            y[1] = y[0]                     # a repeat (needs a blank in between)
        labels[b, :n] = y
        ys.append(y)
    zf = z[:, :V].clone().view(B, T, V).requires_grad_(True)
    lp = zf.transpose(0, 1).log_softmax(2)
    ref = F.ctc_loss(lp, torch.cat(ys), torch.tensor(lens), torch.tensor(ylens), blank=0, reduction="sum", zero_infinity=True) / B
    ref.backward()
    zd = z.to(dev)
    ilen = torch.tensor(lens, dtype=torch.int32, device=dev)
    loss, state = ops.ctc_fwd(zd, Vp, labels.to(dev), ilen, B, T, V)
    assert abs(loss.item() - ref.item()) <= 2e-4 * max(1.0, abs(ref.item()))
    gout = torch.ones((), device=dev)
    dz = ops.ctc_grad(zd, Vp, labels.to(dev), ilen, B, T, V, state, gout, Vp)
    got = dz.float().cpu()
    assert torch.all(got[:, V:] == 0)
    assert _rel_err(got[:, :V], zf.grad.reshape(B * T, V)) < 6e-3


def test_embed_pos_and_ls_loss():
    from oracle import lrs_oracle as O
    from syncvsr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    B, L, D, V = 3, 9, 128, 41
    tok = torch.randint(0, V, (B, L), generator=g)
    emb = torch.randn(V, D, generator=g)
    pe = O.abs_pos_emb(L, D)
    x = ops.embed_pos_fwd(tok.to(dev), emb.to(dev), pe.to(dev), L, D, math.sqrt(D))
    ref = F.embedding(tok, emb) * math.sqrt(D) + pe
    assert _rel_err(x.float().cpu(), ref.reshape(B * L, D)) < 4e-3
    dx = torch.randn(B * L, D, generator=g).to(BF)
    demb = torch.zeros(V, D, device=dev)
    ops.embed_pos_bwd(tok.to(dev), dx.to(dev), demb, D, math.sqrt(D))
    ref_d = torch.zeros(V, D).index_add_(0, tok.flatten(), dx.float() * math.sqrt(D))
    assert _rel_err(demb.cpu(), ref_d) < 1e-5

    Vp = 64
    z = torch.randn(B * L, Vp, generator=g)
    target = torch.randint(0, V, (B, L), generator=g)
    target[0, 6:] = -1
    target[2, 4:] = -1
    zf = z[:, :V].clone().view(B, L, V).requires_grad_(True)
    for smoothing, norm_len in ((0.1, False), (0.0, False), (0.2, True)):
        zf.grad = None
        ref_loss = O.label_smoothing_loss(zf, target, smoothing, norm_len)
        ref_loss.backward()
        live = int((target >= 0).sum())
        inv = 1.0 / live if norm_len else 1.0 / B
        loss, lse, counts = ops.ls_loss_fwd(z.to(dev), Vp, target.to(dev), B * L, V, smoothing, inv)
        assert abs(loss.item() - ref_loss.item()) <= 2e-5 * max(1.0, abs(ref_loss.item()))
        assert counts[1].item() == live
        assert abs(counts[0].item() / live - O.th_accuracy(zf.detach(), target)) < 1e-6
        dz = ops.ls_loss_bwd(z.to(dev), Vp, target.to(dev), B * L, V, smoothing, inv, lse, torch.ones((), device=dev), Vp)
        assert _rel_err(dz.float().cpu()[:, :V], zf.grad.reshape(B * L, V)) < 6e-3


def test_ln768_pre_norm_and_relu_alpha_epilogues():
    """LayerNorm at D=768 (+ skip-path addend in the backward), GEMM epilogue act=ReLU / alpha, ReLU bias backward, scale."""
    from syncvsr_amd import ops
    dev = _dev()
    R, D, N = 70, 768, 256
    x = _r(R, D, seed=1).to(BF)
    gam, bet = 1 + 0.1 * _r(D, seed=2), 0.1 * _r(D, seed=3)
    y, mean, rstd = ops.add_ln_fwd(x.to(dev), None, gam.to(dev), bet.to(dev), 1e-12)
    xf = x.float().requires_grad_(True)
    gf, bf_ = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    ref = F.layer_norm(xf, (D,), gf, bf_, 1e-12)
    assert _rel_err(y.float().cpu(), ref.detach()) < 4e-3
    dy = _r(R, D, seed=4).to(BF)
    skip = _r(R, D, seed=5).to(BF)
    ref.backward(dy.float())
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    ds = ops.add_ln_bwd(dy.to(dev), x.to(dev), None, gam.to(dev), mean, rstd, dg, db, addend=skip.to(dev))
    assert _rel_err(ds.float().cpu(), xf.grad + skip.float()) < 6e-3
    assert _rel_err(dg.cpu(), gf.grad) < 5e-3 and _rel_err(db.cpu(), bf_.grad) < 5e-3

    w = (_r(N, D, seed=6) / math.sqrt(D)).to(BF)
    b = 0.1 * _r(N, seed=7)
    res = _r(R, N, seed=8).to(BF)
    out, _ = ops.linear_fwd(x.to(dev), w.to(dev), b.to(dev), rows=R, K=D, N=N, x_pitch=D, addend=res.to(dev), alpha=0.5)
    ref = res.float() + 0.5 * (x.float() @ w.float().t() + b)
    assert _rel_err(out.float().cpu(), ref) < 5e-3
    out, _ = ops.linear_fwd(x.to(dev), w.to(dev), b.to(dev), rows=R, K=D, N=N, x_pitch=D, relu=True)
    ref = torch.relu(x.float() @ w.float().t() + b)
    assert _rel_err(out.float().cpu(), ref) < 5e-3
    dyn = _r(R, N, seed=9).to(BF)
    dbias = torch.zeros(N, device=dev)
    dz = ops.bias_act_bwd(dyn.to(dev), out, dbias, R=R, N=N, n_valid=N, ld=N, relu=True)
    refdz = dyn.float() * (ref > 0)
    assert _rel_err(dz.float().cpu(), refdz) < 1e-6 + 4e-3
    assert _rel_err(dbias.cpu(), refdz.sum(0)) < 4e-3
    s = ops.scale_bf16(dyn.to(dev), 0.5)
    assert _rel_err(s.float().cpu(), dyn.float() * 0.5) < 1e-6
    # the residual branch's gradient as a second output of the LayerNorm backward (svsr_add_ln_bwd_branch) == svsr_scale_bf16 of its first output,
    # bit for bit, with and without a dropout mask; the first output and the parameter-gradient partials are those of the plain launch
    seed = torch.tensor([12345], dtype=torch.int32, device=dev)
    for alpha, drop in ((0.5, (seed, 77, 0.1)), (1.0, (seed, 5, 0.25)), (0.5, None)):
        dg1, db1, dg2, db2 = (torch.zeros(D, device=dev) for _ in range(4))
        d1, d2 = [], []
        ds1 = ops.add_ln_bwd(dy.to(dev), x.to(dev), None, gam.to(dev), mean, rstd, dg1, db1, addend=skip.to(dev), defer=d1)
        ds2, br = ops.add_ln_bwd(dy.to(dev), x.to(dev), None, gam.to(dev), mean, rstd, dg2, db2, addend=skip.to(dev), defer=d2, branch=(alpha, drop))
        for fn, _ in d1 + d2:
            fn()
        assert torch.equal(ds1, ds2) and torch.equal(dg1, dg2) and torch.equal(db1, db2)
        assert torch.equal(br, ops.scale_bf16(ds1, alpha, drop=drop))
        if drop is not None:
            assert 0.02 < float((br == 0).float().mean()) < 0.4


@pytest.mark.parametrize("C,res", [(64, True), (768, False), (128, False)])
def test_bn_swish(C, res):
    """BatchNorm + Swish (LRS resnet.py:90-107, convolution.py:69) forward/backward incl. C=768 rows layout."""
    from syncvsr_amd import ops
    dev = _dev()
    N = 300
    x = _r(N, C, seed=1).to(BF)
    r = _r(N, C, seed=2).to(BF) if res else None
    gam, bet = 1 + 0.1 * _r(C, seed=3), 0.1 * _r(C, seed=4)
    dy = _r(N, C, seed=5).to(BF)
    xf = x.float().requires_grad_(True)
    rf = r.float().requires_grad_(True) if res else None
    gf, bf_ = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    mean = xf.mean(0)
    var = xf.var(0, unbiased=False)
    z = (xf - mean) * torch.rsqrt(var + 1e-5) * gf + bf_ + (rf if res else 0)
    yref = z * torch.sigmoid(z)
    yref.backward(dy.float())
    md, rd = mean.detach().to(dev), torch.rsqrt(var.detach() + 1e-5).to(dev)
    xd = x.to(dev).view(N, 1, 1, C)
    rdv = r.to(dev).view(N, 1, 1, C) if res else None
    y = ops.bn_act_fwd(xd, rdv, md, rd, gam.to(dev), bet.to(dev), 2)
    assert _rel_err(y.float().cpu().view(N, C), yref.detach()) < 5e-3
    coef = torch.empty(3 * C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dx, dres = ops.bn_act_bwd(dy.to(dev).view(N, 1, 1, C), y, xd, md, rd, gam.to(dev), coef, dg, db, 2, res, beta=bet.to(dev), res=rdv)
    assert _rel_err(dx.float().cpu().view(N, C), xf.grad) < 1e-2
    assert _rel_err(dg.cpu(), gf.grad) < 6e-3 and _rel_err(db.cpu(), bf_.grad) < 6e-3
    if res:
        assert _rel_err(dres.float().cpu().view(N, C), rf.grad) < 6e-3


@pytest.mark.parametrize("R,K,N,pitch", [(960, 512, 1536, 1536), (2400, 768, 5049, 5056), (70, 768, 256, 256), (32, 512, 500, 512)])
def test_wgrad_fused_bias_gradient(R, K, N, pitch):
    """svsr_igemm_wgrad's optional dbias output = column sums of dy (accumulated), alongside the weight gradient."""
    from syncvsr_amd import ops
    dev = _dev()
    x = _r(R, K, seed=1).to(BF)
    dy = torch.zeros(R, pitch)
    dy[:, :N] = _r(R, N, seed=2)
    dy = dy.to(BF)
    dw = torch.zeros(N, K, device=dev)
    db = torch.full((N,), 0.5, device=dev)
    ops.linear_wgrad(x.to(dev), dy.to(dev), dw, rows=R, K=K, N=N, x_pitch=K, dy_pitch=pitch, db=db)
    assert _rel_err(dw.cpu(), dy.float()[:, :N].t() @ x.float()) < 3e-3
    assert _rel_err(db.cpu() - 0.5, dy.float()[:, :N].sum(0)) < 2e-3
