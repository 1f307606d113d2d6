"""Whole-path parity (-m gpu): the HIP model against oracle/lrw_oracle.py (fp32 CPU) on identical seeded inputs and
weights, plus the committed reference goldens.  The HIP path computes in bf16 storage / fp32 accumulation, so:
  * losses: |hip - oracle| <= 1e-3 * |oracle| on the full-size cases (north_star's bound; measured 6e-4 and below) and
    <= 6e-3 on the tiny cases, whose BatchNorm statistics come from as few as 96 values per channel (measured <= 2e-3)
  * features / word logits: relative L2 error <= 3e-2; audio logits (after 6 more bf16 layers) <= 8e-2
  * parameter gradients at B = 2: min cosine >= 0.85, median cosine >= 0.995, norm ratio within 15 % for every tensor whose
    oracle norm is not numerically zero (key biases are analytically zero) — the fidelity of torch's own bf16 autocast on
    this case is min 0.878 / median 0.998, see DESIGN.md.  The benchmark batch (B = 32, test_benchmark_batch_matches_*) measures
    the same floor (min 0.890, median 0.9988, ratios 0.935 .. 1.096): it is not a statistics effect but the bf16 forward itself
    — activations rounded to bf16 flip ~1 % of the ReLU masks per layer relative to the fp32 run, and the early trunk tensors
    see that through 17 BatchNorm/ReLU stages; run-to-run the HIP gradients are bit-identical (tests/test_gpu_train.py).
"""
import json
import os

import numpy as np
import pytest
import torch

from golden_cases import build_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _run_pair(name, dev):
    from oracle import lrw_oracle as O
    from syncvsr_amd.model import Model

    cfg, sd, batch, training, gold = build_case(name)
    model = Model(cfg)
    model.keep_audio_logits = True          # logits_audio for the comparisons below (the loss itself comes from the fused audio head)
    missing, unexpected = model.load_state_dict(sd, strict=True)
    model.to(dev).train(training)
    gbatch = [t.to(dev) for t in batch]
    out = model(*gbatch)
    if training:
        out["loss_total"].backward()
    torch.cuda.synchronize()
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    keep, stats = {}, {}
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    ref = O.forward(osd, cfg, *batch, training=training, keep=keep, stats_out=stats)
    if training:
        ref["loss_total"].backward()
    return cfg, model, out, osd, ref, keep, stats, gold


def _emu_grad_stats(cfg, sd, batch, model):
    """HIP gradients against the oracle's bf16-storage emulation (oracle.lrw_oracle.forward(emu=True)): -> (trunk rows, other rows) of
    (cosine, norm ratio, name), sorted.  attention.self.key.bias is left out: softmax is shift-invariant, its gradient is analytically 0."""
    from oracle import lrw_oracle as O

    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    O.forward(osd, cfg, *batch, training=True, emu=True)["loss_total"].backward()
    rows = []
    for n, p in model.named_parameters():
        g, r = p.grad.detach().float().cpu().flatten(), osd[n].grad.flatten()
        if r.norm().item() > 1e-6 and not n.endswith("attention.self.key.bias"):
            rows.append((float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-30)), float(g.norm() / r.norm()), n))
    rows.sort()
    return [r for r in rows if r[2].startswith(("resnet.", "stem3d."))], [r for r in rows if not r[2].startswith(("resnet.", "stem3d."))]


def _report(name, rows):
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", f"parity_{name}.json"), "w") as f:
        json.dump(rows, f, indent=1)


@pytest.mark.parametrize("name,loss_tol", [("lrw_full_b2", 1e-3), ("lrw_tiny", 6e-3), ("lrw_tiny_soft_ls", 6e-3), ("lrw_tiny_hard_ls", 6e-3)])
def test_model_matches_oracle(dev, name, loss_tol):
    cfg, model, out, osd, ref, keep, stats, gold = _run_pair(name, dev)
    rows = {}
    for k in ("loss_total", "loss_category", "loss_audio"):
        rows[k] = dict(hip=out[k].item(), oracle=ref[k].item(), golden=float(gold[k]))
    last = model._last
    B = keep["feats"].shape[0]
    def rel(a, b):
        a, b = a.detach().float().cpu().flatten(), b.detach().float().flatten()
        return ((a - b).norm() / (b.norm() + 1e-30)).item()
    rows["rel.feats"] = rel(last["feats"], keep["feats"])
    rows["rel.hidden"] = rel(last["hidden"], keep["hidden"])
    rows["rel.logits_category"] = rel(last["logits_category"], keep["logits_category"])
    rows["rel.logits_audio"] = rel(last["logits_audio"], keep["logits_audio"])
    grads = {}
    st = model.store()
    for n, p in model.named_parameters():
        g = p.grad.detach().float().cpu().flatten()
        r = osd[n].grad.flatten()
        rn = r.norm().item()
        grads[n] = dict(cos=float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-30)), ratio=float(g.norm() / (rn + 1e-30)), ref_norm=rn)
    rows["grads"] = grads
    bufs = {}
    for n in ("stem3d.1.running_var", "resnet.layer1.0.bn1.running_mean", "resnet.layer4.1.bn2.running_var"):
        bufs[n] = rel(dict(model.named_buffers())[n], stats[n])
    rows["buffers"] = bufs
    _report(name, rows)
    print(json.dumps({k: v for k, v in rows.items() if k != "grads"}, indent=1))
    worst = sorted(((v["cos"], n) for n, v in grads.items() if v["ref_norm"] > 1e-6))[:5]
    print("worst grad cosines:", worst)
    for k in ("loss_total", "loss_category", "loss_audio"):
        assert abs(rows[k]["hip"] - rows[k]["oracle"]) <= loss_tol * abs(rows[k]["oracle"]), (k, rows[k])
    assert abs(out["accuracy_top1"].item() - ref["accuracy_top1"].item()) < 1e-6 or name != "lrw_full_b2"
    if name == "lrw_full_b2":
        assert rows["rel.feats"] <= 3e-2 and rows["rel.logits_audio"] <= 8e-2 and rows["rel.logits_category"] <= 3e-2, rows
        # Yardstick: the reference's own bf16 path (torch CPU autocast of the oracle) measured against fp32 on this very
        # case gives min cosine 0.878 (stem3d.1.weight), median 0.998 (DESIGN.md "Parity").  The HIP path must be at
        # least that faithful to the fp32 reference.
        coss = sorted(v["cos"] for v in grads.values() if v["ref_norm"] > 1e-6)
        assert coss[0] >= 0.85 and coss[len(coss) // 2] >= 0.995, (coss[0], coss[len(coss) // 2])
        for n, v in grads.items():
            if v["ref_norm"] > 1e-6:
                assert 0.85 <= v["ratio"] <= 1.15, (n, v)
        for n, v in bufs.items():
            assert v <= 1e-2, (n, v)
        # Against the bf16-storage emulation the rounding points agree and only their amplification across the 17 ReLU stages is
        # left (one value landing on the other side of a rounding boundary moves it by one bf16 ulp; ~4 % of a block's outputs do,
        # and the next block multiplies that): measured trunk min 0.935 / median 0.968, norm ratios 0.944-1.078, encoder + heads
        # min 0.9986.  The SHARP statement about the kernels is tests/test_gpu_blockwise.py (every block on its own: >= 0.9999).
        _, sd2, batch2, _, _ = build_case(name)
        trunk, other = _emu_grad_stats(cfg, sd2, batch2, model)
        print("bf16-emulation: trunk worst", trunk[:3], "median", trunk[len(trunk) // 2][0], "encoder/heads worst", other[:2])
        assert trunk[0][0] >= 0.915 and trunk[len(trunk) // 2][0] >= 0.955, (trunk[:3], trunk[len(trunk) // 2])
        assert all(0.92 <= r <= 1.10 for _, r, _ in trunk), sorted(trunk, key=lambda t: -abs(t[1] - 1))[:3]
        assert other[0][0] >= 0.997 and all(0.99 <= r <= 1.012 for _, r, _ in other), (other[:3], sorted(other, key=lambda t: -abs(t[1] - 1))[:3])


def test_benchmark_batch_matches_oracle_and_reference_golden(dev):
    """BASELINE.json configs[1] at its full batch (32 clips = 928 frames): the HIP step launches exactly the kernel
    instantiations bench.py times.  The fp32 oracle is pinned to the golden the reference itself produced for this batch
    (tests/golden/make_golden_lrw.py lrw_full_b32, fp64) and the HIP path is compared with the oracle."""
    cfg, model, out, osd, ref, keep, stats, gold = _run_pair("lrw_full_b32", dev)
    for k in ("loss_total", "loss_category", "loss_audio", "accuracy_top1", "accuracy_top5"):
        assert abs(ref[k].item() - float(gold[k])) <= 2e-5 * max(1.0, abs(float(gold[k]))), ("oracle vs reference golden", k)
    names = [str(n) for n in gold["grad_names"]]
    onorm = np.array([osd[n].grad.double().norm().item() for n in names])
    big = gold["grad_norms"] > 1e-6 * gold["grad_norms"].max()          # key biases have analytically zero gradients
    np.testing.assert_allclose(onorm[big], gold["grad_norms"][big], rtol=5e-3)      # fp32 oracle vs fp64 reference
    rows = {}
    for k in ("loss_total", "loss_category", "loss_audio"):
        rows[k] = dict(hip=out[k].item(), oracle=ref[k].item(), golden=float(gold[k]))
        assert abs(rows[k]["hip"] - rows[k]["oracle"]) <= 1e-3 * abs(rows[k]["oracle"]), (k, rows[k])
    assert abs(out["accuracy_top1"].item() - ref["accuracy_top1"].item()) < 1e-6

    def rel(a, b):
        a, b = a.detach().float().cpu().flatten(), b.detach().float().flatten()
        return ((a - b).norm() / (b.norm() + 1e-30)).item()

    last = model._last
    for k, tol in (("feats", 3e-2), ("logits_category", 3e-2), ("logits_audio", 8e-2)):
        rows[f"rel.{k}"] = rel(last[k], keep[k])
        assert rows[f"rel.{k}"] <= tol, (k, rows[f"rel.{k}"])
    grads = {}
    for n, p in model.named_parameters():
        g = p.grad.detach().float().cpu().flatten()
        r = osd[n].grad.flatten()
        rn = r.norm().item()
        grads[n] = dict(cos=float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-30)), ratio=float(g.norm() / (rn + 1e-30)), ref_norm=rn)
    rows["grads"] = grads
    _report("lrw_full_b32", rows)
    live = {n: v for n, v in grads.items() if v["ref_norm"] > 1e-6}
    coss = sorted((v["cos"], n) for n, v in live.items())
    print(json.dumps({k: v for k, v in rows.items() if k != "grads"}, indent=1))
    print("worst grad cosines:", coss[:5], "median", coss[len(coss) // 2][0])
    assert coss[0][0] >= 0.87 and coss[len(coss) // 2][0] >= 0.998, (coss[:3], coss[len(coss) // 2])
    # encoder and heads sit above the 17 ReLU stages: they must agree tightly
    enc = sorted((v["cos"], n) for n, v in live.items() if not n.startswith(("resnet.", "stem3d.")))
    assert enc[0][0] >= 0.99, enc[:3]
    for n, v in live.items():
        assert 0.86 <= v["ratio"] <= 1.14, (n, v)        # (fp32 reference: measured 0.920-1.124 across builds — mask flips, see below)
    for n in ("stem3d.1.running_var", "resnet.layer1.0.bn1.running_mean", "resnet.layer4.1.bn2.running_var"):
        assert rel(dict(model.named_buffers())[n], stats[n]) <= 1e-2, n
    # bf16-storage emulation of the oracle (same rounding points as the HIP path): measured trunk min 0.950 / median 0.971, norm
    # ratios 0.930-1.063, encoder + heads min 0.9987 and ratios 0.999-1.003.  What is left is the amplification of single-ulp
    # differences through the trunk's depth; the kernels themselves are pinned block by block in tests/test_gpu_blockwise.py.
    _, sd2, batch2, _, _ = build_case("lrw_full_b32")
    trunk, other = _emu_grad_stats(cfg, sd2, batch2, model)
    print("bf16-emulation: trunk worst", trunk[:3], "median", trunk[len(trunk) // 2][0], "encoder/heads worst", other[:2])
    assert trunk[0][0] >= 0.93 and trunk[len(trunk) // 2][0] >= 0.96, (trunk[:3], trunk[len(trunk) // 2])
    # EVERY tensor's gradient norm against the emulation: within 8 % inside the trunk (measured 0.930-1.063), within 0.6 % above it — a 10 %
    # scale error in any single tensor cannot pass
    assert all(0.92 <= r <= 1.08 for _, r, _ in trunk), sorted(trunk, key=lambda t: -abs(t[1] - 1))[:3]
    assert other[0][0] >= 0.9975 and all(0.995 <= r <= 1.006 for _, r, _ in other), (other[:3], sorted(other, key=lambda t: -abs(t[1] - 1))[:3])


def test_dropout_matches_oracle_with_shared_masks(dev):
    """Training forward/backward with emb_dropout 0.05, hidden dropout 0.1, attention dropout 0.15 (HF BertConfig's defaults are
    0.1 / 0.1): the oracle replays the library's counter-based masks (oracle.lrw_oracle.DropPlan), so losses and gradients must
    agree as tightly as without dropout; masks change from step to step and eval mode draws none."""
    from oracle import lrw_oracle as O
    from syncvsr_amd.dropout import keep_mask, lrw_sites
    from syncvsr_amd.model import Model

    cfg, sd, batch, training, gold = build_case("lrw_full_b2")
    cfg = cfg.copy()
    cfg.model.bert.emb_dropout, cfg.model.bert.hidden_dropout_prob, cfg.model.bert.attention_probs_dropout_prob = 0.05, 0.1, 0.15
    model = Model(cfg)
    model.load_state_dict(sd, strict=True)
    model.to(dev).train()
    model.reseed_dropout(41)
    gb = [t.to(dev) for t in batch]
    out = model(*gb)
    out["loss_total"].backward()
    torch.cuda.synchronize()
    assert int(model._drop_word.item()) == 42
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    dp = O.DropPlan(42, 0.1, 0.15, 0.05, lrw_sites(int(cfg.model.bert.num_hidden_layers)))
    keep = {}
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    ref = O.forward(osd, cfg, *batch, training=True, keep=keep, dp=dp)
    ref["loss_total"].backward()
    nodrop = O.forward(sd, cfg, *batch, training=True)
    for k in ("loss_total", "loss_category", "loss_audio"):
        assert abs(out[k].item() - ref[k].item()) <= 2e-3 * abs(ref[k].item()), (k, out[k].item(), ref[k].item())
    assert abs(ref["loss_audio"].item() - nodrop["loss_audio"].item()) > 1e-4 * abs(nodrop["loss_audio"].item()), "dropout had no effect"
    a, b = model._last["hidden"].float().cpu().flatten(), keep["hidden"].flatten()
    assert float((a - b).norm() / b.norm()) <= 7e-2
    coss = []
    for n, p in model.named_parameters():
        if n.startswith(("resnet.", "stem3d.")):
            continue                                   # below the encoder the comparison is the dropout-free one (bf16 ReLU-mask noise)
        g, r = p.grad.detach().float().cpu().flatten(), osd[n].grad.flatten()
        if r.norm() > 1e-6:
            coss.append((float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-30)), n))
    coss.sort()
    print("worst encoder/head cosines with dropout:", coss[:5])
    assert coss[0][0] >= 0.985 and coss[len(coss) // 2][0] >= 0.997, coss[:5]
    out2 = model(*gb)
    assert int(model._drop_word.item()) == 43 and abs(out2["loss_total"].item() - out["loss_total"].item()) > 1e-5
    model.eval()
    e1, e2 = model(*gb)["loss_total"].item(), model(*gb)["loss_total"].item()
    assert e1 == e2 and int(model._drop_word.item()) == 43
    m = keep_mask(42, 3, 0.1, 200000)
    assert abs(m.mean() - 0.9) < 3e-3


def test_eval_mode_matches_oracle(dev):
    cfg, model, out, osd, ref, keep, stats, gold = _run_pair("lrw_tiny_eval", dev)
    for k in ("loss_total", "loss_category", "loss_audio"):
        assert abs(out[k].item() - ref[k].item()) <= 1e-2 * abs(ref[k].item()), (k, out[k].item(), ref[k].item())


def test_state_dict_roundtrip_and_determinism(dev):
    from syncvsr_amd.model import Model

    cfg, sd, batch, training, gold = build_case("lrw_tiny")
    model = Model(cfg)
    model.load_state_dict(sd)
    model.to(dev).train()
    gb = [t.to(dev) for t in batch]
    l1 = model(*gb)["loss_total"].item()
    sd2 = {k: v.cpu() for k, v in model.state_dict().items()}
    for k, v in sd.items():
        if "running" in k or "num_batches" in k:
            continue
        assert torch.equal(sd2[k], v), k
        assert sd2[k].shape == v.shape
    m2 = Model(cfg)
    m2.load_state_dict(sd)
    m2.to(dev).train()
    l2 = m2(*gb)["loss_total"].item()
    assert abs(l1 - l2) <= 1e-4 * abs(l1)


@pytest.mark.parametrize("B,T,size", [(1, 29, 88), (3, 11, 96), (5, 7, 40), (2, 30, 64)])
def test_lrw_other_shapes(dev, B, T, size):
    """Clip geometries besides the headline 29 x 88 x 88: the 96 x 96 variant of the shipped yaml, a single clip,
    odd batch / frame counts, T = 30 (31 encoder tokens) — losses and feature parity against the oracle."""
    from oracle import lrw_oracle as O
    from syncvsr_amd.config import default_lrw_config
    from syncvsr_amd.init import init_state_dict, synthetic_batch
    from syncvsr_amd.model import Model

    cfg = default_lrw_config(model__bert__num_hidden_layers=2)
    sd = init_state_dict(cfg, seed=21, perturb_norm=True)
    batch = synthetic_batch(cfg, batch=B, frames=T, size=size, seed=33)
    model = Model(cfg)
    model.load_state_dict(sd)
    model.to(dev).train()
    out = model(*[t.to(dev) for t in batch])
    out["loss_total"].backward()
    torch.cuda.synchronize()
    keep = {}
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    with torch.no_grad():
        ref = O.forward(sd, cfg, *batch, training=True, keep=keep)
    # fp32 reference: small batches put as few as 343 values into a BatchNorm channel, so one flipped bf16 rounding moves the statistics
    loss_tol = 1e-3 if size == 96 else 2e-2          # the 96 x 96 crop of the shipped yaml is held to north_star's bound
    for k in ("loss_total", "loss_category", "loss_audio"):
        assert abs(out[k].item() - ref[k].item()) <= loss_tol * abs(ref[k].item()), (k, out[k].item(), ref[k].item())
    a, b = model._last["feats"].float().cpu().flatten(), keep["feats"].flatten()
    assert float((a - b).norm() / b.norm()) <= 4e-2
    # gradients against the bf16-storage emulation of the oracle (same rounding points as the HIP path)
    trunk, other = _emu_grad_stats(cfg, sd, batch, model)
    print(f"B={B} T={T} {size}x{size}: loss hip {out['loss_total'].item():.5f} fp32 {ref['loss_total'].item():.5f} | trunk cos min {trunk[0][0]:.4f} median "
          f"{trunk[len(trunk) // 2][0]:.4f} ratio [{min(r for _, r, _ in trunk):.3f}, {max(r for _, r, _ in trunk):.3f}] | encoder/heads cos min {other[0][0]:.4f} "
          f"ratio [{min(r for _, r, _ in other):.3f}, {max(r for _, r, _ in other):.3f}]")
    assert trunk[0][0] >= GRAD_BOUNDS[size][0] and trunk[len(trunk) // 2][0] >= GRAD_BOUNDS[size][1], (trunk[:3], trunk[len(trunk) // 2])
    assert other[0][0] >= GRAD_BOUNDS[size][2], other[:3]
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


# (trunk min cosine, trunk median cosine, encoder/heads min cosine) against the emulation, by crop size: set from the measured values
# printed above with a margin for run-to-run rounding flips (the deeper check is tests/test_gpu_blockwise.py)
GRAD_BOUNDS = {88: (0.90, 0.95, 0.997), 96: (0.90, 0.95, 0.997), 40: (0.92, 0.955, 0.997), 64: (0.92, 0.955, 0.997)}      # measured: 0.926-0.950 / 0.966-0.973 / 0.9985-0.9987


def test_fused_batchnorm_backward_equals_separate_passes(dev):
    """The data-gradient launches that carry the BatchNorm backward's first pass (ops.BN_BWD_FUSED, the default) against the separate
    reduce + apply passes, whole model at the benchmark batch.  Stage by stage the two agree to 1e-6 (same masked gradient bit for
    bit, per-channel sums added in another order: tests/test_gpu_kernels.py); over the 17 BatchNorm stages of the trunk the per-channel
    offsets compound to ~1 % at the stem (measured 1.6e-2 worst, tensors outside the trunk identical) — against the fp32 oracle both
    variants sit at the same distance (stem cosine 0.92518 / 0.92518, scripts/gpu_bnfuse_vs_oracle.sh), so this is summation noise of
    bf16 training, and the bound below is there to catch a wrong mask or a missing term (which give O(1) differences)."""
    from syncvsr_amd import ops
    from syncvsr_amd.model import Model

    cfg, sd, batch, training, gold = build_case("lrw_full_b32")
    gbatch = [t.to(dev) for t in batch]

    def grads(fused):
        ops.tune("bn_bwd_fused", int(fused))
        try:
            model = Model(cfg)
            model.load_state_dict(sd, strict=True)
            model.to(dev).train(True)
            out = model(*gbatch)
            out["loss_total"].backward()
            torch.cuda.synchronize()
            st = model.store()
            return float(out["loss_total"]), {n: st.g32(n).clone() for n in st.offsets}
        finally:
            ops.tune("bn_bwd_fused", 1)

    la, ga = grads(True)
    lb, gb = grads(False)
    assert la == lb, "the forward does not depend on the switch"
    rel = {}
    for n in ga:
        nb = float(gb[n].norm())
        if nb < 1e-12:
            continue
        rel[n] = float((ga[n] - gb[n]).norm()) / nb
    order = sorted(rel, key=rel.get, reverse=True)
    print("fused vs separate: worst", [(n, f"{rel[n]:.2e}") for n in order[:4]], "median", f"{sorted(rel.values())[len(rel) // 2]:.2e}")
    assert rel[order[0]] <= 4e-2, f"worst relative gradient difference {rel[order[0]]:.3e} ({order[0]})"
    assert all(v == 0.0 for n, v in rel.items() if not n.startswith(("resnet", "stem3d"))), "tensors outside the trunk must be identical"
    assert rel["resnet.layer4.1.conv2.weight"] == 0.0 and rel["resnet.layer4.1.conv1.weight"] <= 5e-4
