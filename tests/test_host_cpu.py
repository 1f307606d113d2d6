"""CPU-only checks (-m "not gpu"): the C-ABI library loads and exports every declared symbol, the host layer mirrors the
reference interface (constructor, state-dict names, parameter groups), the flat parameter store, the stride-2 dgrad tap
planning, and the bucketed gradient reducer over gloo with two processes.  No compute entry point is called here."""
import ctypes
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from syncvsr_amd import _lib, build

    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    decls = _lib.parse_header()
    assert len(decls) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name, args in decls.items():
        assert hasattr(lib, name), f"{name} is declared in include/syncvsr_hip.h but not exported"
        for ctype, argname in args:
            assert "*" in ctype or ctype.replace("const", "").strip() in ("int", "float", "int64_t", "unsigned", "hipStream_t"), (name, ctype)
    handle = _lib.load()
    assert handle.svsr_igemm_fwd.argtypes is not None and len(handle.svsr_igemm_fwd.argtypes) == len(decls["svsr_igemm_fwd"])


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from syncvsr_amd import _lib

    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    _lib.load.cache_clear()
    with pytest.raises(_lib.SvsrError):
        _lib.load()
    _lib.load.cache_clear()


def test_cpu_forward_is_refused():
    from syncvsr_amd.config import default_lrw_config
    from syncvsr_amd.init import synthetic_batch
    from syncvsr_amd.model import Model

    cfg = default_lrw_config(model__bert__num_hidden_layers=1)
    model = Model(cfg)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(*synthetic_batch(cfg, 1, frames=3, size=16))


def test_state_dict_names_match_reference_golden():
    """Parameter names/shapes are exactly the reference module's (recorded in the goldens' grad_names)."""
    from syncvsr_amd.config import default_lrw_config
    from syncvsr_amd.model import Model

    gold = np.load(os.path.join(ROOT, "tests", "golden", "lrw_full_b2.npz"))
    model = Model(default_lrw_config())
    names = [n for n, _ in model.named_parameters()]
    assert sorted(names) == sorted(str(n) for n in gold["grad_names"])
    sd = model.state_dict()
    assert sd["stem3d.0.weight"].shape == (64, 1, 5, 7, 7)
    assert sd["resnet.layer2.0.downsample.0.weight"].shape == (128, 64, 1, 1)
    assert sd["resnet.layer1.0.bn1.num_batches_tracked"].dtype == torch.long
    assert sd["audio_projection.weight"].shape == (2560, 512)
    assert sd["encoder.encoder.layer.5.attention.self.query.weight"].shape == (512, 512)
    groups = model.configure_optimizers()
    assert all(p.ndim >= 2 for p in groups[0]["params"]) and all(p.ndim < 2 for p in groups[1]["params"])
    assert groups[1]["weight_decay"] == 0.0
    assert any(p is model.cls_token for p in groups[0]["params"])      # 3-D cls_token is decayed, as in the reference


def test_unsupported_configs_raise():
    from syncvsr_amd.config import default_lrw_config
    from syncvsr_amd.model import Model

    with pytest.raises(NotImplementedError):
        Model(default_lrw_config(model__bert__type="x-transformers"))
    with pytest.raises(NotImplementedError):
        Model(default_lrw_config(data__use_word_boundary=True))
    # dropout is supported; keys a reference-style config omits take the reference's defaults (BertConfig: 0.1 / 0.1;
    # emb_dropout is read directly and must be present, lightning.py:45,92)
    cfg = default_lrw_config()
    del cfg.model.bert["hidden_dropout_prob"], cfg.model.bert["attention_probs_dropout_prob"]
    m = Model(cfg)
    assert (m.drop_p, m.attn_drop_p, m.emb_drop_p) == (0.1, 0.1, 0.0)
    m = Model(default_lrw_config(model__bert__hidden_dropout_prob=0.2, model__bert__emb_dropout=0.3))
    assert (m.drop_p, m.attn_drop_p, m.emb_drop_p) == (0.2, 0.0, 0.3)


def test_param_store_layout_on_cpu():
    from syncvsr_amd.config import default_lrw_config
    from syncvsr_amd.init import init_state_dict
    from syncvsr_amd.model import Model, _ParamStore

    cfg = default_lrw_config(model__bert__num_hidden_layers=2)
    model = Model(cfg)
    sd = init_state_dict(cfg, seed=7, perturb_norm=True)
    model.load_state_dict(sd)
    st = _ParamStore(model, torch.device("cpu"))
    assert st.owns(model)
    names = list(st.offsets)
    # decayed tensors first, in forward order, then the 1-D tail
    first_1d = next(i for i, n in enumerate(names) if len(st.offsets[n][2]) < 2)
    assert all(len(st.offsets[n][2]) >= 2 for n in names[:first_1d]) and all(len(st.offsets[n][2]) < 2 for n in names[first_1d:])
    assert names[0] == "stem3d.0.weight" and names[first_1d - 1] == "category_classifier.weight"
    assert st.offsets[names[first_1d - 1]][0] + st.offsets[names[first_1d - 1]][1] <= st.decay_end <= st.offsets[names[first_1d]][0]
    for n, p in model.named_parameters():
        assert torch.equal(p.detach(), sd[n]), n                      # values survive the move into the flat buffer
        assert p.grad is not None and p.grad.shape == p.shape
        assert st.offsets[n][0] % 4 == 0
    o, numel, shape = st.offsets["resnet.layer3.0.conv1.weight"]
    assert torch.equal(st.flat[o : o + numel].view(256, 3, 3, 128), sd["resnet.layer3.0.conv1.weight"].permute(0, 2, 3, 1))
    # q/k/v of a layer are adjacent so one [3D, D] GEMM covers them
    q = st.offsets["encoder.encoder.layer.1.attention.self.query.weight"][0]
    assert st.offsets["encoder.encoder.layer.1.attention.self.value.weight"][0] == q + 2 * 512 * 512
    p = model.audio_projection.weight
    p.grad.fill_(3.0)
    assert float(st.g32("audio_projection.weight").sum()) == 3.0 * p.numel()


def test_xtransformers_word_boundary_store_is_padded_and_logical_on_cpu():
    """`type: x-transformers` + use_word_boundary (the shipped WB yaml): 513-wide tensors live in 576-wide storage whose pads are
    zero; parameters / state dict keep the reference modules' logical shapes; to_q/to_k/to_v are adjacent for the fused GEMM."""
    from syncvsr_amd.config import xtransformers_lrw_config
    from syncvsr_amd.init import init_state_dict, param_specs
    from syncvsr_amd.model import Model, _ParamStore

    cfg = xtransformers_lrw_config(True, model__bert__depth=2)
    model = Model(cfg)
    assert (model.dim, model.dim_p, model.inter, model.inter_p, model.glu_p) == (513, 576, 2052, 2112, 4160)
    assert (model.drop_p, model.attn_drop_p, model.emb_drop_p, model.layer_drop_p) == (0.3, 0.0, 0.0, 0.2)
    sd = init_state_dict(cfg, seed=7, perturb_norm=True)
    assert sd["cls_token"][0, 0, -1] == 0.0                            # lightning.py:109-110
    model.load_state_dict(sd, strict=True)
    st = _ParamStore(model, torch.device("cpu"))
    assert st.owns(model)
    for n, p in model.named_parameters():
        assert torch.equal(p.detach(), sd[n]), n
        assert p.grad is not None and p.grad.shape == p.shape == sd[n].shape
    q = st.offsets["encoder.layers.2.1.to_q.weight"][0]
    assert st.offsets["encoder.layers.2.1.to_k.weight"][0] == q + 512 * 576 and st.offsets["encoder.layers.2.1.to_v.weight"][0] == q + 2 * 512 * 576
    o, numel, shape = st.offsets["encoder.layers.1.1.ff.3.weight"]
    assert (numel, shape, st.phys["encoder.layers.1.1.ff.3.weight"]) == (576 * 2112, (513, 2052), (576, 2112))
    full = st.flat[o:o + numel].view(576, 2112)
    assert torch.equal(full[:513, :2052], sd["encoder.layers.1.1.ff.3.weight"]) and not full[513:].any() and not full[:, 2052:].any()
    assert st.t_offsets["encoder.layers.0.1.qkv"][1] == (576, 1, 1536)
    assert st.t_offsets["encoder.layers.1.1.ff.0.proj.weight"][1] == (576, 1, 4160)
    assert st.t_offsets["category_classifier.weight"][1] == (576, 1, 512)
    # the no-word-boundary variant needs no padding at all
    cfg2 = xtransformers_lrw_config(False, model__bert__depth=1)
    m2 = Model(cfg2)
    assert (m2.dim, m2.dim_p, m2.inter_p, m2.glu_p) == (512, 512, 2048, 4096)
    assert all(tuple(m2._phys[n]) == tuple(shp) for n, shp, _ in param_specs(cfg2))
    with pytest.raises(NotImplementedError):
        Model(xtransformers_lrw_config(True, model__bert__ff_glu=False))


def test_stride2_dgrad_tap_plan():
    """Each input-parity class of a stride-2 3x3 transposed conv uses only its own taps (1, 2, 2, 4 of the 9)."""
    k, stride, pad = 3, 2, 1
    counts = {}
    for py in range(2):
        for px in range(2):
            taps = [(kh, kw) for kh in range(k) for kw in range(k) if (py + pad - kh) % stride == 0 and (px + pad - kw) % stride == 0]
            counts[(py, px)] = len(taps)
    assert counts == {(0, 0): 1, (0, 1): 2, (1, 0): 2, (1, 1): 4}


_WORKER = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, {root!r})
    from syncvsr_amd.config import default_lrw_config
    from syncvsr_amd.model import Model, _ParamStore
    from syncvsr_amd.engine import GradReducer
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[3]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = default_lrw_config(model__bert__num_hidden_layers=1)
    model = Model(cfg)
    model._store = _ParamStore(model, torch.device("cpu"))
    st = model._store
    # DDP construction semantics: whatever a rank holds, it starts from rank 0's parameters and BatchNorm statistics
    st.flat.add_(float(rank)); st.bufflat.add_(float(rank))
    mine = st.flat.clone()
    red = GradReducer(model, None, bucket_mb=8.0)
    both = [torch.empty_like(st.flat) for _ in range(world)]
    dist.all_gather(both, st.flat)
    assert all(torch.equal(b, both[0]) for b in both) and (rank != 0 or torch.equal(st.flat, mine))
    bufs = [torch.empty_like(st.bufflat) for _ in range(world)]
    dist.all_gather(bufs, st.bufflat)
    assert all(torch.equal(b, bufs[0]) for b in bufs)
    # sync_dist=True logging: rank-mean of a step's scalars in one collective (dict for LRW, tuple for LRS)
    from syncvsr_amd.engine import reduce_metrics
    m = reduce_metrics({{"loss_total": torch.tensor(1.0 + rank), "accuracy_top1": torch.tensor(0.5 * rank)}})
    assert abs(float(m["loss_total"]) - (1.0 + (world - 1) / 2)) < 1e-6 and abs(float(m["accuracy_top1"]) - 0.25 * (world - 1)) < 1e-6
    t = reduce_metrics((torch.tensor(2.0 * rank), torch.tensor(3.0)))
    assert isinstance(t, tuple) and abs(float(t[0]) - (world - 1)) < 1e-6 and float(t[1]) == 3.0
    red.begin_step()
    st.grad.copy_(torch.arange(st.numel, dtype=torch.float32) % 97 + rank)       # rank-dependent pattern
    # replay the order in which the hand-written backward reports progress
    order = ["audio_projection.weight"] + [f"encoder.encoder.layer.{{i}}.attention.self.query.weight" for i in reversed(range(1))] + ["cls_token"]
    from syncvsr_amd.init import resnet_block_specs
    order += [f"{{p}}.conv1.weight" for p, *_ in reversed(list(resnet_block_specs()))]
    for name in order:
        red.on_ready(st.offsets[name][0])
    red.on_ready(0)
    red.finish()
    expect = torch.arange(st.numel, dtype=torch.float32) % 97 + (world - 1) / 2.0
    assert torch.allclose(st.grad, expect), (st.grad - expect).abs().max()
    covered = sorted(red.launched)
    pos = 0
    for lo, hi in covered:                      # buckets tile [0, numel) exactly once
        assert lo == pos, (lo, pos)
        pos = hi
    assert pos == st.numel and len(covered) >= 3
    assert covered[0] == (0, min(1 << 20, covered[0][1])), "the bucket reduced after the backward holds at most 4 MB"
    dist.barrier(); dist.destroy_process_group()
    print("ok", rank)
""")


def test_grad_reducer_two_process_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


# ---------------------------------------------------------------------------------------------------------
# LRS host logic (no GPU): state-dict names, target preparation, dropout twin, unsupported configs, layout
# ---------------------------------------------------------------------------------------------------------
def test_lrs_host_logic():
    import numpy as np
    import torch

    from golden_cases import build_lrs_case
    from oracle import lrs_oracle as O
    from syncvsr_amd.dropout import keep_mask, lrs_sites
    from syncvsr_amd.lrs_init import default_lrs_args, lrs_param_specs
    from syncvsr_amd.lrs_model import E2E

    args, odim, sd, batch, training, gold = build_lrs_case("lrs_tiny_b3")
    m = E2E(odim, args)
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict(sd, strict=True)
    # decay split follows the reference (p.ndim >= 2, LRS/video/lightning.py:89-96)
    groups = m.configure_optimizers()
    assert all(p.ndim >= 2 for p in groups[0]["params"]) and all(p.ndim < 2 for p in groups[1]["params"])
    assert groups[1]["weight_decay"] == 0.0
    # targets: same tensors as the reference's add_sos_eos (restated in the oracle)
    label = batch[3]
    tg = m.prepare_targets(label)
    ys = [y[y != -1] for y in label.view(label.size(0), -1)]
    ys_in, ys_out = O.add_sos_eos(ys, odim - 1, odim - 1)
    assert torch.equal(tg.ys_in, ys_in) and torch.equal(tg.ys_out, ys_out)
    assert torch.equal(tg.labels, label.view(label.size(0), -1))
    # CPU tensors are refused (no fallback); unsupported configurations raise
    with pytest.raises(RuntimeError):
        m(*batch)
    for bad in (dict(macaron_style=False), dict(adim=128, aheads=2, ddim=256, dheads=2), dict(cnn_module_kernel=33), dict(mtlalpha=1.0),
                dict(transformer_input_layer="conv1d")):
        with pytest.raises(NotImplementedError):
            E2E(odim, default_lrs_args(**bad))
    # the shipped config: 250.38 M parameters + the 1.97 M audio_classifier
    n = sum(int(np.prod(s)) for _, s, _ in lrs_param_specs(default_lrs_args(), 5049))
    assert abs(n - 252.35e6) < 0.05e6, n
    # dropout twin: keep rate, determinism, site separation
    sites = lrs_sites(12, 6)
    assert len(set(sites.values())) == len(sites) == 4 + 12 * 7 + 6 * 6
    a, b = keep_mask(5, sites["enc.0.ff.out"], 0.1, 100000), keep_mask(5, sites["enc.0.ff.out"], 0.1, 100000)
    c = keep_mask(5, sites["enc.1.ff.out"], 0.1, 100000)
    assert (a == b).all() and abs(a.mean() - 0.9) < 5e-3 and abs((a == c).mean() - 0.82) < 1e-2


def test_plan_cache_is_bounded_for_variable_shapes(monkeypatch):
    """ops._PLAN_CACHE (per-shape launch plans / shape queries): variable-length batches meet new shapes for ever — the cache starts over
    at PLAN_CACHE_MAX entries, except once a HIP graph has captured launches that reference the plans' device words."""
    from syncvsr_amd import ops

    cache = ops._BoundedCache()
    monkeypatch.setattr(ops, "PLAN_CACHE_MAX", 5)
    monkeypatch.setattr(ops, "PLAN_CACHE_PINNED", False)
    for i in range(5):
        cache[i] = i
    assert len(cache) == 5
    cache[5] = 5                      # the sixth shape: start over
    assert list(cache) == [5]
    monkeypatch.setattr(ops, "PLAN_CACHE_PINNED", True)
    for i in range(10, 20):
        cache[i] = i
    assert len(cache) == 11           # pinned: nothing is dropped


def test_header_parser_handles_every_declaration_form():
    """_lib.parse_header: int and int64_t return types, (void) parameter lists, pointer / scalar argument kinds."""
    from syncvsr_amd import _lib

    h = _lib.parse_header()
    assert "svsr_stem_conv_fwd_ws_bytes" in h and _lib._RESTYPE["svsr_stem_conv_fwd_ws_bytes"] == "int64_t"
    assert _lib._RESTYPE["svsr_igemm_fwd"] == "int"
    assert [t for t, _ in h["svsr_tune"]] == ["const char*", "int"]
    assert all(name.startswith("svsr_") for name in h) and len(h) >= 69


def test_step_list_registry_covers_every_stream_entry_point():
    """csrc/steplist.hip re-issues recorded launches through a table of entry points: every function of include/syncvsr_hip.h that
    takes a hipStream_t (except the step-list / stream plumbing itself) must be in it, or a recorded step would refuse the launch."""
    from syncvsr_amd import _lib

    lib = _lib.load()
    hdr = _lib.parse_header()
    plumbing = {"svsr_stream_wait", "svsr_memset_async", "svsr_debug_occupy_start", "svsr_clock_probe"}      # (stream management / test aids, not launches of a step)
    launches = [n for n, a in hdr.items() if a and a[-1][0] == "hipStream_t" and not n.startswith("svsr_steplist") and n not in plumbing]
    assert len(launches) >= 50
    missing = [n for n in launches if not lib.svsr_steplist_knows(n.encode())]
    assert not missing, missing
    assert not lib.svsr_steplist_knows(b"svsr_conv_plan")            # host-side query, not a launch


def test_step_list_segments_and_argument_checks():
    """Host logic of the step list without a GPU: segment bookkeeping, arity check, unknown names."""
    import ctypes

    from syncvsr_amd import _lib

    lib = _lib.load()
    h = lib.svsr_steplist_create()
    try:
        assert lib.svsr_steplist_segments(h) == 1 and lib.svsr_steplist_size(h) == 0
        slots = (ctypes.c_int64 * 3)(0, 1, 0)
        assert lib.svsr_steplist_push_call(h, b"svsr_word_add", slots, 3) == 0
        assert lib.svsr_steplist_push_call(h, b"svsr_word_add", slots, 2) == 1001          # wrong arity
        assert lib.svsr_steplist_push_call(h, b"svsr_no_such_entry", slots, 3) == 1001
        assert lib.svsr_steplist_push_break(h) == 1
        assert lib.svsr_steplist_segments(h) == 2 and lib.svsr_steplist_size(h) == 1
        failed = ctypes.c_int(-1)
        assert lib.svsr_steplist_run(h, 5, ctypes.byref(failed)) == 1001                   # no such segment
        assert lib.svsr_steplist_run(h, 1, ctypes.byref(failed)) == 0                      # empty segment: nothing to issue
    finally:
        assert lib.svsr_steplist_destroy(h) == 0


def test_c64_pixel_table_matches_the_padded_grid():
    """svsr_conv3x3_c64_pixtab: entry PAD + q = pixel of padded coordinate q ((H+2) x (W+2) grid per image) or -1 (host function, no GPU)."""
    import ctypes

    from syncvsr_amd import _lib

    lib = _lib.load()
    N, H, W = 3, 4, 5
    n = int(lib.svsr_conv3x3_c64_pixtab(N, H, W, None, 0))
    buf = (ctypes.c_int * n)()
    assert int(lib.svsr_conv3x3_c64_pixtab(N, H, W, buf, n)) == n
    assert int(lib.svsr_conv3x3_c64_pixtab(N, H, W, buf, n - 1)) < 0
    tab = list(buf)
    pad = 64
    Q = (H + 2) * (W + 2)
    assert all(v == -1 for v in tab[:pad]) and n >= pad + N * Q
    seen = []
    for q in range(N * Q):
        img, rem = divmod(q, Q)
        yp, xp = divmod(rem, W + 2)
        want = (img * H + yp - 1) * W + xp - 1 if 1 <= yp <= H and 1 <= xp <= W else -1
        assert tab[pad + q] == want
        if want >= 0:
            seen.append(want)
    assert seen == list(range(N * H * W)) and all(v == -1 for v in tab[pad + N * Q:])


def test_deferred_reductions_are_batched_without_merging_contributions_to_one_gradient(monkeypatch):
    """ops.run_deferred (host logic, no device): neighbouring postponed column sums go out as svsr_colsum_rows_multi records — but two
    contributions to the same gradient never share a launch (a launch's records must not overlap), a closure that is not a column sum runs
    in its place, and a single leftover falls back to svsr_colsum_rows."""
    from syncvsr_amd import ops

    calls = []
    monkeypatch.setattr(ops, "_call", lambda name, *a, **k: calls.append((name, a)))
    monkeypatch.setattr(ops, "_stream", lambda: 0)
    mk = lambda n: torch.zeros(n)
    outs = [mk(8) for _ in range(6)]
    parts = [mk(64) for _ in range(8)]
    other = []
    fns = [ops._deferred_colsum(parts[0], 8, 8, outs[0], 8),
           ops._deferred_colsum(parts[1], 8, 8, outs[1], 4, outs[2], 4),
           ops._deferred_colsum(parts[2], 8, 8, outs[0], 8),                 # second contribution to outs[0]: a new launch
           ops._deferred_colsum(parts[3], 8, 8, outs[3], 8),
           (lambda: other.append("ran")),                                    # not a column sum: flushes, then runs
           ops._deferred_colsum(parts[4], 8, 8, outs[4], 8)]
    ops.run_deferred(fns)
    names = [c[0] for c in calls]
    assert names == ["svsr_colsum_rows_multi", "svsr_colsum_rows_multi", "svsr_colsum_rows"], names
    assert calls[0][1][1] == 2 and calls[1][1][1] == 2 and other == ["ran"]
    rec = calls[0][1][0]
    assert ctypes.sizeof(rec[0]) == 64 and rec[1].n0 == 4 and rec[1].n1 == 4 and rec[1].out1 == outs[2].data_ptr() and rec[0].accumulate == 1
    # more than 16 records still go out in one call (the library cuts it into launches of 16)
    calls.clear()
    ops.run_deferred([ops._deferred_colsum(mk(64), 8, 8, mk(8), 8) for _ in range(19)])
    assert [c[0] for c in calls] == ["svsr_colsum_rows_multi"] and calls[0][1][1] == 19
    # the knob off: every closure by itself
    calls.clear()
    monkeypatch.setattr(ops, "COLSUM_MULTI", False)
    ops.run_deferred(fns[:2])
    assert [c[0] for c in calls] == ["svsr_colsum_rows", "svsr_colsum_rows"]


def test_reduction_splits_are_planned_for_a_fixed_compute_unit_count():
    """The order in which the partial sums of a weight gradient (unit lists of svsr_igemm_wgrad) and of the layer1 kernel's BatchNorm statistics are
    added must not depend on the device the run happens to sit on (a checkpoint resumed on a partition of another size continues bit for bit):
    the splits are planned for the knob `reduce_cus` = 256, not for the device's compute-unit count, and the knobs that decide the order are
    readable (TrainStep.state_dict() records them)."""
    from syncvsr_amd import ops

    assert ops.REDUCTION_KNOBS[0] == "reduce_cus" and ops.tune_value("reduce_cus") == 256
    before = ops.reduction_plan_params()
    assert len(before) == len(ops.REDUCTION_KNOBS) and before[0] == 256
    try:
        ops.tune("reduce_cus", 128)
        assert ops.tune_value("reduce_cus") == 128 and ops.reduction_plan_params() != before
    finally:
        ops.tune("reduce_cus", 256)
    assert ops.reduction_plan_params() == before
    with pytest.raises(Exception):
        ops.tune_value("no_such_knob")
