"""Training steps beside a FOREIGN RESIDENT KERNEL (-m gpu): what one GPU can show about the data-parallel configurations
(BASELINE configs[2] / [4], reference strategy="ddp": LRW/video/src/train.py:28, LRS/video/main.py:38).  On 8 GPUs a peer-waiting
collective kernel sits on some compute units — holding LDS — while the step's persistent kernels run.  The stand-in here is
svsr_debug_occupy_start: 32 workgroups x 96 KiB of LDS on a third stream that stay resident until the host releases them.

  * Every kernel must only get slower: same losses, bit for bit (the reductions are fixed-order and the grids are sized by the device, not
    by what happens to be free).  That includes the fused encoder (csrc/enc_fused.hip), whose 32 clusters of 8 workgroups spin on each
    other: workgroups are dispatched in order, a cluster is 8 consecutive workgroup ids, so with 32 compute units taken the launch runs
    as whole clusters in two rounds and no bounded wait gives up (measured here).
  * If a cluster wait DOES give up (provoked with svsr_debug_enc_spin_limit(1)), that must not end the run: the poisoned step is skipped
    (svsr_adamw_step's non-finite guard), TrainStep re-routes the encoder to the per-layer launch chain with a warning
    (engine.TrainStep._watch_fused_encoder) and training continues.
"""
import time
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu


def _occupy(stream, workgroups=32, lds=96 * 1024):
    from syncvsr_amd import _lib

    _lib.check(_lib.load().svsr_debug_occupy_start(workgroups, lds, stream.cuda_stream), "svsr_debug_occupy_start")


def _release():
    from syncvsr_amd import _lib

    _lib.load().svsr_debug_occupy_stop()          # (an error code only says that nothing was started)


def _lrw(batch_size=32):
    from syncvsr_amd.config import default_lrw_config
    from syncvsr_amd.init import synthetic_batch
    from syncvsr_amd.model import Model

    dev = torch.device("cuda:0")
    cfg = default_lrw_config()
    cfg.train.batch_size = batch_size
    cfg.optim.scheduler.num_warmup_steps = 1
    model = Model(cfg, seed=0).to(dev).train()
    model.reseed_dropout(321)
    batch = [t.to(dev) for t in synthetic_batch(cfg, batch_size, seed=77)]
    return cfg, model, batch


def test_lrw_step_beside_a_cotenant_gives_the_same_losses():
    """Benchmark batch (the fused encoder launches 256 workgroups of 160 KiB LDS) beside 32 occupied compute units: bit-identical losses, no
    cluster wait gives up, no step skipped."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from syncvsr_amd import ops
    from syncvsr_amd.engine import TrainStep

    if not (ops.ENC_FUSED and ops.ENC_BWD_FUSED):
        pytest.skip("the fused encoder is switched off in this process")
    side = torch.cuda.Stream()

    def run(beside: bool):
        cfg, model, batch = _lrw()
        ts = TrainStep(model, cfg, native=True)
        ts.step(*batch)                       # recorded alone
        torch.cuda.synchronize()
        if beside:
            _occupy(side)
            time.sleep(0.05)                  # (resident before the steps are enqueued)
        t0 = time.perf_counter()
        kept = [ts.step(*batch)["loss_total"].clone() for _ in range(4)]
        torch.cuda.current_stream().synchronize()      # (a device-wide synchronisation or a read-back would wait for the co-tenant itself)
        ts.model._side.stream.synchronize()
        dt = time.perf_counter() - t0
        if beside:
            _release()
            torch.cuda.synchronize()
        losses = [float(t.item()) for t in kept]
        return losses, dt, ts.state(), ts.fused_encoder_fallbacks

    try:
        alone, t_alone, st0, _ = run(False)
        shared, t_shared, st1, fb = run(True)
    finally:
        _release()
        torch.cuda.synchronize()
    assert fb == 0 and st1["skipped_steps"] == 0 and not ops.check_enc_clusters(reset=True), (fb, st1)
    assert alone == shared, (alone, shared)
    assert st0["step"] == st1["step"] == 5
    print(f"LRW 4 steps alone {t_alone * 1e3:.1f} ms, beside the co-tenant {t_shared * 1e3:.1f} ms")


def test_a_cluster_wait_that_gives_up_falls_back_to_the_launch_chain():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from syncvsr_amd import _lib, ops
    from syncvsr_amd.engine import TrainStep

    keep = (ops.ENC_FUSED, ops.ENC_BWD_FUSED)
    if not (ops.ENC_FUSED and ops.ENC_BWD_FUSED):
        pytest.skip("the fused encoder is switched off in this process")
    lib = _lib.load()
    cfg, model, batch = _lrw()
    try:
        ts = TrainStep(model, cfg, native=True)
        clean = [float(ts.step(*batch)["loss_total"].item()) for _ in range(3)]
        assert all(x == x for x in clean) and ts.fused_encoder_fallbacks == 0
        p_before = model.store().flat.clone()
        torch.cuda.synchronize()
        lib.svsr_debug_enc_spin_limit(1)      # every cluster wait that has to poll twice gives up: the launches poison their outputs
        ts._rec = None                        # (the spin limit travels in the launch arguments: record the step again)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            bad = [float(ts.step(*batch)["loss_total"].item()) for _ in range(2)]
            lib.svsr_debug_enc_spin_limit(0)
            good = [float(ts.step(*batch)["loss_total"].item()) for _ in range(3)]     # the watch re-routes the encoder before the first of these
            torch.cuda.synchronize()
        st = ts.state()
        assert any(x != x for x in bad), f"the give-up path was not provoked: {bad}"
        assert ts.fused_encoder_fallbacks == 1, (ts.fused_encoder_fallbacks, bad, good)
        assert any("fused encoder disabled" in str(w.message) for w in caught), [str(w.message) for w in caught]
        assert not ops.ENC_FUSED and not ops.ENC_BWD_FUSED
        assert st["skipped_steps"] >= 1 and st["step"] == 3 + 2 + 3 - st["skipped_steps"], st
        assert all(x == x for x in good), f"the steps after the fall-back still give NaN: {good}"
        flat = model.store().flat
        assert bool(torch.isfinite(flat).all()), "a poisoned step reached the parameters"
        assert not torch.equal(flat, p_before), "no step was applied after the fall-back"
        assert good[-1] < clean[0], (clean, bad, good)
        print(f"forced give-up: clean {clean}, poisoned {bad}, after the fall-back {good}, skipped {st['skipped_steps']}")
    finally:
        lib.svsr_debug_enc_spin_limit(0)
        torch.cuda.synchronize()
        ops.check_enc_clusters(reset=True)
        ops.ENC_FUSED, ops.ENC_BWD_FUSED = keep


def test_lrs_step_beside_a_cotenant_gives_the_same_losses():
    """No kernel of the sentence-level step waits for another workgroup: beside the co-tenant it is slower, and bit-identical."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from syncvsr_amd.engine import TrainStep, lrs_train_config
    from syncvsr_amd.lrs_init import LRS_ODIM, default_lrs_args, lrs_synthetic_batch
    from syncvsr_amd.lrs_model import E2E

    dev = torch.device("cuda:0")
    args = default_lrs_args(dropout_rate=0.1, transformer_attn_dropout_rate=0.1)
    cpu_batch = lrs_synthetic_batch(args, 4, 64, seed=5)
    side = torch.cuda.Stream()

    def run(beside: bool):
        model = E2E(LRS_ODIM, args, seed=0).to(dev).train()
        model.reseed_dropout(123)
        ts = TrainStep(model, lrs_train_config(scheduler__num_warmup_steps=1), native=True)
        batch = [t.to(dev) for t in cpu_batch]
        ts.step(*batch)                       # recorded alone
        torch.cuda.synchronize()
        if beside:
            _occupy(side)
            time.sleep(0.05)
        t0 = time.perf_counter()
        kept = [ts.step(*batch)[0].clone() for _ in range(3)]
        torch.cuda.current_stream().synchronize()      # (a device-wide synchronisation or a read-back would wait for the co-tenant itself)
        ts.model._side.stream.synchronize()
        dt = time.perf_counter() - t0
        if beside:
            _release()
            torch.cuda.synchronize()
        losses = [float(t.item()) for t in kept]
        return losses, dt, ts.state()

    try:
        alone, t_alone, st0 = run(False)
        shared, t_shared, st1 = run(True)
    finally:
        _release()
        torch.cuda.synchronize()
    assert alone == shared, (alone, shared)
    assert st0["step"] == st1["step"] == 4 and st1["skipped_steps"] == 0
    print(f"LRS 3 steps alone {t_alone * 1e3:.1f} ms, beside the co-tenant {t_shared * 1e3:.1f} ms")
