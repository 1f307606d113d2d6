"""Thin host-side wrappers over the C ABI (include/syncvsr_hip.h): tensors in, launches out.

PyTorch is used here only as the owner of device memory and streams; every computation is a call into
libsyncvsr_hip.so on the current HIP stream.  Tensors are NHWC bf16 activations unless stated.
"""
from __future__ import annotations

import ctypes
import os
import struct
from contextlib import contextmanager
from typing import Callable, Optional, Sequence

import torch

from . import _lib

BF16 = torch.bfloat16
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
HALO_WGRAD = True      # nine-taps-per-pass weight gradient for stride-1 3x3 convs (False: generic per-tap kernel)


_DEV_INDEX: Optional[int] = None
STREAM_OVERRIDE: Optional[int] = None      # raw hipStream_t that launches go to instead of torch's current stream (model._SideStream)


def _stream() -> int:
    """Raw hipStream_t of torch's current stream.  `torch.cuda.current_stream().cuda_stream` builds a Stream object per call
    (~9 us, a fifth of the host time of an eager step); the raw getter is ~0.3 us.  One device per process."""
    global _DEV_INDEX
    if STREAM_OVERRIDE is not None:
        return STREAM_OVERRIDE
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(_DEV_INDEX)


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _drop(drop) -> tuple:
    """drop = None | (seed tensor [1] int32/uint32 on the device, site id, p) -> the three C arguments."""
    if drop is None or drop[2] <= 0.0:
        return None, 0, 0.0
    return drop[0].data_ptr(), int(drop[1]), float(drop[2])


# Per-stream fp32 scratch for the partial rows / split-K slabs of the reproducible reductions (include/syncvsr_hip.h): every
# producer and its fixed-order finaliser are enqueued back to back on one stream, so one buffer per stream serves them all.
# Outgrown buffers are kept alive: a captured HIP graph may still hold their address.
_SCRATCH: dict[int, torch.Tensor] = {}
_SCRATCH_KEEP: list[torch.Tensor] = []


def scratch(nfloats: int) -> torch.Tensor:
    key = _stream()
    t = _SCRATCH.get(key)
    if t is None or t.numel() < nfloats:
        if t is not None:
            _SCRATCH_KEEP.append(t)
        t = _SCRATCH[key] = torch.empty(max(int(nfloats), 1 << 22), dtype=torch.float32, device="cuda")
    return t


class _BoundedCache(dict):
    """Per-shape launch plans and shape queries.  Variable-length batches (LRS without the length buckets) meet a new shape almost every
    step: beyond PLAN_CACHE_MAX entries the cache starts over, so host and device memory stay bounded (a plan costs ~50 us to rebuild)."""

    def __setitem__(self, key, value):
        if len(self) >= PLAN_CACHE_MAX and not PLAN_CACHE_PINNED:
            self.retire()
        super().__setitem__(key, value)

    def retire(self) -> None:
        """Starts over.  The evicted plans stay alive for one more generation (or for ever once a graph / step list holds their
        device words): a weight-gradient launch on the side stream may still be reading them, and the caching allocator would hand
        their memory to the next allocation on the main stream."""
        global _PLAN_RETIRED
        old = list(self.values())
        _PLAN_RETIRED = (_PLAN_RETIRED + old) if PLAN_CACHE_PINNED else old
        self.clear()


PLAN_CACHE_MAX = 20000
PLAN_CACHE_PINNED = False       # set once a HIP graph has captured launches: their plans' device words must stay alive
_PLAN_RETIRED: list = []
_PLAN_CACHE: _BoundedCache = _BoundedCache()


def _query(name: str, *args) -> tuple:
    """Cached host-side shape query (svsr_*_plan / svsr_*_rows): returns the int outputs (or the returned row count)."""
    key = (name, args)
    hit = _PLAN_CACHE.get(key)
    if hit is not None:
        return hit
    fn = getattr(_lib.load(), name)
    if name.endswith(("_rows", "_bytes", "_floats")):
        out = (int(fn(*args)),)
    else:
        nout = {"svsr_conv3x3_wgrad_plan": (1, 1),
                "svsr_stem_conv_wgrad_plan": (1, 1)}[name]
        ints = [ctypes.c_int(0) for _ in range(nout[0])]
        longs = [ctypes.c_int64(0) for _ in range(nout[1])]
        rc = fn(*args, *[ctypes.byref(v) for v in ints], *[ctypes.byref(v) for v in longs])
        if rc != 0:
            _lib.check(rc, name)
        out = tuple(int(v.value) for v in ints) + tuple(int(v.value) for v in longs)
    _PLAN_CACHE[key] = out
    return out


REDUCTION_KNOBS = ("reduce_cus", "wg_blocks", "wg_units", "wg_unit_max", "wg_unit_min", "wg_short_k", "w3_blocks", "w3_waves", "w3_dense")


def tune_value(key: str) -> int:
    """Current value of a library tuning knob (svsr_tune_value)."""
    out = ctypes.c_int(0)
    _lib.check(_lib.load().svsr_tune_value(key.encode(), ctypes.byref(out)), f"svsr_tune_value({key})")
    return int(out.value)


def reduction_plan_params() -> list[int]:
    """The knobs that decide the ORDER in which partial sums of a weight gradient / a BatchNorm statistic are added (REDUCTION_KNOBS): with the
    same values — whatever the device's compute-unit count — two runs produce the same bits.  Stored in TrainStep.state_dict()."""
    return [tune_value(k) for k in REDUCTION_KNOBS]


def tune(key: str, value: int) -> None:
    """Result-preserving tuning knob of the library (svsr_tune); invalidates cached plans.  `bn_bwd_fused` is a host-side choice
    (which launches the trunk backward issues), kept in this module."""
    if key == "transpose_from_bf16":
        global TRANSPOSE_FROM_BF16
        TRANSPOSE_FROM_BF16 = bool(value)
        return
    if key == "bn_bwd_fused":
        global BN_BWD_FUSED
        BN_BWD_FUSED = bool(value)
        return
    if key == "stem_keep_winners":
        global STEM_KEEP_WINNERS
        STEM_KEEP_WINNERS = bool(value)
        return
    if key == "ln_branch_fused":
        global LN_BRANCH_FUSED
        LN_BRANCH_FUSED = bool(value)
        return
    if key == "colsum_multi":
        global COLSUM_MULTI
        COLSUM_MULTI = bool(value)
        return
    if key == "early_sumsq":
        global EARLY_SUMSQ
        EARLY_SUMSQ = bool(value)
        return
    if key == "stem_bwd_fused":
        global STEM_BWD_FUSED
        STEM_BWD_FUSED = bool(value)
        return
    if key == "ctc_side":
        global CTC_SIDE
        CTC_SIDE = bool(value)
        return
    rc = _lib.load().svsr_tune(key.encode(), int(value))
    if rc != 0:
        _lib.check(rc, f"svsr_tune({key})")
    _PLAN_CACHE.retire()          # plans are rebuilt with the new knob; the old ones stay alive for whoever still points at them


_INT_ARRAYS: dict = {}


def _ints(v: Sequence[int]):
    """ctypes int array for a tap list; cached (the same few tap tables recur every step and building one costs ~1 us)."""
    key = tuple(v)
    arr = _INT_ARRAYS.get(key)
    if arr is None:
        arr = _INT_ARRAYS[key] = (ctypes.c_int * len(key))(*key)
    return arr


# Optional per-launch timing with HIP events (bench.py's roofline leg): {kernel label: [(start, end, flops, bytes)]}.
_TIMING: Optional[dict[str, list]] = None


def start_event_timing() -> None:
    global _TIMING
    _TIMING = {}


def stop_event_timing() -> dict[str, dict[str, float]]:
    """-> {label: {launches, ms (sum), flops (sum), bytes (sum)}}; synchronises the device."""
    global _TIMING
    rec, _TIMING = _TIMING or {}, None
    torch.cuda.synchronize()
    out = {}
    for label, evs in rec.items():
        out[label] = dict(launches=len(evs), ms=sum(s.elapsed_time(e) for s, e, _, _ in evs),
                          flops=float(sum(f for _, _, f, _ in evs)), bytes=float(sum(b for _, _, _, b in evs)))
    return out


# --------------------------------------------------------------------------------------------------
# native step enqueuer (csrc/steplist.hip): every launch of one training step is recorded once and re-issued by ONE library
# call per step (engine.TrainStep(native=True))
# --------------------------------------------------------------------------------------------------
class StepRecorder:
    """While active (`with ops.recording(rec)`), every library launch, cross-stream wait and memset issued through this module is
    executed AND appended to a native step list; `host_callback` ends a segment, so that host-side work (a collective) can sit
    between two segments of the replay.  The recorder keeps everything the list points to alive: the tensors allocated during the
    recorded step (their storage is re-used by every replay, exactly as a captured HIP graph would) and the host-side arrays."""

    def __init__(self) -> None:
        self.lib = _lib.load()
        self.handle = self.lib.svsr_steplist_create()
        self.keep: list = []
        self.callbacks: dict[int, list[Callable[[], None]]] = {}      # segment index -> host calls that follow it
        self.main_stream: Optional[int] = None
        self.closed = False

    def __del__(self):
        try:
            if self.handle:
                self.lib.svsr_steplist_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def call(self, name: str, args: tuple) -> None:
        fn = getattr(self.lib, name)
        n = len(args)
        slots = (ctypes.c_int64 * n)()
        for i, (a, t) in enumerate(zip(args, fn.argtypes)):
            if t is ctypes.c_float:
                slots[i] = struct.unpack("<q", struct.pack("<d", float(a)))[0]
            elif t is ctypes.c_void_p:
                if a is None:
                    slots[i] = 0
                elif isinstance(a, int):
                    slots[i] = a
                else:                       # a ctypes array (plan meta record, tap table): host memory the launch reads
                    slots[i] = ctypes.addressof(a)
                    self.keep.append(a)
            else:
                slots[i] = int(a)
        rc = self.lib.svsr_steplist_push_call(self.handle, name.encode(), slots, n)
        if rc != 0:
            raise _lib.SvsrError(f"{name} cannot be recorded into a step list (code {rc}): not a stream launch of include/syncvsr_hip.h?")

    def wait(self, waiter: int, signaller: int) -> None:
        _lib.check(self.lib.svsr_steplist_push_wait(self.handle, waiter, signaller), "svsr_steplist_push_wait")

    def memset(self, ptr: int, value: int, nbytes: int, stream: int) -> None:
        _lib.check(self.lib.svsr_steplist_push_memset(self.handle, ptr, value, nbytes, stream), "svsr_steplist_push_memset")

    def add_callback(self, fn: Callable[[], None]) -> None:
        seg = self.lib.svsr_steplist_push_break(self.handle)
        if seg < 1:
            _lib.check(-seg if seg < 0 else 1001, "svsr_steplist_push_break")
        self.callbacks.setdefault(seg - 1, []).append(fn)

    @property
    def segments(self) -> int:
        return int(self.lib.svsr_steplist_segments(self.handle))

    @property
    def size(self) -> int:
        return int(self.lib.svsr_steplist_size(self.handle))

    def run(self) -> None:
        """Re-issues the recorded step: segment by segment, each followed by its host callbacks."""
        failed = ctypes.c_int(-1)
        run, h, cbs = self.lib.svsr_steplist_run, self.handle, self.callbacks
        if len(cbs) == 0:
            rc = run(h, -1, ctypes.byref(failed))
            if rc != 0:
                _lib.check(rc, f"svsr_steplist_run (op {failed.value})")
            return
        for k in range(self.segments):
            rc = run(h, k, ctypes.byref(failed))
            if rc != 0:
                _lib.check(rc, f"svsr_steplist_run (op {failed.value})")
            for fn in cbs.get(k, ()):
                fn()


_REC: Optional[StepRecorder] = None
_ALLOC_KEEP = ("empty", "empty_like")
_ALLOC_FORBID = ("zeros", "zeros_like", "ones", "ones_like", "full", "full_like")      # would launch a torch kernel the list does not hold


@contextmanager
def recording(rec: StepRecorder):
    """Everything launched through this module inside the block is also appended to `rec`.  torch.empty / torch.empty_like results
    are kept alive by the recorder (the replay re-uses their storage); torch.zeros & co. raise — use ops.zeros, whose fill is
    recorded.  Launch plans are pinned (their device words are referenced by the list)."""
    global _REC, PLAN_CACHE_PINNED
    if _REC is not None or _TIMING is not None:
        raise RuntimeError("nested recording / recording while per-launch event timing is on")
    saved = {n: getattr(torch, n) for n in _ALLOC_KEEP + _ALLOC_FORBID}

    def keeper(fn):
        def w(*a, **k):
            t = fn(*a, **k)
            rec.keep.append(t)
            return t
        return w

    def forbid(name):
        def w(*a, **k):
            raise RuntimeError(f"torch.{name} inside a recorded step launches a kernel the native step list does not hold (use ops.zeros)")
        return w

    for n in _ALLOC_KEEP:
        setattr(torch, n, keeper(saved[n]))
    for n in _ALLOC_FORBID:
        setattr(torch, n, forbid(n))
    _REC = rec
    rec.main_stream = _stream()
    PLAN_CACHE_PINNED = True
    try:
        yield rec
    finally:
        _REC = None
        for n, f in saved.items():
            setattr(torch, n, f)
        rec.closed = True


def host_callback(fn: Callable[[], None]) -> None:
    """Runs fn() now; inside a recorded step it also closes the current segment and is called again after that segment in
    every replay (collectives and their stream joins: engine.GradReducer)."""
    fn()
    if _REC is not None:
        _REC.add_callback(fn)


def stream_wait(waiter: torch.cuda.Stream, signaller: torch.cuda.Stream) -> None:
    """`waiter` waits for everything enqueued so far on `signaller`.  Eager: torch's own event path; recorded steps: the library's."""
    if _REC is None:
        waiter.wait_stream(signaller)
        return
    w, s = waiter.cuda_stream, signaller.cuda_stream
    _lib.check(_lib.load().svsr_stream_wait(w, s), "svsr_stream_wait")
    _REC.wait(w, s)


def device_cus() -> int:
    """Compute units of the current device."""
    return int(_lib.load().svsr_device_cus())


def shader_clock_mhz(blocks: int = 256, iters: int = 20000) -> float:
    """Effective shader clock (MHz) over a ~1 ms block of MFMA work on every compute unit, right now, on the current stream: the ratio of
    s_memtime (shader cycles) to s_memrealtime (100 MHz) inside svsr_clock_probe.  Synchronises the stream."""
    out = torch.zeros(blocks * 2 + 1, dtype=torch.int64, device="cuda")
    _lib.check(_lib.load().svsr_clock_probe(_p(out), blocks, iters, _stream()), "svsr_clock_probe")
    torch.cuda.current_stream().synchronize()
    h = out[: blocks * 2].view(blocks, 2).sum(0).tolist()
    return 100.0 * h[0] / max(1, h[1])


def memset(t: torch.Tensor, value: int = 0) -> None:
    """hipMemsetAsync over a contiguous tensor on the current stream."""
    nbytes, stream = t.numel() * t.element_size(), _stream()
    _lib.check(_lib.load().svsr_memset_async(t.data_ptr(), value, nbytes, stream), "svsr_memset_async")
    if _REC is not None:
        _REC.memset(t.data_ptr(), value, nbytes, stream)


def zeros(shape, dtype, device) -> torch.Tensor:
    t = torch.empty(shape, dtype=dtype, device=device)
    memset(t, 0)
    return t


def word_add(word: torch.Tensor, delta: int) -> None:
    _call("svsr_word_add", _p(word), int(delta), _stream())


def lincomb2(a: torch.Tensor, b: torch.Tensor, wb: float) -> torch.Tensor:
    """0-d fp32: a + wb * b, on the device."""
    out = torch.empty((), dtype=torch.float32, device=a.device)
    _call("svsr_lincomb2", _p(a), _p(b), float(wb), _p(out), _stream())
    return out


def lincomb3_ratio(a, wa: float, b, wb: float, c, wc: float, num=None, den=None):
    """0-d fp32 (wa*a + wb*b) + wc*c and, with num / den, their quotient: on the device, torch's roundings."""
    out0 = torch.empty((), dtype=torch.float32, device=a.device)
    out1 = torch.empty((), dtype=torch.float32, device=a.device) if num is not None else None
    _call("svsr_lincomb3_ratio", _p(a), float(wa), _p(b), float(wb), _p(c), float(wc), _p(out0), _p(num), _p(den), _p(out1), _stream())
    return out0, out1


def _call(name: str, *args, label: Optional[str] = None, flops: float = 0.0, nbytes: float = 0.0) -> None:
    timing = _TIMING
    if timing is not None:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
    rc = getattr(_lib.load(), name)(*args)
    if timing is not None:
        e.record()
        timing.setdefault(label or name, []).append((s, e, flops, nbytes))
    if rc != 0:
        _lib.check(rc, name)
    if _REC is not None:
        _REC.call(name, args)


class Plan:
    """Launch plan of svsr_igemm_fwd for one shape: device copy of the plan words + the host meta record."""
    __slots__ = ("words", "meta", "bm", "bn", "ns", "tiles", "label")

    def __init__(self, builder: str, *args):
        fn = getattr(_lib.load(), builder)
        meta = (ctypes.c_int * 8)()
        n = fn(*args, None, 0, meta)
        if n <= 0:
            _lib.check(-n if n < 0 else 1001, builder)
        host = torch.empty(n, dtype=torch.int32)
        n2 = fn(*args, host.data_ptr(), n, meta)
        if n2 != n:
            _lib.check(1001, builder)
        self.words = host.to("cuda")
        self.meta = meta
        self.bm, self.bn, self.ns, self.tiles = int(meta[0]), int(meta[1]), int(meta[2]), int(meta[3])
        self.label = f"k_igemm_p8<{self.bm},{self.bn},{self.ns}>" if self.bm == 256 else f"k_igemm_fwd_glds<{self.bm},{self.bn},{self.ns}>"


def conv_plan(mode: int, N: int, H: int, W: int, Co_out: int, k: int, stride: int, pad: int) -> Plan:
    """mode 0: forward convolution of N images [H, W]; mode 1: its input-gradient (svsr_conv_plan)."""
    key = ("conv", mode, N, H, W, Co_out, k, stride, pad)
    pl = _PLAN_CACHE.get(key)
    if pl is None:
        pl = _PLAN_CACHE[key] = Plan("svsr_conv_plan", mode, N, H, W, Co_out, k, stride, pad)
    return pl


def rows_plan(Nimg: int, P: int, src0: int, dst0: int, Co_out: int, Ci: Optional[int] = None) -> Plan:
    """Plan of a dense layer over rows grouped in Nimg sequences (svsr_rows_plan); Ci is accepted for the callers' convenience and unused."""
    key = ("rows", Nimg, P, src0, dst0, Co_out)
    pl = _PLAN_CACHE.get(key)
    if pl is None:
        pl = _PLAN_CACHE[key] = Plan("svsr_rows_plan", Nimg, P, src0, dst0, Co_out)
    return pl


class WPlan:
    """Launch plan of svsr_igemm_wgrad for one shape (device words + host meta + workspace size)."""
    __slots__ = ("words", "meta", "bc", "splits", "part_floats", "label")

    def __init__(self, builder: str, *args):
        fn = getattr(_lib.load(), builder)
        meta = (ctypes.c_int * 8)()
        nfl = ctypes.c_int64(0)
        n = fn(*args, None, 0, meta, ctypes.byref(nfl))
        if n <= 0:
            _lib.check(-n if n < 0 else 1001, builder)
        host = torch.empty(n, dtype=torch.int32)
        if fn(*args, host.data_ptr(), n, meta, ctypes.byref(nfl)) != n:
            _lib.check(1001, builder)
        self.words = host.to("cuda")
        self.meta = meta
        self.bc, self.splits, self.part_floats = int(meta[0]), int(meta[2]), int(nfl.value)
        self.label = f"k_igemm_wgrad<{self.bc},{int(meta[1])}>"


def wgrad_conv_plan(N: int, H: int, W: int, Ci: int, Co: int, k: int, stride: int, pad: int) -> WPlan:
    key = ("wconv", N, H, W, Ci, Co, k, stride, pad)
    pl = _PLAN_CACHE.get(key)
    if pl is None:
        pl = _PLAN_CACHE[key] = WPlan("svsr_wgrad_plan", N, H, W, Ci, Co, k, stride, pad)
    return pl


def wgrad_rows_plan(Nimg: int, P: int, src0: int, dst0: int, Ci: int, Co: int, has_bias: bool) -> WPlan:
    key = ("wrows", Nimg, P, src0, dst0, Ci, Co, bool(has_bias))
    pl = _PLAN_CACHE.get(key)
    if pl is None:
        pl = _PLAN_CACHE[key] = WPlan("svsr_wgrad_rows_plan", Nimg, P, src0, dst0, Ci, Co, int(has_bias))
    return pl


# --------------------------------------------------------------------------------------------------
# implicit GEMM
# --------------------------------------------------------------------------------------------------
def igemm_fwd(plan: Plan, inp: torch.Tensor, wt: torch.Tensor, out: torch.Tensor, *, Nimg: int, in_pix: int, Ci: int, in_pitch: int,
              Co: int, out_pix: int, out_pitch: int, wt_taps: int = 1, bias: Optional[torch.Tensor] = None,
              addend: Optional[torch.Tensor] = None, want_stats: bool = False, gelu: bool = False,
              out_pre: Optional[torch.Tensor] = None, out_f32: bool = False, relu: bool = False, alpha: float = 1.0, drop=None,
              flops: float = 0.0):
    """Runs a plan.  -> None, or with want_stats the BatchNorm partials (buffer, rows) for bn_finalize."""
    stats = st = None
    if want_stats:
        stats = scratch(plan.tiles * 2 * Co)
        st = (stats, plan.tiles)
    label = plan.label
    if _TIMING is not None and plan.bm == 64 and int(_lib.load().svsr_igemm_fwd_kgroups(plan.meta, Ci, Co, 0)) == 2:
        label = label[:-1] + ",2>"          # the in-workgroup K-split instantiation: the profiler's k_igemm_fwd_glds<64, 64, 4, 2>
    _call("svsr_igemm_fwd", _p(inp), _p(wt), _p(out), _p(out_pre), _p(bias), _p(addend), _p(stats), plan.words.data_ptr(), plan.meta,
          Nimg, in_pix, Ci, in_pitch, Co, out_pix, out_pitch, wt_taps, 1 if gelu else (2 if relu else 0), int(out_f32), float(alpha),
          *_drop(drop), _stream(), label=label, flops=flops)
    return st


def igemm_wgrad(plan: WPlan, x: torch.Tensor, dyp: torch.Tensor, dw: torch.Tensor, *, Nimg: int, in_pix: int, Ci: int, in_pitch: int,
                Co: int, out_pix: int, out_pitch: int, wt_taps: int = 1, db: Optional[torch.Tensor] = None, flops: float = 0.0) -> None:
    part = scratch(plan.part_floats) if plan.part_floats else None
    _call("svsr_igemm_wgrad", _p(x), _p(dyp), _p(dw), _p(db), plan.words.data_ptr(), plan.meta, Nimg, in_pix, Ci, in_pitch, Co, out_pix,
          out_pitch, wt_taps, _p(part), plan.part_floats, _stream(), label=plan.label, flops=flops)


class _WgradProblem(ctypes.Structure):
    """svsr_wgrad_problem of include/syncvsr_hip.h"""
    _fields_ = [("x", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("dbias", ctypes.c_void_p), ("plan_dev", ctypes.c_void_p),
                ("meta", ctypes.c_void_p), ("Nimg", ctypes.c_int), ("in_pix", ctypes.c_int), ("Ci", ctypes.c_int), ("in_pitch", ctypes.c_int),
                ("Co", ctypes.c_int), ("out_pix", ctypes.c_int), ("out_pitch", ctypes.c_int), ("wt_taps", ctypes.c_int)]


WGRAD_GROUP = True      # the encoder's / heads' linear weight gradients of a backward pass in ONE launch (False: one launch each)


_WG_TABLES: dict = {}


def _group_table(nbytes: int, device) -> torch.Tensor:
    """Problem table of a grouped launch: one grow-only buffer per (device, stream the launch goes to).  The launch may run on the
    model's side stream (STREAM_OVERRIDE) while this function's allocations belong to torch's current stream: a tensor made here and
    dropped on return could be handed to the next main-stream allocation while the side stream still reads it.  In-stream order makes
    re-use by the next grouped launch on the same stream safe (its table writer runs behind this launch's contraction)."""
    key = (str(device), _stream())
    t = _WG_TABLES.get(key)
    if t is None or t.numel() < nbytes:
        if t is not None:
            _SCRATCH_KEEP.append(t)          # a captured graph / recorded step list / pending launch may still hold its address
        t = _WG_TABLES[key] = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
    return t


def linear_wgrad_group(problems: Sequence[dict]) -> None:
    """problems: keyword arguments of linear_wgrad calls (x, dy, dw, rows, K, N, x_pitch, dy_pitch, seq, db).  Those whose plan has no K
    split on 64-wide tiles go out as one svsr_igemm_wgrad_group launch; the others (none at the LRW shapes) as launches of their own."""
    grouped, keep = [], []
    for q in problems:
        rows, K, N, seq, db = q["rows"], q["K"], q["N"], q.get("seq"), q.get("db")
        if seq is None:
            plan, geo = wgrad_rows_plan(rows, 1, 0, 0, K, N, db is not None), (rows, 1, 1)
        else:
            S, s0, n = seq
            plan, geo = wgrad_rows_plan(rows // n, n, s0, 0, K, N, db is not None), (rows // n, S, n)
        if not (WGRAD_GROUP and plan.bc == 64 and int(plan.meta[1]) == 3 and plan.splits == 1):
            linear_wgrad(q["x"], q["dy"], q["dw"], rows=rows, K=K, N=N, x_pitch=q["x_pitch"], dy_pitch=q["dy_pitch"], seq=seq, db=db)
            continue
        grouped.append(_WgradProblem(_p(q["x"]), _p(q["dy"]), _p(q["dw"]), _p(db), plan.words.data_ptr(), ctypes.addressof(plan.meta),
                                     geo[0], geo[1], K, q["x_pitch"], N, geo[2], q["dy_pitch"], 1))
        keep.append(plan)
    if not grouped:
        return
    arr = (_WgradProblem * len(grouped))(*grouped)
    nbytes = int(_lib.load().svsr_igemm_wgrad_group_bytes(len(grouped)))
    table = _group_table(nbytes, problems[0]["x"].device)
    flops = sum(2.0 * q["rows"] * q["K"] * q["N"] for q in problems)
    _call("svsr_igemm_wgrad_group", arr, len(grouped), _p(table), nbytes, _stream(), label="k_igemm_wgrad_group<64,3>", flops=flops)


def conv_out_size(n: int, k: int, stride: int, pad: int) -> int:
    return (n + 2 * pad - k) // stride + 1


def conv2d_fwd(x: torch.Tensor, w16: torch.Tensor, k: int, stride: int, pad: int, want_stats: bool = False):
    """x [N,H,W,Ci] bf16, w16 bf16 [Co][k][k][Ci] -> ([N,Ho,Wo,Co] bf16, stats) where stats is None or the BatchNorm partial
    sums (buffer, rows) to hand to bn_finalize next on this stream."""
    N, H, W, Ci = x.shape
    Co = w16.shape[0]
    Ho, Wo = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    out = torch.empty((N, Ho, Wo, Co), dtype=BF16, device=x.device)
    if _c64_ok(Ci, Co, k, stride, pad, W):
        return out, conv3x3_c64(x, w16, out, None, want_stats, _conv_taps(k, pad))
    st = igemm_fwd(conv_plan(0, N, H, W, Co, k, stride, pad), x, w16, out, Nimg=N, in_pix=H * W, Ci=Ci, in_pitch=Ci, Co=Co,
                   out_pix=Ho * Wo, out_pitch=Co, wt_taps=k * k, want_stats=want_stats, flops=2.0 * N * Ho * Wo * Co * Ci * k * k)
    return out, st


_TAPS: dict = {}


def _conv_taps(k: int, pad: int) -> tuple:
    t = _TAPS.get((k, pad))
    if t is None:
        t = _TAPS[(k, pad)] = tuple((kh - pad, kw - pad, kh * k + kw) for kh in range(k) for kw in range(k))
    return t


C64_CONV = True        # persistent weights-in-LDS kernel for conv3x3(64, 64) stride 1 (False: generic implicit GEMM)


_C64_TABS: dict = {}


def c64_pixtab(N: int, H: int, W: int, device) -> torch.Tensor:
    """Device copy of svsr_conv3x3_c64_pixtab(N, H, W) (padded coordinate -> pixel index or -1), cached per shape like a launch plan."""
    key = (N, H, W, str(device))
    t = _C64_TABS.get(key)
    if t is None:
        lib = _lib.load()
        n = int(lib.svsr_conv3x3_c64_pixtab(N, H, W, None, 0))
        if n <= 0:
            raise _lib.SvsrError(f"svsr_conv3x3_c64_pixtab({N}, {H}, {W}) failed ({n})")
        host = torch.empty(n, dtype=torch.int32)
        got = int(lib.svsr_conv3x3_c64_pixtab(N, H, W, host.data_ptr(), n))
        if got != n:
            raise _lib.SvsrError(f"svsr_conv3x3_c64_pixtab wrote {got} of {n} entries")
        t = _C64_TABS[key] = host.to(device)
    return t


def _c64_ok(Ci: int, Co: int, k: int, stride: int, pad: int, W: int) -> bool:
    return C64_CONV and Ci == 64 and Co == 64 and k == 3 and stride == 1 and pad == 1 and W <= 29


def conv3x3_c64(x: torch.Tensor, wt: torch.Tensor, out: torch.Tensor, addend: Optional[torch.Tensor], want_stats: bool,
                taps: Sequence[tuple[int, int, int]]):
    N, H, W, _ = x.shape
    dy, dx, tw = zip(*taps)
    stats = st = None
    if want_stats:
        rows = _query("svsr_conv3x3_c64_stat_rows", N, H, W)[0]
        stats = scratch(rows * 2 * 64)
        st = (stats, rows)
    _call("svsr_conv3x3_c64", _p(x), _p(wt), _p(out), _p(addend), _p(stats), N, H, W, _ints(dy), _ints(dx), _ints(tw), _p(c64_pixtab(N, H, W, x.device)), _stream(),
          label="k_conv3x3_c64", flops=2.0 * N * H * W * 64 * 64 * 9)
    return st


def conv2d_dgrad(dy: torch.Tensor, w16t: torch.Tensor, k: int, stride: int, pad: int, in_hw: tuple[int, int],
                 addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dy [N,Ho,Wo,Co], w16t bf16 [Ci][k][k][Co] (transposed shadow) -> dx [N,H,W,Ci] (+ addend, in place when given).

    One launch for any stride: the plan's classes cover the output-parity classes of a strided convolution with their own
    taps (the transposed convolution never multiplies by the inserted zeros), and pixels no tap reaches are written as 0."""
    N, Ho, Wo, Co = dy.shape
    Ci = w16t.shape[0]
    H, W = in_hw
    dx = torch.empty((N, H, W, Ci), dtype=BF16, device=dy.device) if addend is None else addend
    if stride == 1 and _c64_ok(Ci, Co, k, stride, pad, W):
        taps = tuple((pad - kh, pad - kw, kh * k + kw) for kh in range(k) for kw in range(k))
        conv3x3_c64(dy, w16t, dx, addend, False, taps)
        return dx
    igemm_fwd(conv_plan(1 if addend is None else 2, N, H, W, Ci, k, stride, pad), dy, w16t, dx, Nimg=N, in_pix=Ho * Wo, Ci=Co, in_pitch=Co, Co=Ci, out_pix=H * W,
              out_pitch=Ci, wt_taps=k * k, addend=addend, flops=2.0 * N * Ho * Wo * Co * Ci * k * k)
    return dx


CTC_SIDE = True         # LRS: the CTC branch of the forward on the model's side stream, beside the attention decoder (tuning knob "ctc_side")
BN_BWD_FUSED = True    # ReLU trunk: first pass of the BatchNorm backward inside the producing data-gradient launch (False: separate pass)


def conv2d_dgrad_bn(dy: torch.Tensor, w16t: torch.Tensor, k: int, stride: int, pad: int, in_hw: tuple[int, int],
                    addend: Optional[torch.Tensor], y: Optional[torch.Tensor], x: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor,
                    gamma: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None, act: int = 1):
    """conv2d_dgrad whose result is the gradient of y = relu(bn(x) [+ residual]) (y, x shaped like the result): the launch stores
    g = (y > 0) * (dgrad + addend) instead and takes the first pass of that BatchNorm's backward in its epilogue.
    y=None (no residual branch): the mask is recomputed from x, gamma, beta and y is not read.
    act=2 (Swish): g = (dgrad + addend) * swish'(bn(x) + r) where `y` is the residual INPUT r (or None); gamma / beta required.
    -> (g, (stats, rows)) for bn_bwd_from_stats.  In place when addend is given."""
    N, Ho, Wo, Co = dy.shape
    Ci = w16t.shape[0]
    H, W = in_hw
    if x.shape != (N, H, W, Ci) or not x.is_contiguous() or (y is not None and (y.shape != x.shape or not y.is_contiguous())):
        raise ValueError("conv2d_dgrad_bn: y / x must be contiguous tensors with the geometry of the result")
    if (y is None or act == 2) and (gamma is None or beta is None):
        raise ValueError("conv2d_dgrad_bn: y=None / act=2 need gamma and beta")
    g = torch.empty((N, H, W, Ci), dtype=BF16, device=dy.device) if addend is None else addend
    if stride == 1 and _c64_ok(Ci, Co, k, stride, pad, W):
        taps = tuple((pad - kh, pad - kw, kh * k + kw) for kh in range(k) for kw in range(k))
        tdy, tdx, tw = zip(*taps)
        rows = _query("svsr_conv3x3_c64_stat_rows", N, H, W)[0]
        stats = scratch(rows * 2 * 64)
        _call("svsr_conv3x3_c64_dgrad_bn", _p(dy), _p(w16t), _p(g), _p(addend), _p(stats), N, H, W, _ints(tdy), _ints(tdx), _ints(tw),
              _p(y), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), act, _p(c64_pixtab(N, H, W, dy.device)), _stream(), label="k_conv3x3_c64+bn", flops=2.0 * N * H * W * 64 * 64 * 9)
        return g, (stats, rows)
    plan = conv_plan(1, N, H, W, Ci, k, stride, pad)        # mode 1: every pixel of the result is visited once
    stats = scratch(plan.tiles * 2 * Ci)
    _call("svsr_igemm_dgrad_bn", _p(dy), _p(w16t), _p(g), _p(addend), _p(stats), plan.words.data_ptr(), plan.meta, N, Ho * Wo, Co, Co, Ci,
          H * W, Ci, k * k, _p(y), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), act, _stream(), label=plan.label + "+bn", flops=2.0 * N * Ho * Wo * Co * Ci * k * k)
    return g, (stats, plan.tiles)


def bn_bwd_from_stats(g, x, mean, rstd, gamma, stats, coef, dgamma, dbeta) -> torch.Tensor:
    """Second half of the BatchNorm backward after conv2d_dgrad_bn: -> dx (gradient of the BatchNorm input x)."""
    part, rows = stats
    C = x.shape[-1]
    dx = torch.empty_like(x)
    _call("svsr_bn_bwd_from_stats", _p(g), _p(x), _p(mean), _p(rstd), _p(gamma), _p(part), rows, _p(coef), _p(dgamma), _p(dbeta), _p(dx),
          x.numel() // C, C, _stream())
    return dx


def conv2d_wgrad(x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, k: int, stride: int, pad: int, use_tr: bool = True) -> None:
    """dw fp32 [Co][k][k][Ci] += sum dy[n,y,x,co] * x[n, y*s+kh-pad, x*s+kw-pad, ci]."""
    N, H, W, Ci = x.shape
    _, Ho, Wo, Co = dy.shape
    if HALO_WGRAD and use_tr and k == 3 and stride == 1 and pad == 1 and W <= 29 and H * W >= 100 and Ci % 64 == 0 and Co % 64 == 0:
        # all nine taps in one pass over zero-padded coordinates (wgrad3x3.hip)
        _, nfl = _query("svsr_conv3x3_wgrad_plan", N, H, W, Ci, Co)
        part = scratch(nfl) if nfl else None
        _call("svsr_conv3x3_wgrad", _p(x), _p(dy), _p(dw), N, H, W, Ci, Co, _p(part), nfl, _stream(), label="k_wgrad3x3_halo",
              flops=2.0 * N * H * W * Co * Ci * 9)
        return
    igemm_wgrad(wgrad_conv_plan(N, H, W, Ci, Co, k, stride, pad), x, dy, dw, Nimg=N, in_pix=H * W, Ci=Ci, in_pitch=Ci, Co=Co,
                out_pix=Ho * Wo, out_pitch=Co, wt_taps=k * k, flops=2.0 * N * Ho * Wo * Co * Ci * k * k)


def halo_wgrad_ok(x: torch.Tensor, dy: torch.Tensor, k: int, stride: int, pad: int) -> bool:
    N, H, W, Ci = x.shape
    Co = dy.shape[-1]
    return HALO_WGRAD and k == 3 and stride == 1 and pad == 1 and W <= 29 and H * W >= 100 and Ci % 64 == 0 and Co % 64 == 0


def linear_fwd(x: torch.Tensor, w16: torch.Tensor, bias: Optional[torch.Tensor], *, rows: int, K: int, N: int, x_pitch: int,
               out: Optional[torch.Tensor] = None, out_pitch: Optional[int] = None, gelu: bool = False,
               out_f32: bool = False, addend: Optional[torch.Tensor] = None,
               seq: Optional[tuple[int, int, int]] = None, relu: bool = False, alpha: float = 1.0,
               drop=None) -> tuple[torch.Tensor, Optional[torch.Tensor]]:
    """out[rows, N] = alpha * dropout(act(x[rows, K] @ w16[N, K]^T + bias)) + addend, act = gelu | relu | none.  `seq=(S, s0, n)` selects rows s0..s0+n-1 of every
    length-S sequence of x (x is [B*S, K]) as the source and writes a dense [B*n, N] result."""
    out_pitch = out_pitch or N
    if out is None:
        out = torch.empty((rows, out_pitch), dtype=torch.float32 if out_f32 else BF16, device=x.device)
    pre = torch.empty_like(out) if gelu else None
    if seq is None:
        plan, geo = rows_plan(rows, 1, 0, 0, N, None if (gelu or out_f32 or out_pitch % 8) else K), dict(Nimg=rows, in_pix=1, out_pix=1)
    else:
        S, s0, n = seq
        plan, geo = rows_plan(rows // n, n, s0, 0, N), dict(Nimg=rows // n, in_pix=S, out_pix=n)
    igemm_fwd(plan, x, w16, out, Ci=K, in_pitch=x_pitch, Co=N, out_pitch=out_pitch, bias=bias, addend=addend, gelu=gelu, out_pre=pre,
              out_f32=out_f32, relu=relu, alpha=alpha, drop=drop, flops=2.0 * rows * N * K, **geo)
    return out, pre


def linear_dgrad(dy: torch.Tensor, w16t: torch.Tensor, *, rows: int, N: int, K: int, dy_pitch: int, out: Optional[torch.Tensor] = None,
                 addend: Optional[torch.Tensor] = None, seq: Optional[tuple[int, int, int]] = None, alpha: float = 1.0,
                 drop=None) -> torch.Tensor:
    """dx[rows, K] = dy[rows, N] @ w16t[K, Npad]^T-of-transpose, i.e. dy @ W.  With `seq=(S, s0, n)` the dense dy rows
    [B*n] are scattered to rows s0.. of every length-S sequence of dx [B*S, K]."""
    Np = w16t.shape[-1]
    if seq is None:
        if out is None:
            out = torch.empty((rows, K), dtype=BF16, device=dy.device)
        plan, geo = rows_plan(rows, 1, 0, 0, K, Np), dict(Nimg=rows, in_pix=1, out_pix=1)
    else:
        S, s0, n = seq
        assert out is not None
        plan, geo = rows_plan(rows // n, n, 0, s0, K), dict(Nimg=rows // n, in_pix=n, out_pix=S)
    igemm_fwd(plan, dy, w16t, out, Ci=Np, in_pitch=dy_pitch, Co=K, out_pitch=K, addend=addend, alpha=alpha, drop=drop,
              flops=2.0 * rows * N * K, **geo)
    return out


def linear_dgrad_bn(dy: torch.Tensor, w16t: torch.Tensor, *, rows: int, N: int, K: int, dy_pitch: int, x: torch.Tensor, mean: torch.Tensor,
                    rstd: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, act: int):
    """linear_dgrad whose result is the gradient of act(bn(x)) (x [rows, K], no residual branch): the launch stores
    g = dgrad * act'(bn(x)) instead and takes the first pass of that BatchNorm's backward in its epilogue (see conv2d_dgrad_bn).
    -> (g, (stats, rows)) for bn_bwd_from_stats."""
    Np = w16t.shape[-1]
    if x.shape != (rows, K) or not x.is_contiguous():
        raise ValueError("linear_dgrad_bn: x must be a contiguous [rows, K] tensor")
    g = torch.empty((rows, K), dtype=BF16, device=dy.device)
    plan = rows_plan(rows, 1, 0, 0, K)
    stats = scratch(plan.tiles * 2 * K)
    _call("svsr_igemm_dgrad_bn", _p(dy), _p(w16t), _p(g), None, _p(stats), plan.words.data_ptr(), plan.meta, rows, 1, Np, dy_pitch, K,
          1, K, 1, None, _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), act, _stream(), label=plan.label + "+bn", flops=2.0 * rows * N * K)
    return g, (stats, plan.tiles)


_CONST_VECS: dict = {}


def linear_dgrad_relu(dy: torch.Tensor, w16t: torch.Tensor, *, rows: int, N: int, K: int, dy_pitch: int, y: torch.Tensor, gscale: float = 1.0):
    """linear_dgrad for a layer whose input was y = dropout(relu(z)) [rows, K]: the launch stores dz = (y > 0 ? gscale * dgrad : 0) and the column
    sums of dz per row tile (svsr_igemm_dgrad_relu) -> (dz, (stats, tiles)): colsum_rows(stats, tiles, 2 * K, db, K) is the bias gradient of
    the layer that produced z."""
    Np = w16t.shape[-1]
    if y.shape != (rows, K) or not y.is_contiguous():
        raise ValueError("linear_dgrad_relu: y must be a contiguous [rows, K] tensor")
    key = (str(dy.device), K)
    cv = _CONST_VECS.get(key)
    if cv is None:
        import numpy as np          # (host arrays + copies: made once, also inside a recorded step — no fill kernel for the recorder to miss)

        cv = _CONST_VECS[key] = tuple(torch.from_numpy(np.full(K, v, dtype=np.float32)).to(dy.device) for v in (0.0, 1.0))
    dz = torch.empty((rows, K), dtype=BF16, device=dy.device)
    plan = rows_plan(rows, 1, 0, 0, K)
    stats = torch.empty(plan.tiles * 2 * K, dtype=torch.float32, device=dy.device)      # (a buffer of its own: its sum may run later, on another stream)
    _call("svsr_igemm_dgrad_relu", _p(dy), _p(w16t), _p(dz), _p(stats), plan.words.data_ptr(), plan.meta, rows, 1, Np, dy_pitch, K, 1, K, 1,
          _p(y), _p(cv[0]), _p(cv[1]), float(gscale), _stream(), label=plan.label + "+relu", flops=2.0 * rows * N * K)
    return dz, (stats, plan.tiles)


def linear_wgrad(x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, *, rows: int, K: int, N: int, x_pitch: int, dy_pitch: int,
                 seq: Optional[tuple[int, int, int]] = None, use_tr: bool = True, db: Optional[torch.Tensor] = None) -> None:
    """dw fp32 [N][K] += dy[rows, N]^T @ x[rows, K];  db fp32 [N] += column sums of dy (bias gradient), if given.
    (`use_tr` is accepted for the callers' sake: the generic kernel always builds its fragments with transpose reads.)"""
    if seq is None:
        plan, geo = wgrad_rows_plan(rows, 1, 0, 0, K, N, db is not None), dict(Nimg=rows, in_pix=1, out_pix=1)
    else:
        S, s0, n = seq
        plan, geo = wgrad_rows_plan(rows // n, n, s0, 0, K, N, db is not None), dict(Nimg=rows // n, in_pix=S, out_pix=n)
    igemm_wgrad(plan, x, dy, dw, Ci=K, in_pitch=x_pitch, Co=N, out_pitch=dy_pitch, db=db, flops=2.0 * rows * N * K, **geo)


# --------------------------------------------------------------------------------------------------
# stem
# --------------------------------------------------------------------------------------------------
_STEM_WS: dict = {}


def stem_conv_fwd(videos: torch.Tensor, w: torch.Tensor, want_stats: bool = False):
    """-> (out [B*T,H/2,W/2,64] bf16, BatchNorm partials (buffer, rows) or None)"""
    B, C, T, H, W = videos.shape
    assert C == 1 and videos.dtype == torch.float32 and videos.is_contiguous()
    out = torch.empty((B * T, H // 2, W // 2, 64), dtype=BF16, device=videos.device)
    stats = st = None
    if want_stats:
        rows = _query("svsr_stem_conv_fwd_stat_rows", B, T, H, W)[0]
        stats = scratch(rows * 2 * 64)
        st = (stats, rows)
    nws = _query("svsr_stem_conv_fwd_ws_bytes", B, T, H, W)[0]
    ws = None
    if nws:         # bf16 copy of the clip + packed weights for the DMA-fed kernel: one grow-only buffer per device and stream (`stats`
        key = (videos.device, _stream())        # lives in the shared scratch; variable-length LRS batches must not pile up one buffer per shape)
        ws = _STEM_WS.get(key)
        if ws is None or ws.numel() < nws:
            if ws is not None:
                _SCRATCH_KEEP.append(ws)      # a captured graph / recorded step list may still hold its address
            ws = _STEM_WS[key] = torch.empty(nws, dtype=torch.uint8, device=videos.device)
    _call("svsr_stem_conv_fwd", _p(videos), _p(w), _p(out), _p(stats), B, T, H, W, _p(ws), nws, _stream(),
          label="k_stem_conv_fwd", flops=2.0 * B * T * (H // 2) * (W // 2) * 64 * 245)
    return out, st


def stem_conv_wgrad(videos: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, use_tr: bool = True) -> None:
    B, _, T, H, W = videos.shape
    _, nfl = _query("svsr_stem_conv_wgrad_plan", B, T, H, W)
    _call("svsr_stem_conv_wgrad", _p(videos), _p(dy), _p(dw), B, T, H, W, int(use_tr), _p(scratch(nfl)), nfl, _stream(),
          label=f"k_stem_conv_wgrad<{'true' if use_tr else 'false'}>", flops=2.0 * B * T * (H // 2) * (W // 2) * 64 * 245)


ACT_NONE, ACT_RELU, ACT_GELU, ACT_SWISH = 0, 1, 1, 2     # ReLU/GELU share code 1: ReLU in the BN passes, GELU in the stem pass


STEM_KEEP_WINNERS = os.environ.get("SVSR_STEM_KEEP_WINNERS", "1") != "0"     # the stem's forward keeps the convolution output at every pooling window's arg-max for its backward


def stem_bn_gelu_pool_fwd(x: torch.Tensor, mean, rstd, gamma, beta, act: int = ACT_GELU, want_win: bool = False):
    """-> (y, amax) or, with want_win, (y, amax, xwin): xwin = the convolution output at every window's arg-max, for
    stem_bn_gelu_pool_bwd(xwin=...) (reduce pass over pooled-size tensors, apply pass without activation derivatives)."""
    N, Hc, Wc, C = x.shape
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    y = torch.empty((N, Hp, Wp, C), dtype=BF16, device=x.device)
    amax = torch.empty((N, Hp, Wp, C), dtype=torch.uint8, device=x.device)
    xwin = torch.empty((N, Hp, Wp, C), dtype=BF16, device=x.device) if want_win else None
    _call("svsr_stem_bn_act_pool_fwd", _p(x), _p(y), _p(amax), _p(mean), _p(rstd), _p(gamma), _p(beta), N, Hc, Wc, Hp, Wp, C, act, _p(xwin), _stream())
    return (y, amax, xwin) if want_win else (y, amax)


def stem_bn_gelu_pool_bwd(dpool, amax, x, mean, rstd, gamma, beta, coef, dgamma, dbeta, act: int = ACT_GELU, xwin=None, want_dx: bool = True):
    """-> dx (gradient of the stem convolution's output); with want_dx=False (needs xwin) -> gpool, and `coef` is left filled: the inputs of
    stem_bwd_wgrad, which makes dx tile by tile inside the weight-gradient pass instead of writing it."""
    N, Hc, Wc, C = x.shape
    _, Hp, Wp, _ = dpool.shape
    assert want_dx or xwin is not None
    dx = torch.empty_like(x) if want_dx else None
    slots = scratch(_query("svsr_stem_bn_act_pool_bwd_rows", N, Hc, Wc, C)[0] * 2 * C)
    gpool = torch.empty_like(dpool) if xwin is not None else None
    _call("svsr_stem_bn_act_pool_bwd", _p(dpool), _p(amax), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(slots), _p(coef),
          _p(dgamma), _p(dbeta), _p(dx), N, Hc, Wc, Hp, Wp, C, act, _p(xwin), _p(gpool), _stream())
    return dx if want_dx else gpool


STEM_BWD_FUSED = True        # the stem's backward apply pass inside its weight-gradient pass (svsr_stem_bwd_wgrad) where the shape allows


def stem_bwd_wgrad_ok(videos: torch.Tensor) -> bool:
    B, _, T, H, W = videos.shape
    return bool(STEM_BWD_FUSED and _lib.load().svsr_stem_bwd_wgrad_ok(B, T, H, W))


def stem_bwd_wgrad(videos: torch.Tensor, gpool, amax, x, mean, rstd, coef, dw: torch.Tensor) -> None:
    """dw += the stem convolution's weight gradient, from the pooled-size gradient `gpool` (stem_bn_gelu_pool_bwd(want_dx=False)),
    the window winners `amax`, the convolution output `x` and the finalised BatchNorm-backward coefficients `coef`."""
    B, _, T, H, W = videos.shape
    _, nfl = _query("svsr_stem_conv_wgrad_plan", B, T, H, W)
    _call("svsr_stem_bwd_wgrad", _p(videos), _p(gpool), _p(amax), _p(x), _p(mean), _p(rstd), _p(coef), _p(dw), B, T, H, W, _p(scratch(nfl)), nfl,
          _stream(), label="k_stem_bwd_wgrad", flops=2.0 * B * T * (H // 2) * (W // 2) * 64 * 245)


# --------------------------------------------------------------------------------------------------
# BatchNorm / activation / pooling
# --------------------------------------------------------------------------------------------------
def bn_finalize(stats, C: int, count: int, mean, rstd, running_mean=None, running_var=None, nbt=None) -> None:
    """stats = (buffer, rows): the partial sums [rows][2][C] the producing convolution just wrote on this stream."""
    part, rows = stats
    _call("svsr_bn_finalize", _p(part), rows, C, float(count), BN_EPS, BN_MOMENTUM, _p(mean), _p(rstd), _p(running_mean), _p(running_var),
          _p(nbt), _stream())


def bn_eval_prepare(running_mean, running_var, mean, rstd) -> None:
    _call("svsr_bn_eval_prepare", _p(running_mean), _p(running_var), running_mean.numel(), BN_EPS, _p(mean), _p(rstd), _stream())


def bn_act_fwd(x, res, mean, rstd, gamma, beta, act: int) -> torch.Tensor:
    C = x.shape[-1]
    y = torch.empty_like(x)
    _call("svsr_bn_act_fwd", _p(x), _p(res), _p(y), _p(mean), _p(rstd), _p(gamma), _p(beta), x.numel() // C, C, act, _stream())
    return y


def bn_act_bwd(dy, y, x, mean, rstd, gamma, coef, dgamma, dbeta, act: int, want_dres: bool, beta=None, res=None):
    C = x.shape[-1]
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    slots = scratch(_query("svsr_bn_act_bwd_rows", x.numel() // C, C)[0] * 2 * C)
    _call("svsr_bn_act_bwd", _p(dy), _p(y), _p(x), _p(mean), _p(rstd), _p(gamma), _p(slots), _p(coef), _p(dgamma), _p(dbeta), _p(dx),
          _p(dres), x.numel() // C, C, act, _p(beta), _p(res), _stream())
    return dx, dres


def avgpool_fwd(x: torch.Tensor) -> torch.Tensor:
    N, H, W, C = x.shape
    y = torch.empty((N, C), dtype=BF16, device=x.device)
    _call("svsr_avgpool_fwd", _p(x), _p(y), N, H * W, C, _stream())
    return y


def avgpool_bwd(dy: torch.Tensor, shape) -> torch.Tensor:
    N, H, W, C = shape
    dx = torch.empty(shape, dtype=BF16, device=dy.device)
    _call("svsr_avgpool_bwd", _p(dy), _p(dx), N, H * W, C, _stream())
    return dx


# --------------------------------------------------------------------------------------------------
# transformer passes
# --------------------------------------------------------------------------------------------------
def add_ln_fwd(a, r, gamma, beta, eps: float):
    R, D = a.shape
    y = torch.empty_like(a)
    mean = torch.empty(R, dtype=torch.float32, device=a.device)
    rstd = torch.empty(R, dtype=torch.float32, device=a.device)
    _call("svsr_add_ln_fwd", _p(a), _p(r), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), R, D, eps, _stream())
    return y, mean, rstd


def colsum_rows(part, rows: int, ld: int, out0, n0: int, out1=None, n1: int = 0) -> None:
    _call("svsr_colsum_rows", _p(part), rows, ld, _p(out0), n0, _p(out1), n1, 1, 1.0, _stream())


class _ColsumEntry(ctypes.Structure):       # include/syncvsr_hip.h svsr_colsum_rows_multi: 64 bytes
    _fields_ = [("ws", ctypes.c_void_p), ("out0", ctypes.c_void_p), ("out1", ctypes.c_void_p), ("ld", ctypes.c_int64), ("n0", ctypes.c_int64),
                ("n1", ctypes.c_int64), ("nrows", ctypes.c_int32), ("accumulate", ctypes.c_int32), ("scale", ctypes.c_float), ("reserved", ctypes.c_int32)]


COLSUM_MULTI = True       # host-side knob "colsum_multi": a layer's postponed parameter-gradient reductions as one launch per 16


def _deferred_colsum(part, rows: int, ld: int, out0, n0: int, out1=None, n1: int = 0):
    """A postponed colsum_rows(...) (accumulating) as a closure for a model's deferred list; run_deferred() merges neighbours into one launch."""
    f = lambda: colsum_rows(part, rows, ld, out0, n0, out1, n1)
    f.colsum = (part, rows, ld, out0, n0, out1, n1)
    return f


def run_deferred(fns) -> None:
    """Runs a model's postponed reductions on the current stream: those made by _deferred_colsum as svsr_colsum_rows_multi launches (results
    identical to the separate launches: every problem keeps its own order of additions), the others one by one."""
    batch: list = []

    def flush():
        if len(batch) == 1:
            colsum_rows(*batch[0])
        elif batch:
            arr = (_ColsumEntry * len(batch))()
            for e, (part, rows, ld, out0, n0, out1, n1) in zip(arr, batch):
                e.ws, e.out0, e.out1, e.ld, e.n0, e.n1, e.nrows, e.accumulate, e.scale = _p(part), _p(out0), _p(out1), ld, n0, n1, rows, 1, 1.0
            _call("svsr_colsum_rows_multi", arr, len(batch), _stream())
        batch.clear()

    for f in fns:
        a = getattr(f, "colsum", None)
        if a is None or not COLSUM_MULTI:
            flush()
            f()
            continue
        outs = {_p(a[3]), _p(a[5])} - {None}
        if any(({_p(b[3]), _p(b[5])} - {None}) & outs for b in batch):       # (two contributions to one gradient: separate launches, in order)
            flush()
        batch.append(a)
    flush()


LN_BRANCH_FUSED = True      # host-side knob "ln_branch_fused": a residual branch's gradient alpha * mask * dx as a second output of the LayerNorm backward in front of it (LRS layers); False: svsr_scale_bf16


def add_ln_bwd(dy, a, r, gamma, mean, rstd, dgamma, dbeta, addend=None, out=None, defer: Optional[list] = None, branch=None):
    """defer = a list: the parameter-gradient reduction is not launched; a closure that launches it (on whatever stream is current when it
    is called) is appended instead — the partial rows then live in a buffer of their own.
    branch = (alpha, drop) (with defer): additionally returns alpha * dropout_mask(ds) — the gradient of the residual branch that ends in this
    sum, bit for bit what scale_bf16(ds, alpha, drop) would compute — as a second tensor: -> (ds, ds_branch)."""
    R, D = a.shape
    ds = torch.empty_like(a) if out is None else out
    if defer is not None:
        rows = _query("svsr_add_ln_bwd_rows", R)[0]
        part = torch.empty(rows * 2 * D, dtype=torch.float32, device=a.device)
        if branch is not None:
            alpha, drop = branch
            ds2 = torch.empty_like(a)
            _call("svsr_add_ln_bwd_branch", _p(dy), _p(a), _p(r), _p(gamma), _p(mean), _p(rstd), _p(ds), R, D, _p(addend), _p(part), _p(ds2), float(alpha),
                  *_drop(drop), _stream())
            defer.append((_deferred_colsum(part, rows, 2 * D, dgamma, D, dbeta, D), part))
            return ds, ds2
        _call("svsr_add_ln_bwd_partials", _p(dy), _p(a), _p(r), _p(gamma), _p(mean), _p(rstd), _p(ds), R, D, _p(addend), _p(part), _stream())
        defer.append((_deferred_colsum(part, rows, 2 * D, dgamma, D, dbeta, D), part))
        return ds
    part = scratch(_query("svsr_add_ln_bwd_rows", R)[0] * 2 * D)
    _call("svsr_add_ln_bwd", _p(dy), _p(a), _p(r), _p(gamma), _p(mean), _p(rstd), _p(ds), _p(dgamma), _p(dbeta), R, D, _p(addend), _p(part),
          _stream())
    return ds


def embed_ln_fwd(feats, cls, pos, type0, gamma, beta, B: int, S: int, D: int, eps: float, drop_in=None, drop_out=None):
    """drop_in / drop_out = (seed word, site, p) of emb_dropout and BertEmbeddings' dropout (same seed word), or None."""
    dev = feats.device
    seed = (drop_in or drop_out or (None,))[0]
    si, pi = (drop_in[1], drop_in[2]) if drop_in is not None else (0, 0.0)
    so, po = (drop_out[1], drop_out[2]) if drop_out is not None else (0, 0.0)
    s = torch.empty((B * S, D), dtype=BF16, device=dev)
    y = torch.empty((B * S, D), dtype=BF16, device=dev)
    mean = torch.empty(B * S, dtype=torch.float32, device=dev)
    rstd = torch.empty(B * S, dtype=torch.float32, device=dev)
    _call("svsr_embed_ln_fwd", _p(feats), _p(cls), _p(pos), _p(type0), _p(gamma), _p(beta), _p(s), _p(y), _p(mean), _p(rstd), B, S, D,
          eps, _p(seed), int(si), float(pi), int(so), float(po), _stream())
    return s, y, mean, rstd


def embed_bwd_scatter(ds, dcls, dpos, dtype0, B: int, S: int, D: int, drop_in=None) -> torch.Tensor:
    dfeats = torch.empty((B * (S - 1), D), dtype=BF16, device=ds.device)
    _call("svsr_embed_bwd_scatter", _p(ds), _p(dfeats), _p(dcls), _p(dpos), _p(dtype0), B, S, D, _p(scratch(S * D)), *_drop(drop_in),
          _stream())
    return dfeats


# -- fused forward of the HF-BERT encoder layers (csrc/enc_fused.hip) ---------------------------------
class _EncLayer(ctypes.Structure):
    """svsr_enc_layer of include/syncvsr_hip.h"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("wqkv", "wo", "w1", "w2", "bqkv", "bo", "b1", "b2", "g1", "be1", "g2", "be2",
                                               "qkv", "probs", "ctx", "ao", "x1", "z", "hg", "f", "xout", "m1", "r1", "m2", "r2")] + \
               [("site_probs", ctypes.c_uint), ("site_ao", ctypes.c_uint), ("site_fo", ctypes.c_uint), ("pad_", ctypes.c_uint)]


ENC_FUSED = os.environ.get("SVSR_ENC_FUSED", "1") != "0"     # the whole encoder forward as one launch (False: seven launches per layer)
_ENC_WS: dict = {}


_ENC_FUSED_USED = False      # a fused-encoder launch has been issued by this process (svsr_enc_fwd / svsr_enc_bwd)


def enc_gave_up_peek() -> bool:
    """True once a fused-encoder launch had a bounded cluster wait give up — read from pinned host memory, NO synchronisation (the value may
    trail the device by the flight time of one store): engine.TrainStep looks at it before every step."""
    return _ENC_FUSED_USED and bool(_lib.load().svsr_enc_gave_up_peek())


def check_enc_clusters(reset: bool = True) -> bool:
    """True if any svsr_enc_fwd / svsr_enc_bwd launch since the last reset had a bounded cluster wait give up (synchronises the device; not
    called at all while no fused launch has been issued).  Raises when the flag itself cannot be read (a sticky HIP error, the wrong device)."""
    if not _ENC_FUSED_USED:
        return False
    rc = int(_lib.load().svsr_enc_gave_up(1 if reset else 0))
    if rc < 0:
        raise _lib.SvsrError("svsr_enc_gave_up could not read the fused encoder's error word (hipMemcpyFromSymbol failed: an earlier HIP error is "
                             "pending on this device, or the library was loaded for another one)")
    return rc == 1


def disable_enc_fused(why: str) -> None:
    """Routes the word-level encoder to the per-layer launch chain for the rest of the process (a cluster wait gave up: the 8 workgroups of a
    sequence were not resident together — a co-tenant kernel holding LDS or compute units, a partitioned device)."""
    global ENC_FUSED, ENC_BWD_FUSED
    import warnings

    if ENC_FUSED or ENC_BWD_FUSED:
        warnings.warn(f"fused encoder disabled: {why}; the encoder runs on the per-layer launch chain (svsr_igemm_fwd / svsr_mha_fwd / "
                      f"svsr_add_ln_fwd ...) from here on", RuntimeWarning, stacklevel=2)
    ENC_FUSED = ENC_BWD_FUSED = False


def enc_fused_ok(D: int, H: int, inter: int, S: int) -> bool:
    return ENC_FUSED and D == 512 and H == 8 and inter == 2048 and 1 <= S <= 32


def enc_fwd(x0: torch.Tensor, layers: Sequence[dict], B: int, S: int, eps: float, seed: Optional[torch.Tensor], p_hidden: float, p_attn: float) -> None:
    """layers: per layer a dict of tensors under the field names of svsr_enc_layer (+ site_probs / site_ao / site_fo ints).
    Up to 8 layers per launch; deeper encoders take several (the last output of one call is the input of the next)."""
    dev = x0.device
    nbytes = int(_lib.load().svsr_enc_fwd_ws_bytes(B))
    key = (str(dev), _stream())
    ws = _ENC_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _SCRATCH_KEEP.append(ws)
        ws = _ENC_WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    x = x0
    for i0 in range(0, len(layers), 8):
        chunk = layers[i0: i0 + 8]
        arr = (_EncLayer * len(chunk))()
        for rec, q in zip(arr, chunk):
            for name, _ in _EncLayer._fields_[:25]:
                setattr(rec, name, q[name].data_ptr())
            rec.site_probs, rec.site_ao, rec.site_fo, rec.pad_ = int(q["site_probs"]), int(q["site_ao"]), int(q["site_fo"]), 0
        global _ENC_FUSED_USED
        _ENC_FUSED_USED = True
        _call("svsr_enc_fwd", _p(x), arr, len(chunk), B, S, float(eps), _p(seed), float(p_hidden), float(p_attn), _p(ws), nbytes, _stream(),
              label="k_enc_fwd", flops=float(len(chunk)) * 2.0 * B * S * (4 * 512 * 512 + 2 * 512 * 2048) + float(len(chunk)) * 4.0 * B * 8 * S * S * 64)
        x = chunk[-1]["xout"]


class _EncBwdLayer(ctypes.Structure):
    """svsr_enc_bwd_layer of include/syncvsr_hip.h"""
    _PTRS = ("w2t", "w1t", "wot", "wqkvt", "g1", "g2", "f", "x1", "ao", "xin", "z", "qkv", "probs", "m1", "r1", "m2", "r2",
             "ds2", "df", "dz", "dx1", "ds1", "dao", "dqkv", "dx", "part1", "part2")
    _fields_ = [(n, ctypes.c_void_p) for n in _PTRS] + \
               [("site_probs", ctypes.c_uint), ("site_ao", ctypes.c_uint), ("site_fo", ctypes.c_uint), ("pad_", ctypes.c_uint)]


ENC_BWD_FUSED = os.environ.get("SVSR_ENC_BWD_FUSED", "1") != "0"     # the whole encoder backward as one launch (False: 13 launches per layer)
_ENC_BWD_WS: dict = {}


def enc_bwd(dy: torch.Tensor, layers: Sequence[dict], B: int, S: int, seed: Optional[torch.Tensor], p_hidden: float, p_attn: float) -> None:
    """layers (forward order, at most 8): per layer a dict of tensors under the field names of svsr_enc_bwd_layer (+ the three site ints)."""
    if len(layers) > 8:
        raise ValueError("svsr_enc_bwd takes at most 8 layers per launch")
    dev = dy.device
    nbytes = int(_lib.load().svsr_enc_bwd_ws_bytes(B))
    key = (str(dev), _stream())
    ws = _ENC_BWD_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _SCRATCH_KEEP.append(ws)
        ws = _ENC_BWD_WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    arr = (_EncBwdLayer * len(layers))()
    for rec, q in zip(arr, layers):
        for name in _EncBwdLayer._PTRS:
            setattr(rec, name, q[name].data_ptr())
        rec.site_probs, rec.site_ao, rec.site_fo, rec.pad_ = int(q["site_probs"]), int(q["site_ao"]), int(q["site_fo"]), 0
    n = len(layers)
    global _ENC_FUSED_USED
    _ENC_FUSED_USED = True
    _call("svsr_enc_bwd", _p(dy), arr, n, B, S, _p(seed), float(p_hidden), float(p_attn), _p(ws), nbytes, _stream(),
          label="k_enc_bwd", flops=float(n) * 2.0 * B * S * (4 * 512 * 512 + 2 * 512 * 2048) + float(n) * 8.0 * B * 8 * S * S * 64)


# -- `type: x-transformers` encoder passes (csrc/xt.hip) ---------------------------------------------
def rmsnorm_fwd(x, g, D: int, eps: float = 1e-8):
    """x bf16 [R][ld] (pad columns zero) -> (y, inv [R])."""
    R, ld = x.shape
    y = torch.empty_like(x)
    inv = torch.empty(R, dtype=torch.float32, device=x.device)
    _call("svsr_rmsnorm_fwd", _p(x), _p(g), _p(y), _p(inv), R, D, ld, float(eps), _stream())
    return y, inv


def rmsnorm_bwd(dy, x, g, inv, dg, D: int, addend=None) -> torch.Tensor:
    R, ld = x.shape
    dx = torch.empty_like(x)
    part = scratch(_query("svsr_rmsnorm_bwd_rows", R)[0] * ld)
    _call("svsr_rmsnorm_bwd", _p(dy), _p(x), _p(g), _p(inv), _p(addend), _p(dx), _p(dg), _p(part), R, D, ld, _stream())
    return dx


def rotary_table(S: int, device, rot: int = 32, theta: float = 10000.0) -> torch.Tensor:
    """fp32 [S][rot]: cos then sin of position * theta^(-2j/rot), computed with the same torch ops as x-transformers' RotaryEmbedding."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, rot, 2).float() / rot))
    freqs = torch.einsum("i,j->ij", torch.arange(S).float(), inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).contiguous().to(device)


def rotary_(qkv: torch.Tensor, tab: torch.Tensor, S: int, heads_total: int, sign: int = 1) -> None:
    R, ld = qkv.shape
    assert tab.shape == (S, 32) and tab.dtype == torch.float32
    _call("svsr_rotary", _p(qkv), _p(tab), R, S, heads_total, ld, int(sign), _stream())


def geglu_fwd(u: torch.Tensor, I: int, ldy: int, drop=None) -> torch.Tensor:
    R, ldu = u.shape
    y = torch.empty((R, ldy), dtype=BF16, device=u.device)
    _call("svsr_geglu_fwd", _p(u), _p(y), R, I, ldu, ldy, *_drop(drop), _stream())
    return y


def geglu_bwd(dy: torch.Tensor, u: torch.Tensor, I: int, drop=None) -> torch.Tensor:
    R, ldu = u.shape
    du = torch.empty_like(u)
    _call("svsr_geglu_bwd", _p(dy), _p(u), _p(du), R, I, ldu, dy.shape[1], *_drop(drop), _stream())
    return du


def xt_embed_fwd(feats, wmask, cls, B: int, S: int, F: int, D: int, ld: int, drop=None) -> torch.Tensor:
    x0 = torch.empty((B * S, ld), dtype=BF16, device=feats.device)
    _call("svsr_xt_embed_fwd", _p(feats), _p(wmask), _p(cls), _p(x0), B, S, F, D, ld, *_drop(drop), _stream())
    return x0


def xt_embed_bwd(dx0, dcls, B: int, S: int, F: int, D: int, drop=None) -> torch.Tensor:
    dfeats = torch.empty((B * (S - 1), F), dtype=BF16, device=dx0.device)
    _call("svsr_xt_embed_bwd", _p(dx0), _p(dfeats), _p(dcls), B, S, F, D, dx0.shape[1], *_drop(drop), _stream())
    return dfeats


def bias_act_bwd(dy, z, db, *, R: int, N: int, n_valid: int, ld: int, relu: bool = False, gscale: float = 1.0, defer: Optional[list] = None) -> torch.Tensor:
    """db[:n_valid] += column sums of dz, where dz = dy * act'(z) if z is given (returned) else dy; act = GELU from the
    pre-activation z, or (relu=True) ReLU from the saved output z."""
    dz = torch.empty_like(dy) if z is not None else None
    rows = _query("svsr_bias_act_bwd_rows", R, N)[0] if db is not None else 0
    if defer is not None and rows:
        part = torch.empty(rows * N, dtype=torch.float32, device=dy.device)
        _call("svsr_bias_act_bwd_partials", _p(dy), _p(z), _p(dz), _p(db), R, N, n_valid, ld, 2 if relu else 1, float(gscale), _p(part), _stream())
        defer.append((_deferred_colsum(part, rows, N, db, n_valid), part))
        return dz if z is not None else dy
    part = scratch(rows * N) if rows else None
    _call("svsr_bias_act_bwd", _p(dy), _p(z), _p(dz), _p(db), R, N, n_valid, ld, 2 if relu else 1, float(gscale), _p(part), _stream())
    return dz if z is not None else dy


# --------------------------------------------------------------------------------------------------
# losses / metric / optimiser
# --------------------------------------------------------------------------------------------------
def ce_fwd(logits, ld: int, target_idx, target_prob, R: int, V: int, smoothing: float):
    dev = logits.device
    loss = torch.empty((), dtype=torch.float32, device=dev)
    lse = torch.empty(R, dtype=torch.float32, device=dev)
    ldt = 0 if target_prob is None else target_prob.shape[-1]
    _call("svsr_ce_fwd", _p(logits), int(logits.dtype == torch.float32), ld, _p(target_idx), _p(target_prob), ldt, R, V,
          float(smoothing), _p(loss), _p(lse), _p(scratch(R)), _stream())
    return loss, lse


def ce_bwd(logits, ld: int, target_idx, target_prob, R: int, V: int, smoothing: float, lse, gout, dlogits, ldo: int) -> None:
    ldt = 0 if target_prob is None else target_prob.shape[-1]
    _call("svsr_ce_bwd", _p(logits), int(logits.dtype == torch.float32), ld, _p(target_idx), _p(target_prob), ldt, R, V,
          float(smoothing), _p(lse), _p(gout), _p(dlogits), ldo, _stream())


def linear_ce_ok(R: int, K: int, G: int, V: int) -> bool:
    """True when the fused projection + cross-entropy kernels (csrc/audio_head.hip) take this shape."""
    return bool(_lib.load().svsr_linear_ce_ok(R, K, G, V))


def linear_ce_fwd(h: torch.Tensor, w16: torch.Tensor, bias, tok: torch.Tensor, R: int, K: int, G: int, V: int, seq=None):
    """-> (loss, lse [R*G]): mean cross-entropy of (h W^T + bias).reshape(-1, V) against tok, logits never stored (svsr_linear_ce_fwd).
    seq = (S, s0, T): row r reads hidden row (r // T) * S + s0 + r % T."""
    dev = h.device
    loss = torch.empty((), dtype=torch.float32, device=dev)
    lse = torch.empty(R * G, dtype=torch.float32, device=dev)
    S, s0, T = seq if seq is not None else (0, 0, 0)
    _call("svsr_linear_ce_fwd", _p(h), _p(w16), _p(bias), _p(tok), R, K, G, V, S, s0, T, _p(loss), _p(lse), _p(scratch(R * G)), _stream(),
          label="k_linear_ce", flops=2.0 * R * K * G * V)
    return loss, lse


def linear_ce_bwd(h: torch.Tensor, w16: torch.Tensor, bias, tok: torch.Tensor, R: int, K: int, G: int, V: int, lse: torch.Tensor, gout: torch.Tensor,
                  dlogits: torch.Tensor, seq=None) -> None:
    """dlogits [R][G*V] bf16 = gout / (R*G) * (softmax - onehot), the logits recomputed (svsr_linear_ce_bwd)."""
    S, s0, T = seq if seq is not None else (0, 0, 0)
    _call("svsr_linear_ce_bwd", _p(h), _p(w16), _p(bias), _p(tok), R, K, G, V, S, s0, T, _p(lse), _p(gout), _p(dlogits), _stream(),
          label="k_linear_ce", flops=2.0 * R * K * G * V)


def topk_acc(logits_f32, labels, soft_labels) -> torch.Tensor:
    B, C = logits_f32.shape
    out = torch.empty(2, dtype=torch.float32, device=logits_f32.device)
    _call("svsr_topk_acc", _p(logits_f32), _p(labels), _p(soft_labels), B, C, _p(out), _p(scratch(2 * B)), _stream())
    return out


def grad_sumsq(g: torch.Tensor, opt_state: torch.Tensor) -> None:
    _call("svsr_grad_sumsq", _p(g), g.numel(), _p(opt_state), _stream())


EARLY_SUMSQ = True      # host-side knob "early_sumsq": the clip's sum of squares over everything but the stem weight on the side stream beside the stem's weight gradient
SUMSQ_PARTS = 1024      # partial sums of squares in the optimiser's device state (loss_optim.hip OPT_PARTS)


def grad_sumsq_parts(g: torch.Tensor, start: int, n: int, opt_state: torch.Tensor, part0: int, nparts: int) -> None:
    """sum of squares of g[start : start + n] into the partial sums [part0, part0 + nparts) (start a multiple of 4 floats)"""
    assert start % 4 == 0
    _call("svsr_grad_sumsq_parts", g.data_ptr() + 4 * start, n, _p(opt_state), part0, nparts, _stream())


def adamw_step(p, g, m, v, shadow, decay_end: int, lr: float, betas, eps: float, weight_decay: float, max_norm: float, warmup: int,
               total_steps: int, opt_state) -> None:
    _call("svsr_adamw_step", _p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), decay_end, lr, betas[0], betas[1], eps, weight_decay,
          max_norm, warmup, total_steps, _p(opt_state), _stream())


def adamw_range(p, g, m, v, shadow, lo: int, hi: int, decay_end: int, lr: float, betas, eps: float, weight_decay: float, max_norm: float, warmup: int,
                total_steps: int, opt_state, advance: bool) -> None:
    """AdamW over elements [lo, hi) of the flat buffers (decay_end: absolute end of the weight-decayed region)."""
    n = hi - lo
    dec = min(max(decay_end - lo, 0), n)
    _call("svsr_adamw_range", p.data_ptr() + 4 * lo, g.data_ptr() + 4 * lo, m.data_ptr() + 4 * lo, v.data_ptr() + 4 * lo, shadow.data_ptr() + 2 * lo, n, dec,
          lr, betas[0], betas[1], eps, weight_decay, max_norm, warmup, total_steps, _p(opt_state), int(advance), _stream())


def transpose_shadows_range(w16, dst, table: torch.Tensor, first: int, count: int) -> None:
    """Entries [first, first + count) of the transposed-shadow table (from the bf16 shadow)."""
    if count > 0:
        _call("svsr_transpose_bf16_multi", _p(w16), _p(dst), table.data_ptr() + 32 * first, count, _stream())


def clip_prep(frames_u8: torch.Tensor, params: torch.Tensor, H: int, W: int, mean: float = 0.421, std: float = 0.165) -> torch.Tensor:
    """uint8 clips [B, T, Hs, Ws] + int32 params [B, 5] = (top, left, h, w, flip) -> fp32 [B, 1, T, H, W] (svsr_clip_prep)."""
    B, T, Hs, Ws = frames_u8.shape
    assert frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous() and params.dtype == torch.int32 and params.shape == (B, 5)
    out = torch.empty((B, 1, T, H, W), dtype=torch.float32, device=frames_u8.device)
    _call("svsr_clip_prep", _p(frames_u8), _p(params), _p(out), B, T, Hs, Ws, H, W, float(mean), float(std), _stream())
    return out


def cast_bf16(src: torch.Tensor, dst: torch.Tensor) -> None:
    _call("svsr_cast_bf16", _p(src), _p(dst), src.numel(), _stream())


def transpose_cast_multi(src, dst, table: torch.Tensor, n_entries: int) -> None:
    _call("svsr_transpose_cast_multi", _p(src), _p(dst), _p(table), n_entries, _stream())


TRANSPOSE_FROM_BF16 = True      # tuning knob "transpose_from_bf16": refresh the transposed shadows from the bf16 shadow (half the bytes read)


def transpose_shadows(flat, w16, dst, table: torch.Tensor, n_entries: int) -> None:
    """Transposed bf16 shadows of the 2-D weights from the (fresh) bf16 shadow w16, or from the fp32 buffer: identical results."""
    if TRANSPOSE_FROM_BF16:
        _call("svsr_transpose_bf16_multi", _p(w16), _p(dst), _p(table), n_entries, _stream())
    else:
        transpose_cast_multi(flat, dst, table, n_entries)


# --------------------------------------------------------------------------------------------------
# LRS: attention, Conformer convolution module, CTC, decoder embedding, label-smoothing loss
# --------------------------------------------------------------------------------------------------
def probs_pitch(Lk: int) -> int:
    return (Lk + 7) // 8 * 8


MHA_FLASH = os.environ.get("SVSR_MHA_FLASH", "1") != "0"     # sentence-level attention on the streamed-key kernels (mha_flash.h); False: the per-tile kernels


class MhaLse:
    """What the flash forward keeps for the backward instead of the probabilities: the rows' log-sum-exp, the output and the masks it ran with."""
    __slots__ = ("lse", "ctx", "klen", "causal", "ldp")

    def __init__(self, lse, ctx, klen, causal, ldp):
        self.lse, self.ctx, self.klen, self.causal, self.ldp = lse, ctx, klen, causal, ldp


def mha_fwd(q, q_pitch: int, k, v, kv_pitch: int, *, B: int, H: int, Lq: int, Lk: int, pe=None, bias_u=None, bias_v=None, klen=None,
            causal: bool = False, drop=None, flash: bool = False):
    """-> (ctx [B*Lq, H*64] bf16, probs [B*H, Lq, ldp] bf16).  q/k/v are views into (fused) projection outputs.
    flash=True (and MHA_FLASH): svsr_mha_flash_fwd — the second result is an MhaLse record (no probabilities are stored); hand it to mha_bwd."""
    ldp = probs_pitch(Lk)
    ctx = torch.empty((B * Lq, H * 64), dtype=BF16, device=q.device)
    flops = 2.0 * B * H * Lq * Lk * 64 * (3 if pe is not None else 2)
    if flash and MHA_FLASH:
        lse = torch.empty((B * H, Lq), dtype=torch.float32, device=q.device)
        _call("svsr_mha_flash_fwd", _p(q), q_pitch, _p(k), _p(v), kv_pitch, _p(pe), 0 if pe is None else pe.stride(0), _p(bias_u), _p(bias_v),
              _p(klen), int(causal), B, H, 64, Lq, Lk, ldp, 0.125, _p(ctx), H * 64, _p(lse), *_drop(drop), _stream(), label="k_mhaf_fwd", flops=flops)
        return ctx, MhaLse(lse, ctx, klen, bool(causal), ldp)
    probs = torch.empty((B * H, Lq, ldp), dtype=BF16, device=q.device)
    _call("svsr_mha_fwd", _p(q), q_pitch, _p(k), _p(v), kv_pitch, _p(pe), 0 if pe is None else pe.stride(0), _p(bias_u), _p(bias_v),
          _p(klen), int(causal), B, H, 64, Lq, Lk, ldp, 0.125, _p(ctx), H * 64, _p(probs), *_drop(drop), _stream(),
          label="k_mha_fwd", flops=flops)
    return ctx, probs


def mha_pe_transpose(pe: torch.Tensor, H: int, Lq: int) -> torch.Tensor:
    """The transposed position table the query pass of the flash backward reads (mha_bwd(..., pet=...)): it depends on pe alone."""
    nws = int(_lib.load().svsr_mha_flash_ws_bytes(H, Lq))
    ws = torch.empty(nws, dtype=torch.uint8, device=pe.device)
    _call("svsr_mha_pe_transpose", _p(pe), pe.stride(0), H, Lq, _p(ws), nws, _stream())
    return ws


def mha_bwd(dctx, q, q_pitch: int, k, v, kv_pitch: int, probs, *, B: int, H: int, Lq: int, Lk: int, dq, dq_pitch: int, dk, dv,
            dkv_pitch: int, pe=None, bias_u=None, bias_v=None, drop=None, pe_later: bool = False, pet: Optional[torch.Tensor] = None):
    """Writes dq/dk/dv (views with the given pitches).  Relative-position form returns (dq_ac, dq_bd, dpe) as well.
    probs: the forward's second result (the probabilities, or the MhaLse record of a flash forward).
    pe_later (flash + relative positions): a fourth result — (fn, keep): fn() issues the position-table pass that fills dpe on the stream
    current WHEN IT IS CALLED (only the weight gradient of linear_pos reads dpe: the model hands fn to its side stream); keep = the tensors
    that pass reads."""
    rel = pe is not None
    D = H * 64
    dq_ac = torch.empty((B * Lq, D), dtype=BF16, device=q.device) if rel else None
    dq_bd = torch.empty((B * Lq, D), dtype=BF16, device=q.device) if rel else None
    dpe = torch.empty((2 * Lq - 1, D), dtype=BF16, device=q.device) if rel else None
    pe_part = torch.empty((B, 2 * Lq - 1, D), dtype=torch.float32, device=q.device) if rel else None
    flops = 2.0 * B * H * Lq * Lk * 64 * (7 if rel else 4)
    if isinstance(probs, MhaLse):
        rec = probs
        ldp = rec.ldp
        pbuf = torch.empty((B * H, Lq, ldp), dtype=BF16, device=q.device)       # workspace: P and dS of the query pass for the key / table passes
        ds = torch.empty_like(pbuf)
        nws = int(_lib.load().svsr_mha_flash_ws_bytes(H, Lq)) if rel else 0
        have_pet = 4 if (rel and pet is not None) else 0          # (mha_pe_transpose made the table ahead of time)
        ws = pet if have_pet else (torch.empty(nws, dtype=torch.uint8, device=q.device) if nws else None)
        def launch(parts: int, fl: float) -> None:
            _call("svsr_mha_flash_bwd_parts", _p(dctx), dctx.stride(0), _p(rec.ctx), rec.ctx.stride(0), _p(rec.lse), _p(q), q_pitch, _p(k), _p(v), kv_pitch,
                  _p(pe), 0 if pe is None else pe.stride(0), _p(bias_u), _p(bias_v), _p(rec.klen), int(rec.causal), _p(pbuf), _p(ds), B, H, 64, Lq, Lk, ldp,
                  0.125, _p(dq), dq_pitch, _p(dq_ac), _p(dq_bd), D, _p(dk), _p(dv), dkv_pitch, _p(dpe), D, _p(pe_part), _p(ws), nws, *_drop(drop), parts,
                  _stream(), label="k_mhaf_bwd" if parts & 1 else "k_mha_bwd_pe", flops=fl)

        if rel and pe_later:
            launch(1 | have_pet, flops * 6.0 / 7.0)
            return dq_ac, dq_bd, dpe, (lambda: launch(2, flops / 7.0), (ds, q, dpe, pe_part))
        launch(3 | have_pet, flops)
        return dq_ac, dq_bd, dpe
    ldp = probs.shape[-1]
    ds = torch.empty_like(probs)
    _call("svsr_mha_bwd", _p(dctx), dctx.stride(0), _p(q), q_pitch, _p(k), _p(v), kv_pitch, _p(pe), 0 if pe is None else pe.stride(0),
          _p(bias_u), _p(bias_v), _p(probs), _p(ds), B, H, 64, Lq, Lk, ldp, 0.125, _p(dq), dq_pitch, _p(dq_ac), _p(dq_bd), D,
          _p(dk), _p(dv), dkv_pitch, _p(dpe), D, _p(pe_part), *_drop(drop), _stream(), label="k_mha_bwd", flops=flops)
    return dq_ac, dq_bd, dpe


def glu_dwconv_fwd(u, w, bias, want_stats: bool, B: int, T: int, D: int, K: int):
    """-> (c [B*T, D] bf16, BatchNorm1d partials (buffer, rows) or None)"""
    c = torch.empty((B * T, D), dtype=BF16, device=u.device)
    stats = st = None
    if want_stats:
        rows = _query("svsr_glu_dwconv_fwd_stat_rows", B, T)[0]
        stats = scratch(rows * 2 * D)
        st = (stats, rows)
    _call("svsr_glu_dwconv_fwd", _p(u), _p(w), _p(bias), _p(c), _p(stats), B, T, D, K, _stream())
    return c, st


DW_SPLITS = 96        # workgroups per 64-channel block of svsr_glu_dwconv_bwd: one (clip, 32-frame tile) each at B = 16, T <= 192 (16 splits left a quarter of the CUs idle with several tiles in a row each)


def glu_dwconv_bwd(dc, u, w, dw, dbias, B: int, T: int, D: int, K: int, reduce_later: bool = False):
    """-> du; reduce_later: -> (du, (fn, keep)) — fn() issues the fixed-order sum of the partial rows into dw / dbias on the stream current when
    it is called (parameter gradients: the model hands fn to its side stream), keep = the partial rows."""
    du = torch.empty_like(u)
    part = torch.empty(DW_SPLITS * D * (K + 1), dtype=torch.float32, device=u.device)

    def launch(parts: int) -> None:
        _call("svsr_glu_dwconv_bwd_parts", _p(dc), _p(u), _p(w), _p(du), _p(dw), _p(dbias), _p(part), DW_SPLITS, B, T, D, K, parts, _stream())

    if reduce_later:
        launch(1)
        return du, (lambda: launch(2), (part,))
    launch(3)
    return du


def lrs_targets(label: torch.Tensor, odim: int, ignore_id: int, eos: int):
    """label int64 [B, L] on the device, ignore_id padding (anywhere in a row: dropped) -> (labels [B, L] with -1, ys_in [B, L+1], ys_out [B, L+1]):
    add_sos_eos in one launch.  A label outside [1, odim) becomes eos and sets the sticky word lrs_target_errors() reads."""
    B, L = label.shape
    label = label.contiguous()
    labels = torch.empty_like(label)
    ys_in = torch.empty((B, L + 1), dtype=torch.int64, device=label.device)
    ys_out = torch.empty_like(ys_in)
    _call("svsr_lrs_targets", _p(label), B, L, odim, ignore_id, eos, _p(labels), _p(ys_in), _p(ys_out), _stream())
    return labels, ys_in, ys_out


def lrs_target_errors(reset: bool = True) -> bool:
    """True if a svsr_lrs_targets launch since the last reset met a label outside [1, odim) (synchronises the device)."""
    rc = int(_lib.load().svsr_lrs_target_errors(1 if reset else 0))
    if rc < 0:
        raise _lib.SvsrError("svsr_lrs_target_errors could not read the error word (an earlier HIP error is pending on this device)")
    return rc == 1


def ctc_fwd(logits, ld: int, labels, ilen, B: int, T: int, V: int):
    """logits fp32 [B*T, ld]; labels int64 [B, Lmax] (-1 padded); ilen int32 [B] -> (loss 0-d, state for ctc_grad)."""
    Lmax = labels.shape[1]
    dev = logits.device
    loss = torch.empty((), dtype=torch.float32, device=dev)
    lse = torch.empty(B * T, dtype=torch.float32, device=dev)
    ab = torch.empty((B, T, 2 * Lmax + 1), dtype=torch.float32, device=dev)
    nll = torch.empty(B, dtype=torch.float32, device=dev)
    _call("svsr_ctc_fwd", _p(logits), ld, _p(labels), Lmax, _p(ilen), B, T, V, _p(lse), _p(ab), _p(nll), _p(loss), _stream())
    return loss, (lse, ab, nll)


def ctc_grad(logits, ld: int, labels, ilen, B: int, T: int, V: int, state, gout, ldo: int) -> torch.Tensor:
    lse, ab, nll = state
    dz = torch.empty((B * T, ldo), dtype=BF16, device=logits.device)
    _call("svsr_ctc_grad", _p(logits), ld, _p(labels), labels.shape[1], _p(ilen), B, T, V, _p(lse), _p(ab), _p(nll), _p(gout), _p(dz), ldo,
          _stream())
    return dz


def ctc_prefix_score(logp: torch.Tensor, r_prev: torch.Tensor, last: torch.Tensor, ids: Optional[torch.Tensor], out_len: int, blank: int,
                     eos: int) -> tuple[torch.Tensor, torch.Tensor]:
    """logp fp32 [T, V], r_prev fp32 [n, T, 2], last int64 [n], ids int64 [n, S] or None (all V labels)
    -> (r_new fp32 [n, S, T, 2], psi fp32 [n, S]).  See svsr_ctc_prefix_score."""
    T, V = logp.shape
    n = r_prev.shape[0]
    assert logp.dtype == torch.float32 and logp.is_contiguous() and r_prev.dtype == torch.float32 and r_prev.shape == (n, T, 2)
    assert r_prev.is_contiguous() and last.dtype == torch.int64 and last.numel() == n
    assert ids is None or (ids.dtype == torch.int64 and ids.is_contiguous() and ids.shape[0] == n)
    S = V if ids is None else ids.shape[1]
    r_new = torch.empty((n, S, T, 2), dtype=torch.float32, device=logp.device)
    psi = torch.empty((n, S), dtype=torch.float32, device=logp.device)
    _call("svsr_ctc_prefix_score", _p(logp), V, _p(r_prev), _p(last), _p(ids), _p(r_new), _p(psi), T, V, n, S, int(out_len), int(blank), int(eos),
          _stream())
    return r_new, psi


def embed_pos_fwd(tok, emb, pe, L: int, D: int, scale: float) -> torch.Tensor:
    R = tok.numel()
    x = torch.empty((R, D), dtype=BF16, device=emb.device)
    _call("svsr_embed_pos_fwd", _p(tok), _p(emb), _p(pe), _p(x), R, L, D, float(scale), _stream())
    return x


def embed_pos_bwd(tok, dx, demb, D: int, scale: float) -> None:
    _call("svsr_embed_pos_bwd", _p(tok), _p(dx), _p(demb), tok.numel(), D, float(scale), _stream())


def ls_loss_fwd(logits, ld: int, target, R: int, V: int, smoothing: float, inv_denom: float):
    dev = logits.device
    loss = torch.empty((), dtype=torch.float32, device=dev)
    lse = torch.empty(R, dtype=torch.float32, device=dev)
    counts = torch.empty(2, dtype=torch.float32, device=dev)
    _call("svsr_ls_loss_fwd", _p(logits), ld, _p(target), R, V, float(smoothing), float(inv_denom), _p(loss), _p(lse), _p(counts),
          _p(scratch(3 * R)), _stream())
    return loss, lse, counts


def ls_loss_bwd(logits, ld: int, target, R: int, V: int, smoothing: float, inv_denom: float, lse, gout, ldo: int) -> torch.Tensor:
    dz = torch.empty((R, ldo), dtype=BF16, device=logits.device)
    _call("svsr_ls_loss_bwd", _p(logits), ld, _p(target), R, V, float(smoothing), float(inv_denom), _p(lse), _p(gout), _p(dz), ldo, _stream())
    return dz


def scale_bf16(x: torch.Tensor, alpha: float, drop=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = alpha * dropout(x) (x contiguous; the dropout element index is the position in x)."""
    y = torch.empty_like(x) if out is None else out
    _call("svsr_scale_bf16", _p(x), _p(y), x.numel(), float(alpha), *_drop(drop), _stream())
    return y
