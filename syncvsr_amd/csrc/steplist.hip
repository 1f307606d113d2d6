// Native step enqueuer (host code only): records the launch list of one training step once and re-issues it with ONE call per
// step, keeping the stream assignment of every launch (main stream / weight-gradient side stream) and the cross-stream
// dependencies — what a captured HIP graph loses and what the Python loop pays ~13 us per launch for.
//
// It stands where Lightning's per-step loop stands in the reference (LRW/video/src/lightning.py:194-202 driven by
// pl.Trainer.fit, LRW/video/src/train.py:23-45): one host call per optimisation step.
//
// A list is a sequence of
//   CALL    one svsr_* entry point of this library with its arguments frozen (pointers, sizes, the hipStream_t),
//   WAIT    stream `waiter` waits for everything enqueued so far on stream `signaller` (hipEventRecord + hipStreamWaitEvent),
//   MEMSET  hipMemsetAsync,
//   BREAK   end of a segment: svsr_steplist_run(list, k) issues segment k and returns, so the host can put a collective
//           (torch.distributed / RCCL) between two segments.
// Everything a CALL points to must stay alive and in place while the list exists: device buffers (the recorder on the Python side
// keeps every tensor of the recorded step), host arrays (plan meta records, tap tables: cached for the life of the process).
// Nothing here launches a kernel of its own; results are bit-identical to the eager step by construction.
#include <stdlib.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "common.h"
#include "../../include/syncvsr_hip.h"

namespace {

typedef int64_t Slot;        // an argument as 64 raw bits: integers sign-extended, floats as the bits of a double, pointers as they are

template <typename T> struct FromSlot;
template <> struct FromSlot<int> { static int get(Slot s) { return (int)s; } };
template <> struct FromSlot<unsigned> { static unsigned get(Slot s) { return (unsigned)s; } };
template <> struct FromSlot<int64_t> { static int64_t get(Slot s) { return s; } };
template <> struct FromSlot<float> { static float get(Slot s) { double d; memcpy(&d, &s, sizeof d); return (float)d; } };
template <typename P> struct FromSlot<P*> { static P* get(Slot s) { return reinterpret_cast<P*>((uintptr_t)s); } };

template <typename... A, size_t... I>
int invoke(int (*fn)(A...), const Slot* s, std::index_sequence<I...>) { return fn(FromSlot<A>::get(s[I])...); }

template <typename... A>
int thunk_n(int (*)(A...)) { return (int)sizeof...(A); }

typedef int (*Thunk)(const Slot*);
struct Entry { Thunk call; int nargs; };

template <typename F, F fn> struct Bind;
template <typename... A, int (*fn)(A...)>
struct Bind<int (*)(A...), fn> {
    static int call(const Slot* s) { return invoke(fn, s, std::index_sequence_for<A...>{}); }
    static constexpr int nargs = (int)sizeof...(A);
};

#define SVSR_REG(f) {#f, Entry{&Bind<decltype(&f), &f>::call, Bind<decltype(&f), &f>::nargs}}

// every entry point that enqueues work on a stream (tests/test_host_cpu.py checks this table against the header)
const std::unordered_map<std::string, Entry>& registry() {
    static const std::unordered_map<std::string, Entry> r = {
        SVSR_REG(svsr_colsum_rows), SVSR_REG(svsr_colsum_rows_multi), SVSR_REG(svsr_igemm_fwd), SVSR_REG(svsr_igemm_wgrad), SVSR_REG(svsr_conv3x3_c64),
        SVSR_REG(svsr_igemm_dgrad_bn), SVSR_REG(svsr_igemm_dgrad_relu), SVSR_REG(svsr_conv3x3_c64_dgrad_bn), SVSR_REG(svsr_bn_bwd_from_stats),
        SVSR_REG(svsr_conv3x3_wgrad), SVSR_REG(svsr_stem_conv_fwd), SVSR_REG(svsr_stem_conv_wgrad), SVSR_REG(svsr_stem_bwd_wgrad), SVSR_REG(svsr_bn_finalize),
        SVSR_REG(svsr_bn_eval_prepare), SVSR_REG(svsr_bn_act_fwd), SVSR_REG(svsr_bn_act_bwd), SVSR_REG(svsr_stem_bn_act_pool_fwd),
        SVSR_REG(svsr_stem_bn_act_pool_bwd), SVSR_REG(svsr_avgpool_fwd), SVSR_REG(svsr_avgpool_bwd), SVSR_REG(svsr_add_ln_fwd),
        SVSR_REG(svsr_add_ln_bwd), SVSR_REG(svsr_embed_ln_fwd), SVSR_REG(svsr_embed_bwd_scatter), SVSR_REG(svsr_rmsnorm_fwd),
        SVSR_REG(svsr_rmsnorm_bwd), SVSR_REG(svsr_rotary), SVSR_REG(svsr_geglu_fwd), SVSR_REG(svsr_geglu_bwd), SVSR_REG(svsr_xt_embed_fwd),
        SVSR_REG(svsr_xt_embed_bwd), SVSR_REG(svsr_bias_act_bwd), SVSR_REG(svsr_ce_fwd), SVSR_REG(svsr_ce_bwd), SVSR_REG(svsr_linear_ce_fwd), SVSR_REG(svsr_linear_ce_bwd), SVSR_REG(svsr_topk_acc),
        SVSR_REG(svsr_grad_sumsq), SVSR_REG(svsr_grad_sumsq_parts), SVSR_REG(svsr_adamw_step), SVSR_REG(svsr_adamw_range), SVSR_REG(svsr_cast_bf16), SVSR_REG(svsr_transpose_cast_multi),
        SVSR_REG(svsr_transpose_bf16_multi), SVSR_REG(svsr_fill_f32), SVSR_REG(svsr_clip_prep), SVSR_REG(svsr_mha_fwd), SVSR_REG(svsr_mha_bwd), SVSR_REG(svsr_mha_flash_fwd), SVSR_REG(svsr_mha_flash_bwd), SVSR_REG(svsr_mha_flash_bwd_parts), SVSR_REG(svsr_mha_pe_transpose),
        SVSR_REG(svsr_glu_dwconv_fwd), SVSR_REG(svsr_glu_dwconv_bwd), SVSR_REG(svsr_glu_dwconv_bwd_parts), SVSR_REG(svsr_ctc_fwd), SVSR_REG(svsr_ctc_grad),
        SVSR_REG(svsr_ctc_prefix_score), SVSR_REG(svsr_lrs_targets), SVSR_REG(svsr_embed_pos_fwd), SVSR_REG(svsr_embed_pos_bwd), SVSR_REG(svsr_ls_loss_fwd),
        SVSR_REG(svsr_ls_loss_bwd), SVSR_REG(svsr_scale_bf16), SVSR_REG(svsr_word_add), SVSR_REG(svsr_lincomb2), SVSR_REG(svsr_igemm_wgrad_group), SVSR_REG(svsr_enc_fwd), SVSR_REG(svsr_enc_bwd), SVSR_REG(svsr_lincomb3_ratio), SVSR_REG(svsr_add_ln_bwd_partials), SVSR_REG(svsr_add_ln_bwd_branch), SVSR_REG(svsr_bias_act_bwd_partials),
    };
    return r;
}

enum { OP_CALL = 0, OP_WAIT, OP_MEMSET, OP_BREAK };
constexpr int MAX_ARGS = 48;

struct Op {
    int kind;
    Thunk call;
    int nargs;
    Slot args[MAX_ARGS];
    hipEvent_t ev;            // WAIT
    hipStream_t a, b;         // WAIT: a waits for b;  MEMSET: a = stream
    void* ptr; int value; size_t bytes;
};

struct StepList {
    std::vector<Op> ops;
    std::vector<size_t> seg_begin{0};
    std::vector<hipEvent_t> events;
    std::string last_error;
};

}  // namespace

extern "C" {

/* 1 if `name` is a launch entry point the step list can record, else 0 */
int svsr_steplist_knows(const char* name) { return name != nullptr && registry().count(name) ? 1 : 0; }

void* svsr_steplist_create(void) { return new StepList(); }

int svsr_steplist_destroy(void* list) {
    StepList* l = static_cast<StepList*>(list);
    if (l == nullptr) return SVSR_ERR_ARG;
    for (hipEvent_t e : l->events) (void)hipEventDestroy(e);
    delete l;
    return SVSR_OK;
}

int svsr_steplist_push_call(void* list, const char* name, const int64_t* slots, int nslots) {
    StepList* l = static_cast<StepList*>(list);
    if (l == nullptr || name == nullptr || (slots == nullptr && nslots > 0)) return SVSR_ERR_ARG;
    auto it = registry().find(name);
    if (it == registry().end() || it->second.nargs != nslots || nslots > MAX_ARGS) return SVSR_ERR_ARG;
    Op op{};
    op.kind = OP_CALL; op.call = it->second.call; op.nargs = nslots;
    for (int i = 0; i < nslots; ++i) op.args[i] = slots[i];
    l->ops.push_back(op);
    return SVSR_OK;
}

/* Flags of the events behind a cross-stream WAIT.  Both streams are on one device: the signaller's kernels release to agent scope when
 * they end, which is all a kernel of the waiting stream needs; the system-scope fence an event performs by default when it is recorded
 * (for the host and for peer devices) costs the SIGNALLING stream ~2.4 us per record (scripts/probes/handover_probe.hip: a dependent
 * main-stream chain with a hand-over after every kernel, 52.9 -> 50.5 us per kernel) — 35 records per word-level step, ~250 per
 * sentence-level step.  Nothing the host or RCCL reads is ordered by these events (the host synchronises the stream, the reducer's
 * comm stream waits through torch's own events).  SVSR_EVENT_SYSTEM_FENCE=1 restores the default. */
static unsigned wait_event_flags() {
    static const unsigned flags = [] {
        const char* v = getenv("SVSR_EVENT_SYSTEM_FENCE");
        return (unsigned)hipEventDisableTiming | ((v != nullptr && v[0] == '1') ? 0u : (unsigned)hipEventDisableSystemFence);
    }();
    return flags;
}

int svsr_steplist_push_wait(void* list, hipStream_t waiter, hipStream_t signaller) {
    StepList* l = static_cast<StepList*>(list);
    if (l == nullptr) return SVSR_ERR_ARG;
    Op op{};
    op.kind = OP_WAIT; op.a = waiter; op.b = signaller;
    hipError_t e = hipEventCreateWithFlags(&op.ev, wait_event_flags());
    if (e != hipSuccess) return (int)e;
    l->events.push_back(op.ev);
    l->ops.push_back(op);
    return SVSR_OK;
}

int svsr_steplist_push_memset(void* list, void* ptr, int value, int64_t bytes, hipStream_t stream) {
    StepList* l = static_cast<StepList*>(list);
    if (l == nullptr || ptr == nullptr || bytes < 0) return SVSR_ERR_ARG;
    Op op{};
    op.kind = OP_MEMSET; op.ptr = ptr; op.value = value; op.bytes = (size_t)bytes; op.a = stream;
    l->ops.push_back(op);
    return SVSR_OK;
}

/* closes the current segment; returns the index of the segment that starts here */
int svsr_steplist_push_break(void* list) {
    StepList* l = static_cast<StepList*>(list);
    if (l == nullptr) return -SVSR_ERR_ARG;
    l->seg_begin.push_back(l->ops.size());
    return (int)l->seg_begin.size() - 1;
}

int svsr_steplist_segments(void* list) {
    StepList* l = static_cast<StepList*>(list);
    return l == nullptr ? 0 : (int)l->seg_begin.size();
}

int64_t svsr_steplist_size(void* list) {
    StepList* l = static_cast<StepList*>(list);
    return l == nullptr ? 0 : (int64_t)l->ops.size();
}

/* issues segment `segment` (all of them for segment < 0); returns 0 or the first non-zero code, with the failing op's index in *failed */
int svsr_steplist_run(void* list, int segment, int* failed) {
    StepList* l = static_cast<StepList*>(list);
    if (l == nullptr || segment >= (int)l->seg_begin.size()) return SVSR_ERR_ARG;
    const size_t lo = segment < 0 ? 0 : l->seg_begin[segment];
    const size_t hi = (segment < 0 || segment + 1 == (int)l->seg_begin.size()) ? l->ops.size() : l->seg_begin[segment + 1];
    for (size_t i = lo; i < hi; ++i) {
        const Op& op = l->ops[i];
        int rc = 0;
        switch (op.kind) {
            case OP_CALL: rc = op.call(op.args); break;
            case OP_WAIT: {
                hipError_t e = hipEventRecord(op.ev, op.b);
                if (e == hipSuccess) e = hipStreamWaitEvent(op.a, op.ev, 0);
                rc = (int)e;
                break;
            }
            case OP_MEMSET: rc = (int)hipMemsetAsync(op.ptr, op.value, op.bytes, op.a); break;
            default: break;
        }
        if (rc != 0) {
            if (failed != nullptr) *failed = (int)i;
            return rc;
        }
    }
    return SVSR_OK;
}

/* `waiter` waits for everything enqueued so far on `signaller` (eager twin of the list's WAIT op) */
int svsr_stream_wait(hipStream_t waiter, hipStream_t signaller) {
    static hipEvent_t ring[64];
    static unsigned next = 0;
    static bool made = false;
    if (!made) {
        for (hipEvent_t& e : ring) {
            hipError_t rc = hipEventCreateWithFlags(&e, wait_event_flags());
            if (rc != hipSuccess) return (int)rc;
        }
        made = true;
    }
    hipEvent_t ev = ring[next++ & 63u];
    hipError_t e = hipEventRecord(ev, signaller);
    if (e == hipSuccess) e = hipStreamWaitEvent(waiter, ev, 0);
    return (int)e;
}

int svsr_memset_async(void* ptr, int value, int64_t bytes, hipStream_t stream) {
    if (ptr == nullptr || bytes < 0) return SVSR_ERR_ARG;
    return (int)hipMemsetAsync(ptr, value, (size_t)bytes, stream);
}

}  // extern "C"
