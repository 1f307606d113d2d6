// Fused forward of the 512-wide HF-BERT encoder layers (gfx950): ALL layers of the encoder in ONE launch.
//
// Replaces, for the word-level model's `type: huggingface` encoder (reference LRW/video/src/lightning.py:92,152-156 -> HF
// BertLayer: BertSelfAttention, BertSelfOutput, BertIntermediate, BertOutput), the chain of seven launches per layer
// (qkv GEMM, attention, output GEMM, add+LayerNorm, GELU GEMM, output GEMM, add+LayerNorm).  At 960 rows that chain is pure
// latency: 42 launches of 5-15 us for 36 GFLOP, with the chip almost idle.
//
// Rows of different sequences never meet inside an encoder layer, so a sequence (S <= 32 rows) is an independent pipeline through
// all layers.  A CLUSTER of H = 8 workgroups owns one sequence; workgroup h of the cluster owns head h and the h-th eighth of the
// columns of every GEMM of the layer:
//   P1  q_h | k_h | v_h = X W_qkv[h]^T (+bias) -> qkv;  S = q k^T / 8, softmax -> probs;  ctx_h = dropout(P) v      (no exchange)
//   P2  ao[:, h] = dropout(ctx W_o[h]^T + b)                                   needs every head's ctx      -> cluster barrier 1
//   P3  x1 = LN(ao + X) (every workgroup, redundantly);  z_h = x1 W_1[h]^T + b, hg_h = gelu(z_h)           -> cluster barrier 2
//   P4  f[:, h] = dropout(hg W_2[h]^T + b)                                      needs all of hg             -> cluster barrier 3
//   P5  X' = LN(f + x1) (redundantly) = the next layer's input                                             -> cluster barrier 4
// The activations a layer keeps for the backward (qkv, probs, ctx, ao, x1, z, hg, f, X', LayerNorm statistics) are written exactly
// where the unfused path writes them, with the same arithmetic (bf16 rounding points, dropout element indices, LayerNorm in fp32
// with one wave per row), so the hand-written backward and the parity tests are unchanged.
//
// Exchange between the workgroups of a cluster follows the placement-independent recipe of the programming guide (section 6,
// guideline 16): payload written with write-through (sc1) stores, every storing wave drains its stores, ONE lane bumps the
// cluster's arrival counter with a relaxed agent-scope atomic; consumers poll that one word (relaxed, with s_sleep) and read the
// payload with sc1 loads (LDS-DMA with the sc1 bit for GEMM operands).  Counters are zeroed by a memset node in front of the launch
// and count monotonically through the 4 x layers barriers of a launch; spins are bounded (the error word is set and the launch
// finishes with garbage instead of hanging).  Weights are read-only and stream through a 3-deep LDS ring by LDS-DMA.
// Grid: 8 workgroups per sequence, all of which must be resident together: at most 32 sequences per launch (256 CUs); the host
// splits larger batches into several launches (sequences are independent).
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "../../include/syncvsr_hip.h"

namespace {

constexpr int ED = 512, EH = 8, EI = 2048, TR = 32;            // width, heads (= workgroups per cluster), FFN width, rows per tile
constexpr int IH = EI / EH;                                    // FFN1 columns per workgroup
#define EF_SWZ(row, chunk) ((row) * 64 + (((chunk) ^ (((row) >> 1) & 7)) << 3))       // [rows][64 k] bf16 block, 16-byte chunks XOR-swizzled

// LDS map (bytes): the A operand with the whole K = 512 resident (also attention scratch and epilogue staging) + a 4-deep ring of 32 KiB
// slots (three tiles = 96 KiB in flight: the weight stream is bound by bytes in flight / L2 latency, ~55 GB/s per CU with two)
constexpr int BUFA = 0;                         // bf16 [8 kb][32][64] (32 KiB)
constexpr int RING = 32768;
constexpr int SLOT = 32768, NSLOT = 4;
constexpr int LDS_TOTAL = RING + NSLOT * SLOT;  // 160 KiB
static_assert(LDS_TOTAL <= 160 * 1024, "one workgroup per CU");

struct EncArgs {
    const bf16_t* x0;            // [R][512] input of the first layer
    const svsr_enc_layer* Ls;    // DEVICE copy of the layer records (a kernel-argument array indexed with a run-time value would be moved to scratch)
    int layers, S, seq0, nseq;   // sequences seq0 .. seq0 + nseq - 1 of the batch
    float eps;
    const unsigned* seed; unsigned th_hidden, th_attn; float sc_hidden, sc_attn;
    unsigned* cnt;               // [nseq] arrival counters (zero at launch)
    unsigned* err;               // set to 1 when a bounded spin gave up
    unsigned* host_flag;         // the same word in pinned HOST memory (svsr_enc_gave_up_peek reads it without a synchronisation), or null
    unsigned spin_limit;         // polls of a cluster wait before it gives up (2^20 ~ 1 s; svsr_debug_enc_spin_limit lowers it for the fall-back test)
    unsigned long long* trace;   // debug (svsr_debug_enc_trace): s_memtime stamps of workgroup 0 at the phase boundaries, or null
};

__device__ __forceinline__ void glds16(const void* src, void* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
__device__ __forceinline__ void glds16_sc1(const void* src, void* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 16);
}
// 16 bytes of exchanged payload: write-through stores / L1-bypassing loads (two 8-byte relaxed agent-scope accesses each)
__device__ __forceinline__ void st16_sc1(bf16_t* p, const u32x4& v) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    __hip_atomic_store(q, ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, ((unsigned long long)v.w << 32) | v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u32x4 ld16_sc1(const bf16_t* p) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u32x4 v; v.x = (unsigned)a; v.y = (unsigned)(a >> 32); v.z = (unsigned)b; v.w = (unsigned)(b >> 32);
    return v;
}

#define EF_WAIT_VM(n) SVSR_WAIT_VM(n)      /* (vmcnt(0) in the SVSR_SYNC_DEBUG build: common.h) */
#define EF_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// arrival: every wave's payload stores have left (write-through), then one lane counts the workgroup in
__device__ __forceinline__ void cluster_signal(unsigned* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (the same when weight tiles of the next phase are already in flight: vmcnt(0) waits for them too — the price of one counter for loads and stores)
__device__ __forceinline__ void cluster_signal_keep_dma(unsigned* cnt) { cluster_signal(cnt); }
// A wait that gives up (the 8 workgroups of a sequence were not resident together: a CU-masked or partitioned device, a co-tenant holding
// LDS) must not pass silently: the launch's error word is set (cleared by the next launch's memset: the tests read it), the STICKY word
// g_enc_gave_up is set (never cleared by a launch: svsr_enc_gave_up reads it; the same word in pinned host memory lets engine.TrainStep see it
// WITHOUT a synchronisation and fall back to the per-layer launch chain: svsr_enc_gave_up_peek), and the workgroup poisons its
// slice of the launch's final output with NaN (enc_poison below), so the loss of this step (forward) or the gradient norm and every
// parameter after it (backward) turn NaN without a host synchronisation.
__device__ unsigned g_enc_gave_up = 0;
__device__ __forceinline__ void cluster_wait(unsigned* cnt, unsigned target, unsigned* err, unsigned* host_flag, unsigned spin_limit, bool& gave_up) {
    if (threadIdx.x == 0 && !gave_up) {          // (a workgroup that gave up once does not wait again: its launch is lost anyway, it only has to end)
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > spin_limit) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&g_enc_gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (host_flag != nullptr) __hip_atomic_store(host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                gave_up = true;
                break;
            }
        }
    }
    __syncthreads();
}
// end of a launch: thread 0 of a workgroup whose wait gave up overwrites the first 8 columns of its slice of row 0 of the final output with NaN
__device__ __forceinline__ void enc_poison(bool gave_up, bf16_t* dst) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && gave_up) {
        u32x4 nan; nan.x = nan.y = nan.z = nan.w = 0x7fc07fc0u;
        st16_sc1(dst, nan);
    }
}

__device__ __forceinline__ bf16x8 lds_frag(const bf16_t* blk, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(blk + EF_SWZ(row, chunk));
}

// eight rows of 512 columns per wave (lane owns columns lane*8 .. +7 of each): y = LayerNorm(a + r) * gamma + beta with the arithmetic of
// k_add_ln_fwd (bert.hip) — the eight rows' shuffle reductions are interleaved, each row's own order of additions is unchanged
__device__ __forceinline__ void ln8(const u32x4 (&ra)[8], const u32x4 (&rr)[8], const float* gamma, const float* beta, int lane, float eps,
                                    u32x4 (&out)[8], float (&mu)[8], float (&rs)[8]) {
    float v[8][8], s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float fa[8], fr[8];
        unpack8(ra[i], fa);
        unpack8(rr[i], fr);
        s[i] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[i][k] = fa[k] + fr[k]; s[i] += v[i][k]; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = wave_sum(s[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        mu[i] = s[i] / (float)ED;
        s[i] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mu[i]; s[i] = __builtin_fmaf(d, d, s[i]); }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = wave_sum(s[i]);
    float g8[8], b8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { g8[k] = gamma[lane * 8 + k]; b8[k] = beta[lane * 8 + k]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        rs[i] = rsqrtf(s[i] / (float)ED + eps);
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = __builtin_fmaf((v[i][k] - mu[i]) * rs[i], g8[k], b8[k]);
        out[i] = pack8(o);
    }
}

// accumulator block (32 x 32, MFMA layout) -> fp32 [32][pitch] staging at column col0
__device__ __forceinline__ void acc_to_stage(const f32x16& a, float* st, int pitch, int col0, int lane) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * pitch + col0 + (lane & 31)] = a[r];
}

}  // namespace

// Accumulation order = the launch chain's (tests/test_gpu_enc_fused.py compares bit for bit): every output element of the qkv, attention-output
// and intermediate GEMMs is ONE accumulator walking k upwards (k_igemm_fwd_glds<64,64,4,1>), the output GEMM (K = 2048) is two halves of
// K in two accumulators added at the end (k_igemm_fwd_glds<64,64,4,2>), softmax sums and the P.V contraction follow k_mha_fwd4.
__global__ __launch_bounds__(256, 1) void k_enc_fwd(const EncArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* bufA = reinterpret_cast<bf16_t*>(smem + BUFA);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // cluster / head of this workgroup: with a multiple of 8 sequences the 8 workgroups of a cluster share blockIdx % 8 (observed to be
    // the XCD: the exchanged tensors stay in one L2's neighbourhood; with a HEAD per XCD instead — weights L2-resident — the weight
    // stream was no faster and every exchange slower: 458 vs 319 us).  Speed only.
    int c, h;
    {
        const int i = blockIdx.x;
        if ((p.nseq & 7) == 0) { const int j = i >> 3; h = j & 7; c = (i & 7) + 8 * (j >> 3); }
        else { c = i >> 3; h = i & 7; }
    }
    const int S = p.S;
    const long row0 = (long)(p.seq0 + c) * S;                       // first row of this sequence
    const int bh = (p.seq0 + c) * EH + h;
    unsigned* cnt = p.cnt + c;
    unsigned arrivals = 0;                                          // barrier target so far
    bool gave_up = false;                                           // (thread 0's copy counts)
    int tix = 0;
    auto stamp = [&]() {
        if (p.trace != nullptr && blockIdx.x == 0 && tid == 0 && tix < 500) p.trace[tix] = __builtin_amdgcn_s_memtime();
        ++tix;
    };
    int fix = 0;
    auto fstamp = [&](int l_) {          // fine-grained stamps of layer 1 only, at p.trace[200 ..]
        if (p.trace != nullptr && blockIdx.x == 0 && tid == 0 && l_ == 1 && fix < 60) p.trace[200 + fix++] = __builtin_amdgcn_s_memtime();
    };
    stamp();
    const int slot8 = tid & 7, r32 = tid >> 3;                      // DMA: lane writes 16-byte chunk slot8 of row r32 (+ 32 i) ...
    const int csw = slot8 ^ ((r32 >> 1) & 7);                       // ... which holds global chunk csw (swizzle on the source)
    const int rsrc = r32 < S ? r32 : S - 1;                         // padding rows of the tile repeat the last row (results never stored)
    const bool drop_on = p.seed != nullptr;
    auto ring = [&](int slot) { return reinterpret_cast<bf16_t*>(smem + RING + slot * SLOT); };

    // ---- layer input X -> bufA (A layout) ---------------------------------------------------------------------------------
    {
        const bf16_t* src = p.x0 + (row0 + rsrc) * ED + csw * 8;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) glds16(src + kb * 64, bufA + kb * 2048 + wave * 8 * 64);
    }

    for (int l = 0; l < p.layers; ++l) {
        const svsr_enc_layer L = p.Ls[l];
        const bf16_t* xin = l == 0 ? p.x0 : reinterpret_cast<const bf16_t*>(p.Ls[l - 1].xout);
        const unsigned key_pr = drop_on ? svsr_mix(p.seed[0] * 0x9E3779B9u + L.site_probs * 0x7F4A7C15u + 0x165667B1u) : 0u;
        const unsigned key_ao = drop_on ? svsr_mix(p.seed[0] * 0x9E3779B9u + L.site_ao * 0x7F4A7C15u + 0x165667B1u) : 0u;
        const unsigned key_fo = drop_on ? svsr_mix(p.seed[0] * 0x9E3779B9u + L.site_fo * 0x7F4A7C15u + 0x165667B1u) : 0u;

        // =========================== P1: q | k | v of head h, attention =====================================================
        {
            const bf16_t* W = reinterpret_cast<const bf16_t*>(L.wqkv);
            // B tile of step s: rows {q, k, v} x 64 of head h, k-chunk s: 6 DMA pieces per thread
            auto stageB = [&](int s, int slot) {
                bf16_t* dst = ring(slot);
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const long wrow = (long)(i >> 1) * ED + h * 64 + (i & 1) * 32 + r32;
                    glds16(W + wrow * ED + s * 64 + csw * 8, dst + (i * 32 + wave * 8) * 64);
                }
            };
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            // (the queue may still hold the previous layer's stores: counted waits below must only ever see DMA pieces behind them)
            if (l > 0) { EF_WAIT_VM(0); EF_BARRIER(); }      // ... and every wave has read its rows of f / x1 out of slots 0 and 1 (P5)
            stageB(0, 0); stageB(1, 1); stageB(2, 2);
            for (int s = 0; s < 8; ++s) {
                if (s < 6) EF_WAIT_VM(12); else if (s == 6) EF_WAIT_VM(6); else EF_WAIT_VM(0);
                EF_BARRIER();
                if (s + 3 < 8) stageB(s + 3, (s + 3) & 3);
                if (wave < 3) {
                    const bf16_t* B = ring(s & 3) + wave * 64 * 64;
                    const bf16_t* A = bufA + s * 2048;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const int ch = ks * 2 + (lane >> 5);
                        const bf16x8 fa = lds_frag(A, lane & 31, ch);
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, lds_frag(B, j * 32 + (lane & 31), ch), acc[j], 0, 0, 0);
                    }
                }
            }
            stamp();                                         // [1] P1 GEMM done
            __syncthreads();                                 // everybody is done with X in bufA and with the ring
            // ring slot 0: fp32 staging [3][32][64]; bufA: attention operands
            float* stage = reinterpret_cast<float*>(ring(0));
            bf16_t* Qa = bufA;                               // [32][64] A layout
            bf16_t* Kb = bufA + 2048;                        // [32 keys][64] B layout
            bf16_t* Vt = bufA + 4096;                        // [64 d][64: keys 0..31] B layout of V^T
            bf16_t* Pa = bufA + 8192;                        // [32][64: keys 0..31] A layout of dropout(P)
            if (wave < 3) {
                float* st = stage + wave * 2048;
                acc_to_stage(acc[0], st, 64, 0, lane);
                acc_to_stage(acc[1], st, 64, 32, lane);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const float* bias = L.bqkv + wave * ED + h * 64;
                const int c8 = lane & 7;
                float b8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) b8[k] = bias[c8 * 8 + k];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = (lane >> 3) + 8 * i;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(st + row * 64 + c8 * 8), hi = *reinterpret_cast<const f32x4*>(st + row * 64 + c8 * 8 + 4);
                    float v[8] = {lo[0] + b8[0], lo[1] + b8[1], lo[2] + b8[2], lo[3] + b8[3], hi[0] + b8[4], hi[1] + b8[5], hi[2] + b8[6], hi[3] + b8[7]};
                    const u32x4 pk = pack8(v);
                    if (row < S) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(L.qkv) + (row0 + row) * (3 * ED) + wave * ED + h * 64 + c8 * 8) = pk;
                    if (wave == 0) *reinterpret_cast<u32x4*>(Qa + EF_SWZ(row, c8)) = pk;
                    else if (wave == 1) *reinterpret_cast<u32x4*>(Kb + EF_SWZ(row, c8)) = pk;
                    else {
                        const unsigned w4[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) Vt[EF_SWZ(c8 * 8 + e, row >> 3) + (row & 7)] = (bf16_t)((w4[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
                    }
                }
            }
            // the next phase's first weight tiles (slots 1..3; slot 0 holds the staging): requested before the attention runs
            const bf16_t* Wo = reinterpret_cast<const bf16_t*>(L.wo) + (long)(h * 64) * ED;
            auto stageWo = [&](int s, int slot) {      // 64 rows x 256 k (4 chunks of 64): 8 DMA pieces per thread
                bf16_t* dst = ring(slot);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int sub = i >> 1, row = (i & 1) * 32 + r32;
                    glds16(Wo + (long)row * ED + s * 256 + sub * 64 + csw * 8, dst + sub * 4096 + ((i & 1) * 32 + wave * 8) * 64);
                }
            };
            fstamp(l);                                       // f0: qkv epilogue done
            stageWo(0, 1); stageWo(1, 2);
            fstamp(l);                                       // f1: Wo issued
            EF_BARRIER();                                    // (LDS visibility only: the weight tiles stay in flight across it)
            {   // scores of the whole 32 x 32 block on every wave (four MFMAs); wave w finishes rows 8w .. 8w+7 (accumulator registers 4w .. 4w+3)
                f32x16 sc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int ch = ks * 2 + (lane >> 5);
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(Qa, lane & 31, ch), lds_frag(Kb, lane & 31, ch), sc, 0, 0, 0);
                }
                fstamp(l);                                   // f2: scores done
                const int j = lane & 31;
                const int ldp = (S + 7) & ~7;
                float sv[4], e[4], sum[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x = sc[0];
                    // (accumulator register 4*wave + q, selected with compile-time indices)
                    if (wave == 0) x = sc[q]; else if (wave == 1) x = sc[4 + q]; else if (wave == 2) x = sc[8 + q]; else x = sc[12 + q];
                    sv[q] = j < S ? x * 0.125f : -INFINITY;
                }
                // row maximum over the 32 lanes of a half (any tree: a maximum is exact), then the row sum in k_mha_fwd4's order: eight partial
                // sums ((e[p] + e[p+8]) + e[p+16]) + e[p+24] — valid in lanes 0..7 of each half — a butterfly over the eight (its l^4 step pairs
                // quads of equal values: row_half_mirror pairs the same quads), broadcast from lane 0 of the half
                float m[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x = sv[q];
                    x = fmaxf(x, dpp_xor1(x)); x = fmaxf(x, dpp_xor2(x)); x = fmaxf(x, dpp_half_mirror(x)); x = fmaxf(x, dpp_mirror(x));
                    m[q] = fmaxf(x, swz_xor16(x));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) e[q] = __expf(sv[q] - m[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float u = swz_xor16(e[q]);
                    float a = ((e[q] + dpp_ror8(e[q])) + u) + dpp_ror8(u);
                    a += dpp_xor1(a);
                    a += dpp_xor2(a);
                    a += dpp_half_mirror(a);
                    sum[q] = swz_lane0_of_32(a);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 8 * wave + q + 4 * (lane >> 5);
                    const float inv = sum[q] > 0.f ? 1.f / sum[q] : 0.f;
                    float pr = (j < S && inv > 0.f) ? e[q] * inv : 0.f;
                    const long idx = ((long)bh * S + i) * ldp + j;
                    if (i < S && j < ldp) reinterpret_cast<bf16_t*>(L.probs)[idx] = f2bf(pr);
                    if (drop_on) pr = drop_keep(key_pr, p.th_attn, (unsigned)idx) ? pr * p.sc_attn : 0.f;
                    Pa[EF_SWZ(i, j >> 3) + (j & 7)] = f2bf(pr);
                }
                fstamp(l);                                   // f3: softmax + stores done
            }
            EF_BARRIER();
            fstamp(l);                                       // f4: barrier
            if (wave < 2) {       // ctx[:, wave*32 ..] = P' V: one accumulator per 16-key slice, added at the end (k_mha_fwd4's two k-slice waves)
                f32x16 c0, c1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(Pa, lane & 31, lane >> 5), lds_frag(Vt, wave * 32 + (lane & 31), lane >> 5), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(Pa, lane & 31, 2 + (lane >> 5)), lds_frag(Vt, wave * 32 + (lane & 31), 2 + (lane >> 5)), c1, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) c0[r] += c1[r];
                float* st = stage + wave * 1024;                                         // [32][32] per wave (the q / k staging is consumed)
                acc_to_stage(c0, st, 32, 0, lane);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int c4 = lane & 3;                                                  // 8 columns each: 4 groups per 32-column row
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = (lane >> 2) + 16 * i;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(st + row * 32 + c4 * 8), hi = *reinterpret_cast<const f32x4*>(st + row * 32 + c4 * 8 + 4);
                    const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if (row < S) st16_sc1(reinterpret_cast<bf16_t*>(L.ctx) + (row0 + row) * ED + h * 64 + wave * 32 + c4 * 8, pack8(v));
                }
            }
            stamp();                                         // [2] attention done
            cluster_signal(cnt);
            stamp();                                         // [3] signalled
        }

        // =========================== P2: ao[:, h*64 ..] = dropout(ctx W_o^T + b) =============================================
        {
            arrivals += EH;
            cluster_wait(cnt, arrivals, p.err, p.host_flag, p.spin_limit, gave_up);
            stamp();                                         // [4] barrier 1 passed
            {   // every head's ctx -> bufA
                const bf16_t* src = reinterpret_cast<const bf16_t*>(L.ctx) + (row0 + rsrc) * ED + csw * 8;
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) glds16_sc1(src + kb * 64, bufA + kb * 2048 + wave * 8 * 64);
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            EF_WAIT_VM(0);
            EF_BARRIER();
            // the next phase's first weight tiles (W_1: 256 rows x 64 k per step) into the free slots 3 and 0
            const bf16_t* W1 = reinterpret_cast<const bf16_t*>(L.w1) + (long)(h * IH) * ED;
            auto stageW1 = [&](int s, int slot) {
                bf16_t* dst = ring(slot);
#pragma unroll
                for (int i = 0; i < 8; ++i) glds16(W1 + (long)(i * 32 + r32) * ED + s * 64 + csw * 8, dst + (i * 32 + wave * 8) * 64);
            };
            stageW1(0, 3); stageW1(1, 0);
            if (wave < 2) {       // one accumulator per 32-column block walks k upwards
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16_t* B = ring(1 + s);
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            const int ch = ks * 2 + (lane >> 5);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(bufA + (s * 4 + sub) * 2048, lane & 31, ch),
                                                                           lds_frag(B + sub * 4096, wave * 32 + (lane & 31), ch), acc, 0, 0, 0);
                        }
                }
            }
            EF_BARRIER();                                    // ctx in bufA is consumed: bufA becomes the fp32 staging [32][64]
            float* stage = reinterpret_cast<float*>(bufA);
            if (wave < 2) acc_to_stage(acc, stage, 64, wave * 32, lane);
            EF_BARRIER();
            {   // all 256 threads: row tid / 8, eight columns: bias, dropout, write-through store
                const int row = tid >> 3, c8 = tid & 7, n = h * 64 + c8 * 8;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * 64 + c8 * 8), hi = *reinterpret_cast<const f32x4*>(stage + row * 64 + c8 * 8 + 4);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const long off = (row0 + row) * ED + n;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[k] += L.bo[n + k];
                    if (drop_on) v[k] = drop_keep(key_ao, p.th_hidden, (unsigned)(off + k)) ? v[k] * p.sc_hidden : 0.f;
                }
                if (row < S) st16_sc1(reinterpret_cast<bf16_t*>(L.ao) + off, pack8(v));
            }
            stamp();                                         // [5] P2 done
            cluster_signal_keep_dma(cnt);
        }

        // =========================== P3: x1 = LN(ao + X);  z_h | hg_h = gelu(x1 W_1^T + b) ===================================
        {
            const bf16_t* W1 = reinterpret_cast<const bf16_t*>(L.w1) + (long)(h * IH) * ED;
            auto stageW1 = [&](int s, int slot) {
                bf16_t* dst = ring(slot);
#pragma unroll
                for (int i = 0; i < 8; ++i) glds16(W1 + (long)(i * 32 + r32) * ED + s * 64 + csw * 8, dst + (i * 32 + wave * 8) * 64);
            };
            // the layer input X is complete since barrier 1 (every workgroup signalled it after its previous LN2): requested ahead of the wait
            {
                const bf16_t* sx = xin + (row0 + rsrc) * ED + csw * 8;
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) glds16_sc1(sx + kb * 64, ring(2) + kb * 2048 + wave * 8 * 64);
            }
            arrivals += EH;
            cluster_wait(cnt, arrivals, p.err, p.host_flag, p.spin_limit, gave_up);
            stamp();                                         // [6] barrier 2 passed
            // LayerNorm of rows wave*8 .. +7 (lane owns 8 columns of each) -> bufA (A layout); this workgroup's column slice -> x1.
            // ao and X arrive by LDS-DMA (sc1) in slots 1 and 2 (free: W_o is consumed; W_1's first tiles sit in 3 and 0): one round trip
            {
                const bf16_t* sa = reinterpret_cast<const bf16_t*>(L.ao) + (row0 + rsrc) * ED + csw * 8;
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) glds16_sc1(sa + kb * 64, ring(1) + kb * 2048 + wave * 8 * 64);
                fstamp(l);                                   // f5: LN1 DMA issued
                EF_WAIT_VM(0);
                EF_BARRIER();
                fstamp(l);                                   // f6: LN1 DMA landed
                u32x4 ra[8], rr[8], o8[8];
                float mu[8], rs[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = wave * 8 + i;
                    ra[i] = *reinterpret_cast<const u32x4*>(ring(1) + (lane >> 3) * 2048 + EF_SWZ(row, lane & 7));
                    rr[i] = *reinterpret_cast<const u32x4*>(ring(2) + (lane >> 3) * 2048 + EF_SWZ(row, lane & 7));
                }
                ln8(ra, rr, L.g1, L.be1, lane, p.eps, o8, mu, rs);
                fstamp(l);                                   // f7: ln8 done
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = wave * 8 + i;
                    *reinterpret_cast<u32x4*>(bufA + (lane >> 3) * 2048 + EF_SWZ(row, lane & 7)) = o8[i];
                    if (row < S) {
                        if ((lane >> 3) == h) st16_sc1(reinterpret_cast<bf16_t*>(L.x1) + (row0 + row) * ED + lane * 8, o8[i]);
                        if (h == 0 && lane == 0) { L.m1[row0 + row] = mu[i]; L.r1[row0 + row] = rs[i]; }
                    }
                }
            }
            stamp();                                         // [7] LN1 done
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            EF_WAIT_VM(0);                                   // tiles 0 and 1 (requested in P2) and this phase's own stores
            EF_BARRIER();                                    // every wave has read its rows of ao / X out of slots 1 and 2
            stageW1(2, 1);
            for (int s = 0; s < 8; ++s) {                    // tile s lives in slot (s + 3) & 3
                if (s == 0) {} else if (s < 6) EF_WAIT_VM(16); else if (s == 6) EF_WAIT_VM(8); else EF_WAIT_VM(0);
                EF_BARRIER();
                if (s + 3 < 8) stageW1(s + 3, (s + 6) & 3);
                const bf16_t* B = ring((s + 3) & 3) + wave * 64 * 64;
                const bf16_t* A = bufA + s * 2048;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int ch = ks * 2 + (lane >> 5);
                    const bf16x8 fa = lds_frag(A, lane & 31, ch);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, lds_frag(B, j * 32 + (lane & 31), ch), acc[j], 0, 0, 0);
                }
            }
            __syncthreads();                                 // x1 in bufA and the ring are consumed
            // the next phase's weight half-tiles (W_2) do not depend on the other workgroups: requested before the epilogue
            const bf16_t* W2 = reinterpret_cast<const bf16_t*>(L.w2) + (long)(h * 64) * EI;
            auto stageW2 = [&](int s, int slot) {           // chunks s (K group 0) and 16 + s (K group 1): 64 rows x 64 k each, 4 pieces per thread
                bf16_t* dst = ring(slot);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int g = i >> 1, row = (i & 1) * 32 + r32;
                    glds16(W2 + (long)row * EI + (g * 16 + s) * 64 + csw * 8, dst + g * 4096 + ((i & 1) * 32 + wave * 8) * 64);
                }
            };
            stageW2(0, 0); stageW2(1, 1); stageW2(2, 2);
            {
                float* st = reinterpret_cast<float*>(bufA) + wave * 2048;               // fp32 [4][32][64] = 32 KiB = all of bufA
                acc_to_stage(acc[0], st, 64, 0, lane);
                acc_to_stage(acc[1], st, 64, 32, lane);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int c8 = lane & 7, n = h * IH + wave * 64 + c8 * 8;
                float b8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) b8[k] = L.b1[n + k];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = (lane >> 3) + 8 * i;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(st + row * 64 + c8 * 8), hi = *reinterpret_cast<const f32x4*>(st + row * 64 + c8 * 8 + 4);
                    float v[8] = {lo[0] + b8[0], lo[1] + b8[1], lo[2] + b8[2], lo[3] + b8[3], hi[0] + b8[4], hi[1] + b8[5], hi[2] + b8[6], hi[3] + b8[7]};
                    const long off = (row0 + row) * EI + n;
                    if (row < S) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(L.z) + off) = pack8(v);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = gelu_erf(v[k]);
                    if (row < S) st16_sc1(reinterpret_cast<bf16_t*>(L.hg) + off, pack8(v));
                }
            }
            stamp();                                         // [8] P3 done
            cluster_signal_keep_dma(cnt);
        }

        // =========================== P4: f[:, h*64 ..] = dropout(hg W_2^T + b) ==============================================
        {
            const bf16_t* W2 = reinterpret_cast<const bf16_t*>(L.w2) + (long)(h * 64) * EI;
            const bf16_t* HG = reinterpret_cast<const bf16_t*>(L.hg) + (row0 + rsrc) * EI + csw * 8;
            auto stageW2 = [&](int s, int slot) {
                bf16_t* dst = ring(slot);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int g = i >> 1, row = (i & 1) * 32 + r32;
                    glds16(W2 + (long)row * EI + (g * 16 + s) * 64 + csw * 8, dst + g * 4096 + ((i & 1) * 32 + wave * 8) * 64);
                }
            };
            auto stageHG = [&](int s, int slot) {           // hg rows, chunks s and 16 + s: 2 pieces per thread (sc1)
                bf16_t* dst = ring(slot) + 8192;
#pragma unroll
                for (int g = 0; g < 2; ++g) glds16_sc1(HG + (g * 16 + s) * 64, dst + g * 2048 + wave * 8 * 64);
            };
            arrivals += EH;
            cluster_wait(cnt, arrivals, p.err, p.host_flag, p.spin_limit, gave_up);
            stamp();                                         // [9] barrier 3 passed
            EF_WAIT_VM(0);                                   // the three prefetched weight half-tiles (and this wave's own stores)
            stageHG(0, 0); stageHG(1, 1); stageHG(2, 2);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int g = wave >> 1, jb = wave & 1;          // K group, 32-column block
            for (int s = 0; s < 16; ++s) {
                // outstanding behind tile s: s = 0: the hg pieces of tiles 1, 2 (4); s = 1: hg of tile 2 + tile 3 (8); then tiles s+1, s+2 (12); the tail drains
                if (s == 0) EF_WAIT_VM(4); else if (s == 1) EF_WAIT_VM(8); else if (s < 14) EF_WAIT_VM(12); else if (s == 14) EF_WAIT_VM(6); else EF_WAIT_VM(0);
                EF_BARRIER();
                if (s + 3 < 16) { stageW2(s + 3, (s + 3) & 3); stageHG(s + 3, (s + 3) & 3); }
                const bf16_t* T = ring(s & 3);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int ch = ks * 2 + (lane >> 5);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(T + 8192 + g * 2048, lane & 31, ch), lds_frag(T + g * 4096, jb * 32 + (lane & 31), ch), acc, 0, 0, 0);
                }
            }
            __syncthreads();
            float* stage = reinterpret_cast<float*>(bufA);   // fp32 [2 K groups][32][64]
            acc_to_stage(acc, stage + g * 2048, 64, jb * 32, lane);
            __syncthreads();
            {
                const int row = tid >> 3, c8 = tid & 7, n = h * 64 + c8 * 8;
                const f32x4 lo0 = *reinterpret_cast<const f32x4*>(stage + row * 64 + c8 * 8), hi0 = *reinterpret_cast<const f32x4*>(stage + row * 64 + c8 * 8 + 4);
                const f32x4 lo1 = *reinterpret_cast<const f32x4*>(stage + 2048 + row * 64 + c8 * 8), hi1 = *reinterpret_cast<const f32x4*>(stage + 2048 + row * 64 + c8 * 8 + 4);
                float v[8] = {lo0[0] + lo1[0], lo0[1] + lo1[1], lo0[2] + lo1[2], lo0[3] + lo1[3], hi0[0] + hi1[0], hi0[1] + hi1[1], hi0[2] + hi1[2], hi0[3] + hi1[3]};
                const long off = (row0 + row) * ED + n;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[k] += L.b2[n + k];
                    if (drop_on) v[k] = drop_keep(key_fo, p.th_hidden, (unsigned)(off + k)) ? v[k] * p.sc_hidden : 0.f;
                }
                if (row < S) st16_sc1(reinterpret_cast<bf16_t*>(L.f) + off, pack8(v));
            }
            stamp();                                         // [10] P4 done
            cluster_signal(cnt);
        }

        // =========================== P5: X' = LN(f + x1) -> bufA, column slice -> xout =======================================
        {
            {   // x1 is complete since barrier 3: requested ahead of the wait (the whole ring is idle here)
                const bf16_t* sx = reinterpret_cast<const bf16_t*>(L.x1) + (row0 + rsrc) * ED + csw * 8;
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) glds16_sc1(sx + kb * 64, ring(1) + kb * 2048 + wave * 8 * 64);
            }
            arrivals += EH;
            cluster_wait(cnt, arrivals, p.err, p.host_flag, p.spin_limit, gave_up);
            stamp();                                         // [11] barrier 4 passed
            {   // f by LDS-DMA (sc1) into slot 0
                const bf16_t* sa = reinterpret_cast<const bf16_t*>(L.f) + (row0 + rsrc) * ED + csw * 8;
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) glds16_sc1(sa + kb * 64, ring(0) + kb * 2048 + wave * 8 * 64);
                EF_WAIT_VM(0);
                EF_BARRIER();
            }
            u32x4 ra[8], rr[8], o8[8];
            float mu[8], rs[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = wave * 8 + i;
                ra[i] = *reinterpret_cast<const u32x4*>(ring(0) + (lane >> 3) * 2048 + EF_SWZ(row, lane & 7));
                rr[i] = *reinterpret_cast<const u32x4*>(ring(1) + (lane >> 3) * 2048 + EF_SWZ(row, lane & 7));
            }
            ln8(ra, rr, L.g2, L.be2, lane, p.eps, o8, mu, rs);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = wave * 8 + i;
                *reinterpret_cast<u32x4*>(bufA + (lane >> 3) * 2048 + EF_SWZ(row, lane & 7)) = o8[i];
                if (row < S) {
                    if ((lane >> 3) == h) st16_sc1(reinterpret_cast<bf16_t*>(L.xout) + (row0 + row) * ED + lane * 8, o8[i]);
                    if (h == 0 && lane == 0) { L.m2[row0 + row] = mu[i]; L.r2[row0 + row] = rs[i]; }
                }
            }
            stamp();                                         // [12] LN2 done
            // No barrier here: X' (write-through) is read by the other workgroups only in the next layer's P3, two barriers away, and
            // every tensor of this layer that they may still be reading (f, x1) is never written again.
        }
    }
    enc_poison(gave_up, reinterpret_cast<bf16_t*>(p.Ls[p.layers - 1].xout) + row0 * ED + h * 64);
}


// =================================================================================================================================
// Fused BACKWARD of the same layers (k_enc_bwd): gradient of the last layer's output -> gradient of the first layer's input, and every
// tensor the weight-gradient launches read, in ONE launch.  Replaces 13 launches per layer (two add+LayerNorm backward passes, four
// data-gradient GEMMs, the GELU backward pass, two attention-backward launches and their dropout re-scalings; reference: autograd of
// HF BertLayer as reached from LRW/video/src/lightning.py:92,152-156), a dependent chain of 5-15 us launches at 960 rows.
// The same cluster of 8 workgroups per sequence; workgroup h owns head h and the h-th eighth of every GEMM's output columns:
//   B1  ds2 = LayerNorm2 backward of the incoming gradient (every workgroup, all rows), df = dropout-mask(ds2)      (redundant, no exchange)
//   B2  dz[:, h] = (df W_2)[:, h-th 256] * gelu'(z)                                                                  -> cluster barrier 1
//   B4  dx1[:, h] = dz W_1 + ds2                                   needs all of dz                                   -> cluster barrier 2
//   B5  ds1 = LayerNorm1 backward of dx1, dao = dropout-mask(ds1)   (redundant)
//   B6  dctx_h = (dao W_o)[:, head h]                               (head-local: stays in LDS)
//   B7  attention backward of head h: dP = dctx V^T, dS = P o (dP' - rowsum(P o dP')) / 8, dq = dS K, dk = dS^T q, dv = P'^T dctx   -> cluster barrier 3
//   B8  dx[:, h] = dqkv W_qkv + ds1                                 needs every head's dq | dk | dv                  -> cluster barrier 4
// Rounding points are the launch chain's: every tensor that crosses a launch boundary there (ds2, df, dhg, dz, dx1, ds1, dao, dctx, dS,
// dq | dk | dv, dx) is rounded to bf16 here as well, the GEMMs with K >= 1536 add two K halves, dropout masks are regenerated from the same
// (site, element index) pairs.  Parameter gradients are NOT formed here: df, dz, dao, dqkv (with hg, x1, ctx, x of the forward) feed the
// grouped weight-gradient launch, and the LayerNorm gamma / beta sums leave as one partial row per sequence (part1 / part2, reduced by
// svsr_colsum_rows in a fixed order).
// =================================================================================================================================
namespace {

struct EncBwdArgs {
    const bf16_t* dy;                 // [R][512] gradient of the last layer's output
    const svsr_enc_bwd_layer* Ls;     // DEVICE copy of the layer records, forward order
    int layers, S, seq0, nseq;
    const unsigned* seed; unsigned th_hidden, th_attn; float sc_hidden, sc_attn;
    unsigned* cnt; unsigned* err; unsigned* host_flag; unsigned spin_limit;
    unsigned long long* trace;
};

// LayerNorm backward of the 8 rows wave*8 .. +7 (lane owns columns lane*8 .. +7), k_add_ln_bwd's arithmetic:
//   xhat = (a + r - mean) * rstd, gd = dy * gamma, ds = rstd * (gd - mean(gd) - xhat * mean(gd * xhat))
// dy / a / r sit in three ring slots in the A layout.  ds (bf16) -> this workgroup's column slice of ds_out; dd = dropout-mask(ds) -> bufA
// (A operand of the GEMM that follows) and, where it is a tensor of its own, its column slice of dd_out.  ag / ab: this lane's column sums
// of dy * xhat and dy over the rows < S.
__device__ __forceinline__ void ln_bwd8(const bf16_t* sDy, const bf16_t* sA, const bf16_t* sR, const float (&g8)[8], const float (&mu8)[8], const float (&rs8)[8],
                                        long row0, int S, int wave, int lane, int h, bool drop_on, unsigned key, unsigned th, float sc,
                                        bf16_t* bufA, bf16_t* ds_out, bf16_t* dd_out, float (&ag)[8], float (&ab)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { ag[k] = 0.f; ab[k] = 0.f; }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float xh[4][8], gd[4][8], m1[4], m2[4], rs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 8 + half * 4 + i;
            const float mu = mu8[half * 4 + i];
            rs[i] = rs8[half * 4 + i];
            const int o = (lane >> 3) * 2048 + EF_SWZ(row, lane & 7);
            float fa[8], fr[8], fd[8];
            unpack8(*reinterpret_cast<const u32x4*>(sA + o), fa);
            unpack8(*reinterpret_cast<const u32x4*>(sR + o), fr);
            unpack8(*reinterpret_cast<const u32x4*>(sDy + o), fd);
            m1[i] = 0.f; m2[i] = 0.f;
            const bool live = row < S;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                xh[i][k] = (fa[k] + fr[k] - mu) * rs[i];
                gd[i][k] = fd[k] * g8[k];
                asm volatile("" : "+v"(gd[i][k]));           // (as in k_add_ln_bwd: a rounded product)
                m1[i] += gd[i][k];
                m2[i] = __builtin_fmaf(gd[i][k], xh[i][k], m2[i]);
                if (live) { ag[k] = __builtin_fmaf(fd[k], xh[i][k], ag[k]); ab[k] += fd[k]; }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { m1[i] = wave_sum(m1[i]) / (float)ED; m2[i] = wave_sum(m2[i]) / (float)ED; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 8 + half * 4 + i;
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = rs[i] * __builtin_fmaf(-xh[i][k], m2[i], gd[i][k] - m1[i]);
            const u32x4 ds = pack8(o);
            u32x4 dd = ds;
            const long off = (row0 + row) * ED + lane * 8;
            if (drop_on) {
                float v[8];
                unpack8(ds, v);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = drop_keep(key, th, (unsigned)(off + k)) ? v[k] * sc : 0.f;
                dd = pack8(v);
            }
            *reinterpret_cast<u32x4*>(bufA + (lane >> 3) * 2048 + EF_SWZ(row, lane & 7)) = dd;
            if (row < S && (lane >> 3) == h) {
                st16_sc1(ds_out + off, ds);
                if (dd_out != ds_out) st16_sc1(dd_out + off, dd);
            }
        }
    }
}

}  // namespace

__global__ __launch_bounds__(256, 1) void k_enc_bwd(const EncBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* bufA = reinterpret_cast<bf16_t*>(smem + BUFA);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int c, h;
    {
        const int i = blockIdx.x;
        if ((p.nseq & 7) == 0) { const int j = i >> 3; h = j & 7; c = (i & 7) + 8 * (j >> 3); }
        else { c = i >> 3; h = i & 7; }
    }
    const int S = p.S;
    const long row0 = (long)(p.seq0 + c) * S;
    const int bh = (p.seq0 + c) * EH + h;
    unsigned* cnt = p.cnt + c;
    unsigned arrivals = 0;
    bool gave_up = false;
    int tix = 0;
    auto stamp = [&]() {
        if (p.trace != nullptr && blockIdx.x == 0 && tid == 0 && tix < 500) p.trace[tix] = __builtin_amdgcn_s_memtime();
        ++tix;
    };
    stamp();
    const int slot8 = tid & 7, r32 = tid >> 3;
    const int csw = slot8 ^ ((r32 >> 1) & 7);
    const int rsrc = r32 < S ? r32 : S - 1;
    const bool drop_on = p.seed != nullptr;
    const int ldp = (S + 7) & ~7;
    auto ring = [&](int slot) { return reinterpret_cast<bf16_t*>(smem + RING + slot * SLOT); };
    // a [rows][512] bf16 tensor's rows of this sequence -> one ring slot in the A layout (8 DMA pieces per thread)
    auto rows512 = [&](const bf16_t* t, int slot, bool sc1) {
        const bf16_t* src = t + (row0 + rsrc) * ED + csw * 8;
        bf16_t* dst = ring(slot) + wave * 8 * 64;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            if (sc1) glds16_sc1(src + kb * 64, dst + kb * 2048); else glds16(src + kb * 64, dst + kb * 2048);
        }
    };
    auto ln_consts = [&](const float* gamma, const float* mean, const float* rstd, float (&g8)[8], float (&mu8)[8], float (&rs8)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = wave * 8 + i;
            const long gr = row0 + (row < S ? row : S - 1);
            mu8[i] = mean[gr]; rs8[i] = rstd[gr];
            g8[i] = gamma[lane * 8 + i];
        }
    };
    // this sequence's partial row of a LayerNorm's gamma / beta gradient: the four waves' column sums meet in LDS (slot 3) in a fixed order;
    // workgroup h writes columns h*64 .. +63 of both halves
    auto ln_partials = [&](const float (&ag)[8], const float (&ab)[8], float* part) {
        float* sc = reinterpret_cast<float*>(ring(3));           // [4 waves][2][512]
#pragma unroll
        for (int k = 0; k < 8; ++k) { sc[(wave * 2 + 0) * ED + lane * 8 + k] = ag[k]; sc[(wave * 2 + 1) * ED + lane * 8 + k] = ab[k]; }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, col = h * 64 + (tid & 63);
            part[(long)(p.seq0 + c) * 2 * ED + which * ED + col] =
                ((sc[(0 + which) * ED + col] + sc[(2 + which) * ED + col]) + sc[(4 + which) * ED + col]) + sc[(6 + which) * ED + col];
        }
    };

    for (int l = p.layers - 1; l >= 0; --l) {
        const svsr_enc_bwd_layer L = p.Ls[l];
        const bf16_t* dyin = l == p.layers - 1 ? p.dy : reinterpret_cast<const bf16_t*>(p.Ls[l + 1].dx);
        const unsigned key_pr = drop_on ? svsr_mix(p.seed[0] * 0x9E3779B9u + L.site_probs * 0x7F4A7C15u + 0x165667B1u) : 0u;
        const unsigned key_ao = drop_on ? svsr_mix(p.seed[0] * 0x9E3779B9u + L.site_ao * 0x7F4A7C15u + 0x165667B1u) : 0u;
        const unsigned key_fo = drop_on ? svsr_mix(p.seed[0] * 0x9E3779B9u + L.site_fo * 0x7F4A7C15u + 0x165667B1u) : 0u;

        // =========================== B1: ds2 = LN2 backward, df = mask(ds2) -> bufA ==========================================
        {
            // what the forward kept (f, x1, statistics) does not depend on the other workgroups: requested AHEAD of the cluster wait
            // (for every layer but the first one processed: at the end of the previous iteration, behind its arrival signal)
            if (l == p.layers - 1) {
                rows512(reinterpret_cast<const bf16_t*>(L.f), 1, false);
                rows512(reinterpret_cast<const bf16_t*>(L.x1), 2, false);
            }
            float g8[8], mu8[8], rs8[8];                     // this wave's rows' statistics and this lane's gammas travel with the DMA
            ln_consts(L.g2, L.m2, L.r2, g8, mu8, rs8);
            if (l != p.layers - 1) {                        // the layer above has written every column of its dx
                arrivals += EH;
                cluster_wait(cnt, arrivals, p.err, p.host_flag, p.spin_limit, gave_up);
            }
            stamp();                                         // [b0] layer start
            rows512(dyin, 0, true);
            EF_WAIT_VM(0);
            EF_BARRIER();
            float ag[8], ab[8];
            ln_bwd8(ring(0), ring(1), ring(2), g8, mu8, rs8, row0, S, wave, lane, h, drop_on, key_fo, p.th_hidden, p.sc_hidden, bufA,
                    reinterpret_cast<bf16_t*>(L.ds2), reinterpret_cast<bf16_t*>(L.df), ag, ab);
            ln_partials(ag, ab, L.part2);
            stamp();                                         // [b1] LN2 backward done
        }

        // =========================== B2: dz[:, h*256 ..] = (df W_2)[...] * gelu'(z) ==========================================
        {
            const bf16_t* W = reinterpret_cast<const bf16_t*>(L.w2t) + (long)(h * IH) * ED;          // rows = dhg columns, 512 long
            auto stageW = [&](int s, int slot) {
                bf16_t* dst = ring(slot);
#pragma unroll
                for (int i = 0; i < 8; ++i) glds16(W + (long)(i * 32 + r32) * ED + s * 64 + csw * 8, dst + (i * 32 + wave * 8) * 64);
            };
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            // z of this wave's epilogue rows: requested now, used after the K loop (plain loads: written by the forward launch)
            u32x4 zr[4];
            {
                const int n = h * IH + wave * 64 + (lane & 7) * 8;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = (lane >> 3) + 8 * i;
                    zr[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(L.z) + (row0 + (row < S ? row : S - 1)) * EI + n);
                }
            }
            EF_WAIT_VM(0);                                   // this phase's own stores leave the queue before counted waits start
            __syncthreads();                                 // slots 0..3 (LayerNorm inputs, partial sums) are consumed; df is in bufA
            stageW(0, 0); stageW(1, 1); stageW(2, 2);
            for (int s = 0; s < 8; ++s) {
                if (s < 6) EF_WAIT_VM(16); else if (s == 6) EF_WAIT_VM(8); else EF_WAIT_VM(0);
                EF_BARRIER();
                if (s + 3 < 8) stageW(s + 3, (s + 3) & 3);
                const bf16_t* B = ring(s & 3) + wave * 64 * 64;
                const bf16_t* A = bufA + s * 2048;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int ch = ks * 2 + (lane >> 5);
                    const bf16x8 fa = lds_frag(A, lane & 31, ch);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, lds_frag(B, j * 32 + (lane & 31), ch), acc[j], 0, 0, 0);
                }
            }
            __syncthreads();                                 // df in bufA and the ring are consumed
            {
                float* st = reinterpret_cast<float*>(bufA) + wave * 2048;               // fp32 [4][32][64]
                acc_to_stage(acc[0], st, 64, 0, lane);
                acc_to_stage(acc[1], st, 64, 32, lane);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int c8 = lane & 7, n = h * IH + wave * 64 + c8 * 8;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = (lane >> 3) + 8 * i;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(st + row * 64 + c8 * 8), hi = *reinterpret_cast<const f32x4*>(st + row * 64 + c8 * 8 + 4);
                    const float a8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    float g[8], z[8];
                    unpack8(pack8(a8), g);                   // dhg is a bf16 tensor in the launch chain
                    unpack8(zr[i], z);
#pragma unroll
                    for (int k = 0; k < 8; ++k) g[k] = g[k] * gelu_erf_grad(z[k]) * 1.0f;
                    if (row < S) st16_sc1(reinterpret_cast<bf16_t*>(L.dz) + (row0 + row) * EI + n, pack8(g));
                }
            }
            stamp();                                         // [b2] B2 done
            cluster_signal(cnt);
        }
        // this thread's piece of ds2 (its own workgroup's store, drained by the signal above): the addend of B4's epilogue
        const long eoff = (row0 + ((tid >> 3) < S ? (tid >> 3) : S - 1)) * ED + h * 64 + (tid & 7) * 8;
        const u32x4 ad2 = ld16_sc1(reinterpret_cast<const bf16_t*>(L.ds2) + eoff);

        // =========================== B4: dx1[:, h*64 ..] = dz W_1 + ds2 ======================================================
        {
            const bf16_t* W = reinterpret_cast<const bf16_t*>(L.w1t) + (long)(h * 64) * EI;            // rows = dx1 columns, 2048 long
            const bf16_t* DZ = reinterpret_cast<const bf16_t*>(L.dz) + (row0 + rsrc) * EI + csw * 8;
            auto stageW = [&](int s, int slot) {            // chunks s (K half 0) and 16 + s (K half 1): 64 rows x 64 k each
                bf16_t* dst = ring(slot);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int g = i >> 1, row = (i & 1) * 32 + r32;
                    glds16(W + (long)row * EI + (g * 16 + s) * 64 + csw * 8, dst + g * 4096 + ((i & 1) * 32 + wave * 8) * 64);
                }
            };
            auto stageA = [&](int s, int slot) {
                bf16_t* dst = ring(slot) + 8192;
#pragma unroll
                for (int g = 0; g < 2; ++g) glds16_sc1(DZ + (g * 16 + s) * 64, dst + g * 2048 + wave * 8 * 64);
            };
            // the weight halves of the first three tiles do not depend on the other workgroups: in flight across the cluster wait
            stageW(0, 0); stageW(1, 1); stageW(2, 2);
            arrivals += EH;
            cluster_wait(cnt, arrivals, p.err, p.host_flag, p.spin_limit, gave_up);
            stamp();                                         // [b3] barrier 1 passed
            const u32x4 ad = ad2;
            stageA(0, 0); stageA(1, 1); stageA(2, 2);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int g = wave >> 1, jb = wave & 1;
            for (int s = 0; s < 16; ++s) {
                // outstanding behind tile s: s = 0: the dz pieces of tiles 1, 2 (4); s = 1: dz of tile 2 + tile 3 (8); then tiles s+1, s+2 (12); the tail drains
                if (s == 0) EF_WAIT_VM(4); else if (s == 1) EF_WAIT_VM(8); else if (s < 14) EF_WAIT_VM(12); else if (s == 14) EF_WAIT_VM(6); else EF_WAIT_VM(0);
                EF_BARRIER();
                if (s + 3 < 16) { stageW(s + 3, (s + 3) & 3); stageA(s + 3, (s + 3) & 3); }
                const bf16_t* T = ring(s & 3);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int ch = ks * 2 + (lane >> 5);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(T + 8192 + g * 2048, lane & 31, ch), lds_frag(T + g * 4096, jb * 32 + (lane & 31), ch), acc, 0, 0, 0);
                }
            }
            __syncthreads();
            float* stage = reinterpret_cast<float*>(bufA);   // fp32 [2 K halves][32][64]
            acc_to_stage(acc, stage + g * 2048, 64, jb * 32, lane);
            __syncthreads();
            {
                const int row = tid >> 3, c8 = tid & 7, n = h * 64 + c8 * 8;
                const long off = eoff;
                const f32x4 lo0 = *reinterpret_cast<const f32x4*>(stage + row * 64 + c8 * 8), hi0 = *reinterpret_cast<const f32x4*>(stage + row * 64 + c8 * 8 + 4);
                const f32x4 lo1 = *reinterpret_cast<const f32x4*>(stage + 2048 + row * 64 + c8 * 8), hi1 = *reinterpret_cast<const f32x4*>(stage + 2048 + row * 64 + c8 * 8 + 4);
                float v[8] = {lo0[0] + lo1[0], lo0[1] + lo1[1], lo0[2] + lo1[2], lo0[3] + lo1[3], hi0[0] + hi1[0], hi0[1] + hi1[1], hi0[2] + hi1[2], hi0[3] + hi1[3]};
                float a8[8];
                unpack8(ad, a8);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += a8[k];
                if (row < S) st16_sc1(reinterpret_cast<bf16_t*>(L.dx1) + off, pack8(v));
            }
            stamp();                                         // [b4] B4 done
            cluster_signal(cnt);
        }

        // q | k | v of this head and the probabilities (written by the forward launch: plain loads), used in B7: requested ahead of the barrier
        const int arow = tid >> 3, c8 = tid & 7;
        const bf16_t* qsrc = reinterpret_cast<const bf16_t*>(L.qkv) + (row0 + (arow < S ? arow : S - 1)) * (3 * ED) + h * 64 + c8 * 8;
        const u32x4 q8 = *reinterpret_cast<const u32x4*>(qsrc), k8 = *reinterpret_cast<const u32x4*>(qsrc + ED), v8 = *reinterpret_cast<const u32x4*>(qsrc + 2 * ED);
        float prv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = 8 * wave + q + 4 * (lane >> 5), j = lane & 31;
            prv[q] = (i < S && j < S) ? bf2f(reinterpret_cast<const bf16_t*>(L.probs)[((long)bh * S + i) * ldp + j]) : 0.f;
        }

        // =========================== B5: ds1 = LN1 backward of dx1, dao = mask(ds1) -> bufA ==================================
        {
            rows512(reinterpret_cast<const bf16_t*>(L.ao), 1, false);          // (the ring is idle: B4's K loop is behind a workgroup barrier)
            rows512(reinterpret_cast<const bf16_t*>(L.xin), 2, false);
            float g8[8], mu8[8], rs8[8];
            ln_consts(L.g1, L.m1, L.r1, g8, mu8, rs8);
            arrivals += EH;
            cluster_wait(cnt, arrivals, p.err, p.host_flag, p.spin_limit, gave_up);
            stamp();                                         // [b5] barrier 2 passed
            rows512(reinterpret_cast<const bf16_t*>(L.dx1), 0, true);
            EF_WAIT_VM(0);
            EF_BARRIER();
            float ag[8], ab[8];
            ln_bwd8(ring(0), ring(1), ring(2), g8, mu8, rs8, row0, S, wave, lane, h, drop_on, key_ao, p.th_hidden, p.sc_hidden, bufA,
                    reinterpret_cast<bf16_t*>(L.ds1), reinterpret_cast<bf16_t*>(L.dao), ag, ab);
            ln_partials(ag, ab, L.part1);
            stamp();                                         // [b6] LN1 backward done
        }

        // =========================== B6: dctx of head h = (dao W_o)[:, h*64 ..] -> LDS ========================================
        // =========================== B7: attention backward of head h -> dq | dk | dv ========================================
        {
            const bf16_t* Wo = reinterpret_cast<const bf16_t*>(L.wot) + (long)(h * 64) * ED;           // rows = ctx columns of head h, 512 long
            auto stageWo = [&](int s, int slot) {           // 64 rows x 256 k (4 chunks of 64): 8 pieces per thread
                bf16_t* dst = ring(slot);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int sub = i >> 1, row = (i & 1) * 32 + r32;
                    glds16(Wo + (long)row * ED + s * 256 + sub * 64 + csw * 8, dst + sub * 4096 + ((i & 1) * 32 + wave * 8) * 64);
                }
            };
            EF_WAIT_VM(0);
            __syncthreads();                                 // LayerNorm inputs / partial sums consumed, dao in bufA
            stageWo(0, 1); stageWo(1, 2);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            EF_WAIT_VM(0);
            EF_BARRIER();
            if (wave < 2) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16_t* B = ring(1 + s);
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            const int ch = ks * 2 + (lane >> 5);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(bufA + (s * 4 + sub) * 2048, lane & 31, ch),
                                                                           lds_frag(B + sub * 4096, wave * 32 + (lane & 31), ch), acc, 0, 0, 0);
                        }
                }
            }
            float* stage = reinterpret_cast<float*>(ring(0));     // fp32 [32][64]
            if (wave < 2) acc_to_stage(acc, stage, 64, wave * 32, lane);
            __syncthreads();                                 // dao in bufA is consumed: bufA becomes the attention operands
            bf16_t* Da = bufA;                               // [32 i][64 d]   A layout of dctx
            bf16_t* Vb = bufA + 2048;                        // [32 j][64 d]   B layout of v
            bf16_t* Kt = bufA + 4096;                        // [64 d][64: j]  B layout of k^T
            bf16_t* Qt = bufA + 8192;                        // [64 d][64: i]  B layout of q^T
            bf16_t* Dt = bufA + 12288;                       // [64 d][64: i]  B layout of dctx^T
            bf16_t* dSa = ring(0) + 4096;                    // [32 i][64: j]  A layout of dS         (behind the 8 KiB staging)
            bf16_t* dSt = ring(0) + 6144;                    // [32 j][64: i]  A layout of dS^T
            bf16_t* Pt = ring(0) + 8192;                     // [32 j][64: i]  A layout of dropout(P)^T
            {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + arow * 64 + c8 * 8), hi = *reinterpret_cast<const f32x4*>(stage + arow * 64 + c8 * 8 + 4);
                const float d8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const u32x4 dk = pack8(d8);
                *reinterpret_cast<u32x4*>(Da + EF_SWZ(arow, c8)) = dk;
                *reinterpret_cast<u32x4*>(Vb + EF_SWZ(arow, c8)) = v8;
                const unsigned wd[4] = {dk.x, dk.y, dk.z, dk.w}, wk[4] = {k8.x, k8.y, k8.z, k8.w}, wq[4] = {q8.x, q8.y, q8.z, q8.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int o = EF_SWZ(c8 * 8 + e, arow >> 3) + (arow & 7);
                    Dt[o] = (bf16_t)((wd[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
                    Kt[o] = (bf16_t)((wk[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
                    Qt[o] = (bf16_t)((wq[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
                }
            }
            stamp();                                         // [b7] dctx done, attention operands staged
            EF_BARRIER();
            {   // dP of the whole 32 x 32 block on every wave; wave w finishes rows 8w .. 8w+7
                f32x16 sc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int ch = ks * 2 + (lane >> 5);
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(Da, lane & 31, ch), lds_frag(Vb, lane & 31, ch), sc, 0, 0, 0);
                }
                const int j = lane & 31;
                float dp[4], part[4];
                bool keep[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x = sc[0];
                    if (wave == 0) x = sc[q]; else if (wave == 1) x = sc[4 + q]; else if (wave == 2) x = sc[8 + q]; else x = sc[12 + q];
                    const int i = 8 * wave + q + 4 * (lane >> 5);
                    keep[q] = !drop_on || drop_keep(key_pr, p.th_attn, (unsigned)(((long)bh * S + i) * ldp + j));
                    dp[q] = (i < S && j < S) ? (drop_on ? (keep[q] ? x * p.sc_attn : 0.f) : x) : 0.f;
                    asm volatile("" : "+v"(dp[q]));          // a rounded product (k_mha_bwd_q4 keeps it in LDS): not to be fused into dp - rowsum below
                    part[q] = prv[q] * dp[q];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float a = part[q];
                    a += dpp_xor1(a); a += dpp_xor2(a); a += dpp_half_mirror(a); a += dpp_mirror(a);
                    part[q] = a + swz_xor16(a);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 8 * wave + q + 4 * (lane >> 5);
                    const bf16_t ds = f2bf(prv[q] * (dp[q] - part[q]) * 0.125f);
                    const bf16_t pd = drop_on ? (keep[q] ? f2bf(prv[q] * p.sc_attn) : (bf16_t)0) : f2bf(prv[q]);
                    dSa[EF_SWZ(i, j >> 3) + (j & 7)] = ds;
                    dSt[EF_SWZ(j, i >> 3) + (i & 7)] = ds;
                    Pt[EF_SWZ(j, i >> 3) + (i & 7)] = pd;
                }
            }
            EF_BARRIER();
            float* ost = reinterpret_cast<float*>(ring(1));  // fp32 [3: dq, dk, dv][32][64]  (W_o's tiles are consumed)
            {   // six (matrix, 32-column block) products of two MFMAs each: wave 0/1 dq then dv, wave 2/3 dk; two 16-deep k slices added at the end
                const int nb = wave & 1;
                const bf16_t* A0 = wave < 2 ? dSa : dSt;
                const bf16_t* B0 = wave < 2 ? Kt : Qt;
                f32x16 c0, c1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(A0, lane & 31, lane >> 5), lds_frag(B0, nb * 32 + (lane & 31), lane >> 5), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(A0, lane & 31, 2 + (lane >> 5)), lds_frag(B0, nb * 32 + (lane & 31), 2 + (lane >> 5)), c1, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) c0[r] += c1[r];
                acc_to_stage(c0, ost + (wave < 2 ? 0 : 2048), 64, nb * 32, lane);
                if (wave < 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(Pt, lane & 31, lane >> 5), lds_frag(Dt, nb * 32 + (lane & 31), lane >> 5), c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(Pt, lane & 31, 2 + (lane >> 5)), lds_frag(Dt, nb * 32 + (lane & 31), 2 + (lane >> 5)), c1, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) c0[r] += c1[r];
                    acc_to_stage(c0, ost + 4096, 64, nb * 32, lane);
                }
            }
            __syncthreads();
            if (arow < S) {
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(ost + m * 2048 + arow * 64 + c8 * 8), hi = *reinterpret_cast<const f32x4*>(ost + m * 2048 + arow * 64 + c8 * 8 + 4);
                    const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    st16_sc1(reinterpret_cast<bf16_t*>(L.dqkv) + (row0 + arow) * (3 * ED) + m * ED + h * 64 + c8 * 8, pack8(v));
                }
            }
            stamp();                                         // [b8] attention backward done
            cluster_signal(cnt);
        }
        const u32x4 ad1 = ld16_sc1(reinterpret_cast<const bf16_t*>(L.ds1) + eoff);

        // =========================== B8: dx[:, h*64 ..] = dqkv W_qkv + ds1 ===================================================
        {
            constexpr int KQ = 3 * ED, HC = KQ / 128;        // 1536 deep: two K halves of 12 chunks
            const bf16_t* W = reinterpret_cast<const bf16_t*>(L.wqkvt) + (long)(h * 64) * KQ;
            const bf16_t* DQ = reinterpret_cast<const bf16_t*>(L.dqkv) + (row0 + rsrc) * KQ + csw * 8;
            auto stageW = [&](int s, int slot) {
                bf16_t* dst = ring(slot);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int g = i >> 1, row = (i & 1) * 32 + r32;
                    glds16(W + (long)row * KQ + (g * HC + s) * 64 + csw * 8, dst + g * 4096 + ((i & 1) * 32 + wave * 8) * 64);
                }
            };
            auto stageA = [&](int s, int slot) {
                bf16_t* dst = ring(slot) + 8192;
#pragma unroll
                for (int g = 0; g < 2; ++g) glds16_sc1(DQ + (g * HC + s) * 64, dst + g * 2048 + wave * 8 * 64);
            };
            stageW(0, 0); stageW(1, 1); stageW(2, 2);        // (the attention's LDS scratch in slots 0 and 1 is behind the signal's workgroup barrier)
            arrivals += EH;
            cluster_wait(cnt, arrivals, p.err, p.host_flag, p.spin_limit, gave_up);
            stamp();                                         // [b9] barrier 3 passed
            const u32x4 ad = ad1;
            stageA(0, 0); stageA(1, 1); stageA(2, 2);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int g = wave >> 1, jb = wave & 1;
            for (int s = 0; s < HC; ++s) {
                if (s == 0) EF_WAIT_VM(4); else if (s == 1) EF_WAIT_VM(8); else if (s < HC - 2) EF_WAIT_VM(12); else if (s == HC - 2) EF_WAIT_VM(6); else EF_WAIT_VM(0);
                EF_BARRIER();
                if (s + 3 < HC) { stageW(s + 3, (s + 3) & 3); stageA(s + 3, (s + 3) & 3); }
                const bf16_t* T = ring(s & 3);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int ch = ks * 2 + (lane >> 5);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(T + 8192 + g * 2048, lane & 31, ch), lds_frag(T + g * 4096, jb * 32 + (lane & 31), ch), acc, 0, 0, 0);
                }
            }
            __syncthreads();
            float* stage = reinterpret_cast<float*>(bufA);
            acc_to_stage(acc, stage + g * 2048, 64, jb * 32, lane);
            __syncthreads();
            {
                const int row = tid >> 3, c8 = tid & 7, n = h * 64 + c8 * 8;
                const long off = eoff;
                const f32x4 lo0 = *reinterpret_cast<const f32x4*>(stage + row * 64 + c8 * 8), hi0 = *reinterpret_cast<const f32x4*>(stage + row * 64 + c8 * 8 + 4);
                const f32x4 lo1 = *reinterpret_cast<const f32x4*>(stage + 2048 + row * 64 + c8 * 8), hi1 = *reinterpret_cast<const f32x4*>(stage + 2048 + row * 64 + c8 * 8 + 4);
                float v[8] = {lo0[0] + lo1[0], lo0[1] + lo1[1], lo0[2] + lo1[2], lo0[3] + lo1[3], hi0[0] + hi1[0], hi0[1] + hi1[1], hi0[2] + hi1[2], hi0[3] + hi1[3]};
                float a8[8];
                unpack8(ad, a8);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += a8[k];
                if (row < S) st16_sc1(reinterpret_cast<bf16_t*>(L.dx) + off, pack8(v));
            }
            stamp();                                         // [b10] B8 done
            if (l > 0) {
                cluster_signal(cnt);                         // (its workgroup barrier: the staging in bufA is consumed, the ring is idle)
                const svsr_enc_bwd_layer& Ln = p.Ls[l - 1];
                rows512(reinterpret_cast<const bf16_t*>(Ln.f), 1, false);
                rows512(reinterpret_cast<const bf16_t*>(Ln.x1), 2, false);
            }
        }
    }
    enc_poison(gave_up, reinterpret_cast<bf16_t*>(p.Ls[0].dx) + row0 * ED + h * 64);
}

struct EncBwdTable { svsr_enc_bwd_layer L[8]; };
static_assert(sizeof(EncBwdTable) <= 3584, "the layer records travel in the kernel-argument segment");

__global__ __launch_bounds__(256) void k_enc_bwd_table(const EncBwdTable t, int words, long long* __restrict__ dst) {
    const long long* src = reinterpret_cast<const long long*>(&t);
    for (int i = threadIdx.x; i < words; i += 256) dst[i] = src[i];
}

struct EncTable { svsr_enc_layer L[8]; };
static_assert(sizeof(EncTable) <= 3584, "the layer records travel in the kernel-argument segment");

// the layer records reach the device as the arguments of this writer kernel (no host pointer survives the call: graph- and replay-safe)
__global__ __launch_bounds__(256) void k_enc_table(const EncTable t, int words, long long* __restrict__ dst) {
    const long long* src = reinterpret_cast<const long long*>(&t);
    for (int i = threadIdx.x; i < words; i += 256) dst[i] = src[i];
}

static unsigned long long* g_enc_trace = nullptr;

// Layer-record tables that are already on the device (round 6): a training step calls svsr_enc_fwd / svsr_enc_bwd with the same records every
// time (the recorded step list re-issues the call with frozen arguments), and the writer kernel was a 5-us launch in the main stream's chain
// each time.  A table is uploaded ONCE into a slot of a small pool and found again by its bytes; slots are never recycled (a captured graph
// or an enqueued launch may still read them): when the pool is full, or the stream is being captured, the table goes into the caller's
// workspace as before.  The pool is allocated by the workspace-size queries (never inside a capture).
constexpr int TAB_SLOTS = 64, TAB_BYTES = 3584;
struct TabEntry { unsigned char bytes[TAB_BYTES]; size_t n; hipStream_t stream; };
static char* g_tab_pool = nullptr;
static TabEntry* g_tab_entries = nullptr;
static int g_tab_used = 0;
static void enc_tab_pool_init() {
    if (g_tab_pool != nullptr) return;
    void* d = nullptr;
    if (hipMalloc(&d, (size_t)TAB_SLOTS * TAB_BYTES) != hipSuccess) { (void)hipGetLastError(); return; }
    g_tab_entries = static_cast<TabEntry*>(malloc(sizeof(TabEntry) * TAB_SLOTS));
    if (g_tab_entries == nullptr) { (void)hipFree(d); return; }
    g_tab_pool = static_cast<char*>(d);
}
// -> device address of an uploaded table with these bytes (hit), or of a fresh slot the caller must fill on `stream` (*fill = true), or null
static void* enc_tab_find(const void* t, size_t n, hipStream_t stream, bool* fill) {
    *fill = false;
    static int off = -1;
    if (off < 0) { const char* e = getenv("SVSR_ENC_TAB_CACHE"); off = (e != nullptr && e[0] == '0') ? 1 : 0; }      // SVSR_ENC_TAB_CACHE=0: a writer launch per call, as before
    if (off || g_tab_pool == nullptr || n > (size_t)TAB_BYTES) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (cs != hipStreamCaptureStatusNone) return nullptr;
    for (int i = 0; i < g_tab_used; ++i)
        if (g_tab_entries[i].n == n && g_tab_entries[i].stream == stream && memcmp(g_tab_entries[i].bytes, t, n) == 0) return g_tab_pool + (size_t)i * TAB_BYTES;
    if (g_tab_used >= TAB_SLOTS) return nullptr;
    TabEntry& e = g_tab_entries[g_tab_used];
    memcpy(e.bytes, t, n); e.n = n; e.stream = stream;
    *fill = true;
    return g_tab_pool + (size_t)(g_tab_used++) * TAB_BYTES;
}

// One word of pinned, device-mapped host memory: a giving-up workgroup stores 1 into it (system scope), the host polls it for free.
// Allocated by the first workspace-size query (every launch is preceded by one; never inside a stream capture).
static unsigned g_enc_spin_limit = 1u << 20;       // polls (s_sleep 4 + one L2 read each: ~1 s in all) before a cluster wait gives up
static unsigned* g_enc_host_flag = nullptr;        // host address
static unsigned* g_enc_host_flag_dev = nullptr;    // the device's address of the same word
static void enc_host_flag_init() {
    if (g_enc_host_flag != nullptr) return;
    void* h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return; }
    *static_cast<volatile unsigned*>(h) = 0;
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return; }
    g_enc_host_flag = static_cast<unsigned*>(h);
    g_enc_host_flag_dev = static_cast<unsigned*>(d);
}

// sequences per launch: every workgroup takes a whole compute unit's LDS and spin-waits on its 7 siblings, so a launch may hold at most
// (compute units of the stream) / 8 clusters (32 on the whole chip; fewer on a CU-masked stream)
static int enc_clusters_per_launch(hipStream_t stream) {
    const int n = svsr_stream_cus(stream) / EH;
    return n < 1 ? 1 : n;
}

extern "C" {

/* debug aid of scripts/probes: the next svsr_enc_fwd launches stamp s_memtime of workgroup 0 at their phase boundaries (1 + 12 per layer);
 * svsr_debug_enc_trace(out, n) synchronises and copies the first n stamps out, then switches tracing off */
int svsr_debug_enc_trace(int64_t* out, int n) {
    if (out == nullptr) {
        if (g_enc_trace == nullptr && hipMalloc(reinterpret_cast<void**>(&g_enc_trace), 512 * sizeof(unsigned long long)) != hipSuccess) return SVSR_ERR_LAUNCH;
        return (int)hipMemset(g_enc_trace, 0, 512 * sizeof(unsigned long long));
    }
    if (g_enc_trace == nullptr || n < 1 || n > 512) return SVSR_ERR_ARG;
    (void)hipDeviceSynchronize();
    const int rc = (int)hipMemcpy(out, g_enc_trace, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(g_enc_trace); g_enc_trace = nullptr;
    return rc;
}

/* bytes of the device workspace svsr_enc_fwd needs for B sequences: arrival counters, error word, layer records */
int64_t svsr_enc_fwd_ws_bytes(int B) { enc_host_flag_init(); enc_tab_pool_init(); return B < 1 ? 0 : (int64_t)(((B + 1) * 4 + 255) / 256 * 256) + (int64_t)sizeof(EncTable); }

/* svsr_enc_fwd: forward of `n_layers` consecutive HF-BERT encoder layers (width 512, 8 heads of 64, FFN 2048, sequences of S <= 32
 * rows) in one launch per 32 sequences.  x0 bf16 [B*S][512]: the first layer's input; layers: HOST array of n_layers records (device
 * pointers of the weights' bf16 shadows, fp32 biases / LayerNorm parameters and of the tensors the layer writes: the same tensors, with
 * the same contents, as the unfused launches svsr_igemm_fwd / svsr_mha_fwd / svsr_add_ln_fwd produce).  drop_seed null: no dropout.
 * ws: device workspace of svsr_enc_fwd_ws_bytes(B) bytes (counters are zeroed here with a memset on `stream`; word B is the error flag:
 * non-zero after the launch if a bounded wait gave up). */
int svsr_enc_fwd(const void* x0, const svsr_enc_layer* layers, int n_layers, int B, int S, float ln_eps, const unsigned* drop_seed,
                 float p_hidden, float p_attn, void* ws, int64_t ws_bytes, hipStream_t stream) {
    if (x0 == nullptr || layers == nullptr || n_layers < 1 || n_layers > 8 || B < 1 || S < 1 || S > TR || ws == nullptr || ws_bytes < svsr_enc_fwd_ws_bytes(B))
        return SVSR_ERR_ARG;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_enc_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL); attr = true; }
    unsigned* cnt = static_cast<unsigned*>(ws);
    const size_t cnt_bytes = (size_t)(((B + 1) * 4 + 255) / 256 * 256);
    hipError_t e = hipMemsetAsync(cnt, 0, cnt_bytes, stream);
    if (e != hipSuccess) return (int)e;
    EncTable t;
    memset(&t, 0, sizeof t);
    for (int l = 0; l < n_layers; ++l) t.L[l] = layers[l];
    bool fill = true;
    svsr_enc_layer* tab_dev = static_cast<svsr_enc_layer*>(enc_tab_find(&t, sizeof t, stream, &fill));
    if (tab_dev == nullptr) { tab_dev = reinterpret_cast<svsr_enc_layer*>(static_cast<char*>(ws) + cnt_bytes); fill = true; }
    if (fill) hipLaunchKernelGGL(k_enc_table, dim3(1), dim3(256), 0, stream, t, (int)(sizeof(EncTable) / 8), reinterpret_cast<long long*>(tab_dev));
    EncArgs a;
    a.x0 = (const bf16_t*)x0; a.Ls = tab_dev;
    a.layers = n_layers; a.S = S; a.eps = ln_eps;
    const DropArgs dh = svsr_make_drop(drop_seed, 0, p_hidden), da = svsr_make_drop(drop_seed, 0, p_attn);
    a.seed = (dh.seed != nullptr || da.seed != nullptr) ? drop_seed : nullptr;
    a.th_hidden = dh.thresh; a.sc_hidden = dh.scale; a.th_attn = da.thresh; a.sc_attn = da.scale;
    a.err = cnt + B;
    a.host_flag = g_enc_host_flag_dev;
    a.spin_limit = g_enc_spin_limit;
    a.trace = g_enc_trace;
    const int per = enc_clusters_per_launch(stream);      // all workgroups of a launch must be resident together: 8 per sequence, one per compute unit
    for (int s0 = 0; s0 < B; s0 += per) {
        a.seq0 = s0; a.nseq = B - s0 < per ? B - s0 : per; a.cnt = cnt + s0;
        hipLaunchKernelGGL(k_enc_fwd, dim3(a.nseq * EH), dim3(256), LDS_TOTAL, stream, a);
    }
    return svsr_check_launch();
}

/* 1 if ANY svsr_enc_fwd / svsr_enc_bwd launch of this process had a bounded cluster wait give up (its results were poisoned with NaN), else 0;
 * reset != 0 clears the word afterwards.  Synchronises the device (call it where the host synchronises anyway). */
int svsr_enc_gave_up(int reset) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_enc_gave_up), sizeof v) != hipSuccess) return -1;
    if (reset && v != 0) { const unsigned z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_enc_gave_up), &z, sizeof z); }
    if (reset && g_enc_host_flag != nullptr) *static_cast<volatile unsigned*>(g_enc_host_flag) = 0;
    return v != 0 ? 1 : 0;
}

/* test aid: polls a cluster wait makes before it gives up (0 restores the default, 2^20).  tests/test_gpu_cotenant.py sets 1 to provoke the
 * give-up path on purpose: a wait that does not find its siblings arrived at the second look poisons the launch. */
int svsr_debug_enc_spin_limit(unsigned limit) {
    g_enc_spin_limit = limit == 0 ? (1u << 20) : limit;
    return SVSR_OK;
}

/* The same flag WITHOUT a synchronisation: the giving-up workgroup also stores it into a word of pinned host memory, which this reads
 * (it may trail the device by the flight time of that store).  engine.TrainStep looks at it before every step and, when it is set,
 * switches the encoder to the per-layer launch chain (svsr_igemm_fwd / svsr_mha_fwd / svsr_add_ln_fwd ...) for the rest of the run. */
int svsr_enc_gave_up_peek(void) {
    return g_enc_host_flag != nullptr && *static_cast<volatile unsigned*>(g_enc_host_flag) != 0 ? 1 : 0;
}

/* bytes of the device workspace svsr_enc_bwd needs for B sequences */
int64_t svsr_enc_bwd_ws_bytes(int B) { enc_host_flag_init(); enc_tab_pool_init(); return B < 1 ? 0 : (int64_t)(((B + 1) * 4 + 255) / 256 * 256) + (int64_t)sizeof(EncBwdTable); }

/* svsr_enc_bwd: backward of the layers svsr_enc_fwd ran, in one launch per 32 sequences.  dy bf16 [B*S][512]: gradient of the last layer's
 * output; layers: HOST array of n_layers records in FORWARD order.  Written per layer: ds2 / df / ds1 / dao / dx1 / dx bf16 [R][512], dz bf16
 * [R][2048], dqkv bf16 [R][1536] (df, dz, dao, dqkv are the `dy` operands of the layer's four weight gradients; df may alias ds2 and dao ds1
 * when there is no hidden dropout), part1 / part2 fp32 [B][2][512] (per-sequence sums of dy * xhat | dy of the two LayerNorms: svsr_colsum_rows
 * over B rows of 1024 gives the gamma | beta gradients).  layers[0].dx is the gradient of the first layer's input.  Same workspace
 * conventions as svsr_enc_fwd (word B of ws = error flag). */
int svsr_enc_bwd(const void* dy, const svsr_enc_bwd_layer* layers, int n_layers, int B, int S, const unsigned* drop_seed, float p_hidden, float p_attn,
                 void* ws, int64_t ws_bytes, hipStream_t stream) {
    if (dy == nullptr || layers == nullptr || n_layers < 1 || n_layers > 8 || B < 1 || S < 1 || S > TR || ws == nullptr || ws_bytes < svsr_enc_bwd_ws_bytes(B))
        return SVSR_ERR_ARG;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_enc_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL); attr = true; }
    unsigned* cnt = static_cast<unsigned*>(ws);
    const size_t cnt_bytes = (size_t)(((B + 1) * 4 + 255) / 256 * 256);
    hipError_t e = hipMemsetAsync(cnt, 0, cnt_bytes, stream);
    if (e != hipSuccess) return (int)e;
    EncBwdTable t;
    memset(&t, 0, sizeof t);
    for (int l = 0; l < n_layers; ++l) t.L[l] = layers[l];
    bool fill = true;
    svsr_enc_bwd_layer* tab_dev = static_cast<svsr_enc_bwd_layer*>(enc_tab_find(&t, sizeof t, stream, &fill));
    if (tab_dev == nullptr) { tab_dev = reinterpret_cast<svsr_enc_bwd_layer*>(static_cast<char*>(ws) + cnt_bytes); fill = true; }
    if (fill) hipLaunchKernelGGL(k_enc_bwd_table, dim3(1), dim3(256), 0, stream, t, (int)(sizeof(EncBwdTable) / 8), reinterpret_cast<long long*>(tab_dev));
    EncBwdArgs a;
    a.dy = (const bf16_t*)dy; a.Ls = tab_dev;
    a.layers = n_layers; a.S = S;
    const DropArgs dh = svsr_make_drop(drop_seed, 0, p_hidden), da = svsr_make_drop(drop_seed, 0, p_attn);
    a.seed = (dh.seed != nullptr || da.seed != nullptr) ? drop_seed : nullptr;
    a.th_hidden = dh.thresh; a.sc_hidden = dh.scale; a.th_attn = da.thresh; a.sc_attn = da.scale;
    a.err = cnt + B;
    a.host_flag = g_enc_host_flag_dev;
    a.spin_limit = g_enc_spin_limit;
    a.trace = g_enc_trace;
    const int per = enc_clusters_per_launch(stream);
    for (int s0 = 0; s0 < B; s0 += per) {
        a.seq0 = s0; a.nseq = B - s0 < per ? B - s0 : per; a.cnt = cnt + s0;
        hipLaunchKernelGGL(k_enc_bwd, dim3(a.nseq * EH), dim3(256), LDS_TOTAL, stream, a);
    }
    return svsr_check_launch();
}

}  // extern "C"
