// The SyncVSR audio-token head as ONE contraction per direction (gfx950): audio_projection + per-frame cross-entropy over
// (audio_alignment x vq_groups) groups of audio_vocab_size logits, without the logits tensor.
//
// Replaces, for the reference's
//     logits_audio = self.audio_projection(last_hidden_state[:, 1:, :]).float()
//     loss_audio = F.cross_entropy(logits_audio.reshape(-1, self.audio_vocab_size), audio_tokens.flatten())
// (reference LRW/video/src/lightning.py:168-171; README.md:47-53; SURVEY.md section 8 a11 / 8b "svsr_linear_ce_fwd/bwd"), the three launches
// projection GEMM -> logits [B*T][A*G*V] -> k_ce_fwd (and k_ce_bwd in the backward).
//
// Forward (k_linear_ce<false>): a workgroup owns 128 rows (= frames) x one GROUP of V columns.  The contraction is computed TRANSPOSED —
// MFMA A operand = weight rows (logit columns), B operand = hidden rows — so that the accumulator fragment D[column][row] gives a lane one
// ROW of the logits (row = lane & 31) and 16 of every 32 columns in its registers: the row-wise max / sum-of-exponentials / target pick of
// the cross-entropy are register loops plus ONE exchange with lane ^ 32, no LDS, no logits in memory.  V is walked in blocks of 320 columns
// (10 accumulator tiles) with a running (max, sum) per row, so the 640-wide wav2vec2 codec is the same kernel.  Out: lse [R*G] and the row
// losses lse - z_target [R*G], added in a fixed order by svsr_colsum_rows (mean over R*G = the reference's reduction).
// Backward (k_linear_ce<true>): the same contraction again (2.4 GFLOP at the benchmark batch: cheaper than re-reading logits would be to
// keep), then dlogits = gout / (R*G) * (exp(z - lse) - onehot) in bf16, written for the projection's data- and weight-gradient launches.
// Arithmetic: z = fp32 accumulator + bias (the unfused path rounds the logits to bf16 first: this form is closer to the fp32 reference).
#include "common.h"

#define AH_HSWZ(row, chunk) ((row) * 64 + (((chunk) ^ (((row) >> 1) & 7)) << 3))      // [rows][64 k] bf16 tile (128-byte rows), 16-byte chunks XOR-swizzled
#define AH_WSWZ(row, chunk) ((row) * 32 + (((chunk) ^ (((row) >> 2) & 3)) << 3))      // [rows][32 k] bf16 tile (64-byte rows): a ds_read_b128 lane group's
                                                                                       // rows {0-3, 12-15, 20-27} then cover all 16 slots of the 256-byte bank row
#define AH_CB 320              // columns per block (10 MFMA tiles of 32: five per wave)
#define AH_BM 32               // rows per workgroup: its two waves own the two column halves of 160
#define AH_NSW 6               // weight ring: stages of [320 columns][32 k] = 20 KiB, five in flight
#define AH_THREADS 128

struct LinCeArgs {
    const bf16_t* h;           // hidden rows [*, K] bf16, row pitch K
    const bf16_t* w;           // weight [G*V][K] bf16 (nn.Linear layout)
    const float* bias;         // [G*V] or null
    const long* tok;           // [R*G] target indices
    int R, K, G, V;            // R rows (frames), G groups of V columns
    int seq_S, seq_s0, seq_T;  // row r reads hidden row (r / T) * S + s0 + r % T   (T = 0: row r)
    float* lse;                // [R*G]
    float* row_loss;           // forward: [R*G]
    const float* gout;         // backward: d loss / d (mean cross-entropy), device scalar
    bf16_t* dlogits;           // backward: [R][G*V] bf16
};

__device__ unsigned g_ah_zero_page[64];

// s_waitcnt vmcnt(10 * later) + s_barrier with a compile-time immediate (10 LDS-DMA pieces per thread and stage)
__device__ __forceinline__ void ah_wait_barrier(int later) {
#ifdef SVSR_SYNC_DEBUG
    (void)later;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    if (later >= 4) asm volatile("s_waitcnt vmcnt(40) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (later == 3) asm volatile("s_waitcnt vmcnt(30) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// The launch is latency-bound (2.4 GFLOP on 232 workgroups at the benchmark batch): the 32 hidden rows of a workgroup stay in LDS for the
// whole K (32 KiB at K = 512), the weight block streams through a 6-deep ring of 32-deep stages with five in flight (10 LDS-DMA pieces per
// thread and stage), counted vmcnt + one barrier per stage.
template <bool BWD>
__global__ __launch_bounds__(AH_THREADS) void k_linear_ce(const LinCeArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int KT = p.K / 64;                                                    // 64-deep sub-tiles of the resident hidden rows
    bf16_t* sH = reinterpret_cast<bf16_t*>(smem_raw);                           // [KT][64 rows][64 k]
    bf16_t* sW = sH + (size_t)KT * AH_BM * 64;                                  // [AH_NSW][320][32 k]
    float* sBias = reinterpret_cast<float*>(sW + AH_NSW * AH_CB * 32);          // [AH_CB]
    float* sX = sBias + AH_CB;                                                  // [32 rows][3]: the upper column half's (max, sum, target logit)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = 0, chf = wave;
    const int g = blockIdx.y, m0 = blockIdx.x * AH_BM;
    const int nblk = p.V / AH_CB, NS = p.K / 32;

    // ---- hidden rows: KT x (32 rows x 8 chunks) pieces, thread = (chunk slot, row r0 + 16 i) ----
    {
        const int slot = tid & 7, r0 = tid >> 3, csw = slot ^ ((r0 >> 1) & 7);
        const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(g_ah_zero_page) + slot * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = m0 + r0 + 16 * i;
            long src_row = r;
            if (p.seq_T > 0) { const int b = r / p.seq_T; src_row = (long)b * p.seq_S + p.seq_s0 + (r - b * p.seq_T); }
            const bf16_t* hp = r < p.R ? p.h + src_row * p.K + csw * 8 : nullptr;
            for (int kt = 0; kt < KT; ++kt)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(hp != nullptr ? hp + kt * 64 : zero_src),
                                                 (__attribute__((address_space(3))) void*)(sH + (size_t)kt * AH_BM * 64 + (wave * 8 + 16 * i) * 64), 16, 0, 0);
        }
    }
    // weight stage: 320 rows x 4 chunks = 10 pieces per thread; thread = (chunk slot tid & 3, row (tid >> 2) + 32 i)
    const int wslot = tid & 3, wr0 = tid >> 2, wcs = wslot ^ ((wr0 >> 2) & 3);
    // this lane's row of the logits
    const int row = m0 + rt * 32 + (lane & 31), half = lane >> 5;
    const bool row_ok = row < p.R;
    const long tgt = row_ok ? p.tok[(long)row * p.G + g] : 0;
    float lse_row = 0.f, gs = 0.f;
    if (BWD) {
        lse_row = row_ok ? p.lse[(long)row * p.G + g] : 0.f;
        gs = p.gout[0] / (float)((long)p.R * p.G);
    }
    float run_m = -INFINITY, run_s = 0.f, zt = 0.f;

    for (int blk = 0; blk < nblk; ++blk) {
        const int col0 = g * p.V + blk * AH_CB;                                 // first logit column (= weight row) of the block
        if (blk > 0) __syncthreads();                                           // previous block's ring and bias are dead
        for (int c = tid; c < AH_CB; c += AH_THREADS) sBias[c] = p.bias != nullptr ? p.bias[col0 + c] : 0.f;
        const bf16_t* w_ptr = p.w + (long)(col0 + wr0) * p.K + wcs * 8;
        auto stage = [&](int st) {
            bf16_t* dW = sW + (st % AH_NSW) * AH_CB * 32 + wave * 16 * 32;
#pragma unroll
            for (int i = 0; i < 10; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr + (long)i * 32 * p.K + st * 32),
                                                 (__attribute__((address_space(3))) void*)(dW + i * 32 * 32), 16, 0, 0);
        };
        f32x16 acc[5];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int st = 0; st < AH_NSW - 1; ++st)
            if (st < NS) stage(st);
        for (int st = 0; st < NS; ++st) {
            // stage st (and, the first time, the hidden rows in front of it) has landed once at most min(4, stages left) later stages are outstanding
            const int later = NS - 1 - st < AH_NSW - 2 ? NS - 1 - st : AH_NSW - 2;
            ah_wait_barrier(later);
            if (st + AH_NSW - 1 < NS) stage(st + AH_NSW - 1);                   // into the slot read at step st - 1: everybody has passed this step's barrier
            const bf16_t* cW = sW + (st % AH_NSW) * AH_CB * 32;
            const bf16_t* cH = sH + (size_t)(st >> 1) * AH_BM * 64;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 fh = *reinterpret_cast<const bf16x8*>(cH + AH_HSWZ(rt * 32 + (lane & 31), (st & 1) * 4 + ks * 2 + half));
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const bf16x8 fw = *reinterpret_cast<const bf16x8*>(cW + AH_WSWZ(chf * 160 + j * 32 + (lane & 31), ks * 2 + half));
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fh, acc[j], 0, 0, 0);
                }
            }
        }
        // ---- this lane: row `row`, columns col0 + chf*160 + j*32 + 8*q + 4*half + e  (register r = 4 q + e of tile j) ----
        if (!BWD) {
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(sBias + chf * 160 + j * 32 + 8 * q + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float z = acc[j][4 * q + e] + b4[e];
                        acc[j][4 * q + e] = z;
                        m = fmaxf(m, z);
                        if ((long)(blk * AH_CB + chf * 160 + j * 32 + 8 * q + 4 * half + e) == tgt) zt = z;
                    }
                }
            const float mn = fmaxf(run_m, m);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += __expf(acc[j][r] - mn);
            run_s = run_s * __expf(run_m - mn) + s;        // (run_m = -inf, run_s = 0 at the first block: 0 * exp(-inf) = 0)
            run_m = mn;
        } else {
            bf16_t* drow = p.dlogits + (long)(row_ok ? row : 0) * p.G * p.V + col0 + chf * 160;
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(sBias + chf * 160 + j * 32 + 8 * q + 4 * half);
                    float d[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float z = acc[j][4 * q + e] + b4[e];
                        const float pr = __expf(z - lse_row);
                        const bool hit = (long)(blk * AH_CB + chf * 160 + j * 32 + 8 * q + 4 * half + e) == tgt;
                        d[e] = gs * (pr - (hit ? 1.f : 0.f));
                    }
                    uint2 v;
                    v.x = pack2bf(d[0], d[1]);
                    v.y = pack2bf(d[2], d[3]);
                    // lanes l and l + 32 hold columns 8q .. 8q+3 and 8q+4 .. 8q+7 of the same row: one exchange makes 16-byte stores
                    // (every lane ends with {lower half's value, upper half's value}; even q is stored by the lower half, odd q by the upper)
                    const auto sx = __builtin_amdgcn_permlane32_swap(v.x, v.x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(v.y, v.y, false, false);
                    if (row_ok && half == (q & 1)) {
                        u32x4 o;
                        o.x = sx[0]; o.y = sy[0]; o.z = sx[1]; o.w = sy[1];
                        *reinterpret_cast<u32x4*>(drow + j * 32 + 8 * q) = o;
                    }
                }
        }
    }
    if (!BWD) {
        // a row's four partial results — two lane halves x two column-half waves — combined in a fixed order: lanes l, l ^ 32 first, then the waves
        const auto xm = __builtin_amdgcn_permlane32_swap(__float_as_int(run_m), __float_as_int(run_m), false, false);
        const auto xs = __builtin_amdgcn_permlane32_swap(__float_as_int(run_s), __float_as_int(run_s), false, false);
        const auto xz = __builtin_amdgcn_permlane32_swap(__float_as_int(zt), __float_as_int(zt), false, false);
        const float m0v = __int_as_float(xm[0]), m1v = __int_as_float(xm[1]);
        float mm = fmaxf(m0v, m1v);
        float ss = __int_as_float(xs[0]) * __expf(m0v - mm) + __int_as_float(xs[1]) * __expf(m1v - mm);
        float ztv = __int_as_float(xz[0]) + __int_as_float(xz[1]);               // the target column lives in one place only (the others hold 0)
        float* mine = sX + (rt * 32 + (lane & 31)) * 3;
        if (chf == 1 && half == 0) { mine[0] = mm; mine[1] = ss; mine[2] = ztv; }
        __syncthreads();
        if (chf == 0 && half == 0 && row_ok) {
            const float m2 = mine[0], s2 = mine[1];
            const float mt = fmaxf(mm, m2);
            ss = ss * __expf(mm - mt) + s2 * __expf(m2 - mt);
            ztv += mine[2];
            const float lse = mt + __logf(ss);
            const bool t_ok = tgt >= 0 && tgt < p.V;      // a target outside [0, V) (torch raises a device assert) poisons the loss
            p.lse[(long)row * p.G + g] = lse;
            p.row_loss[(long)row * p.G + g] = t_ok ? lse - ztv : __uint_as_float(0x7fc00000u);
        }
    }
}

extern "C" int svsr_colsum_rows(const float* ws, int nrows, int64_t ld, float* out0, int64_t n0, float* out1, int64_t n1, int accumulate,
                                float scale, hipStream_t stream);

static int linear_ce_check(int R, int K, int G, int V, int seq_S, int seq_s0, int seq_T) {
    if (R < 1 || G < 1 || V < AH_CB || V % AH_CB || K < 64 || K % 64 || K > 576) return SVSR_ERR_ARG;       // (the hidden rows stay in LDS: 36 KiB at K = 576)
    if (seq_T < 0 || (seq_T > 0 && (seq_S < seq_s0 + seq_T || seq_s0 < 0))) return SVSR_ERR_ARG;
    return SVSR_OK;
}

static size_t linear_ce_lds(int K) { return ((size_t)(K / 64) * AH_BM * 64 + (size_t)AH_NSW * AH_CB * 32) * sizeof(bf16_t) + (AH_CB + 32 * 3) * sizeof(float); }

extern "C" {

/* 1 when svsr_linear_ce_fwd / _bwd take this shape (V a multiple of 320, K a multiple of 64 up to 576) */
int svsr_linear_ce_ok(int R, int K, int G, int V) { return linear_ce_check(R, K, G, V, 0, 0, 0) == SVSR_OK ? 1 : 0; }

int svsr_linear_ce_fwd(const void* h, const void* w, const float* bias, const int64_t* tok, int R, int K, int G, int V, int seq_S, int seq_s0,
                       int seq_T, float* loss, float* lse, float* row_loss, hipStream_t stream) {
    const int rc0 = linear_ce_check(R, K, G, V, seq_S, seq_s0, seq_T);
    if (rc0 != SVSR_OK || h == nullptr || w == nullptr || tok == nullptr || loss == nullptr || lse == nullptr || row_loss == nullptr) return SVSR_ERR_ARG;
    LinCeArgs a{(const bf16_t*)h, (const bf16_t*)w, bias, (const long*)tok, R, K, G, V, seq_S, seq_s0, seq_T, lse, row_loss, nullptr, nullptr};
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear_ce<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear_ce<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(k_linear_ce<false>, dim3((R + AH_BM - 1) / AH_BM, G), dim3(AH_THREADS), linear_ce_lds(K), stream, a);
    const int rc = svsr_check_launch();
    if (rc != SVSR_OK) return rc;
    const long n = (long)R * G;
    return svsr_colsum_rows(row_loss, (int)n, 1, loss, 1, nullptr, 0, 0, 1.0f / (float)n, stream);     // *loss = mean of the row losses
}

int svsr_linear_ce_bwd(const void* h, const void* w, const float* bias, const int64_t* tok, int R, int K, int G, int V, int seq_S, int seq_s0,
                       int seq_T, const float* lse, const float* gout, void* dlogits, hipStream_t stream) {
    const int rc0 = linear_ce_check(R, K, G, V, seq_S, seq_s0, seq_T);
    if (rc0 != SVSR_OK || h == nullptr || w == nullptr || tok == nullptr || lse == nullptr || gout == nullptr || dlogits == nullptr) return SVSR_ERR_ARG;
    LinCeArgs a{(const bf16_t*)h, (const bf16_t*)w, bias, (const long*)tok, R, K, G, V, seq_S, seq_s0, seq_T, const_cast<float*>(lse), nullptr, gout,
                (bf16_t*)dlogits};
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear_ce<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear_ce<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(k_linear_ce<true>, dim3((R + AH_BM - 1) / AH_BM, G), dim3(AH_THREADS), linear_ce_lds(K), stream, a);
    return svsr_check_launch();
}

}  // extern "C"
