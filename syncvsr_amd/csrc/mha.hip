// Multi-head attention for the LRS model (gfx950, wave64, MFMA 32x32x16 bf16): one kernel family serves
//   * the Conformer's relative-position self-attention   (reference LRS/video/espnet/nets/pytorch_backend/transformer/
//     attention.py:191-278, rel_shift :216-236)          scores[i,j] = ((q_i+u)·k_j + (q_i+v)·p[Lq-1+j-i]) / sqrt(dk)
//   * the decoder's causal self-attention and source attention (attention.py:38-108, decoder_layer.py:60-121)
// with the reference's mask semantics (attention.py:71-78): masked scores are excluded from the softmax and their
// probabilities are zero.  Head width is 64.  A four-wave workgroup owns a 32-row tile of one (batch, head) (mha_coop.h):
//   forward      S = Q·K^T (+ rel-pos term) -> LDS, softmax, P -> HBM (bf16, kept for the backward), ctx = P·V
//   backward/q   dP = dctx·V^T, dS = P∘(dP - rowsum(dP∘P))·scale -> HBM, dq = dS·K (+ dS_shifted·PE)
//   backward/kv  dV = P^T·dctx, dK = dS^T·(q+u)         (one workgroup per 32-key tile)
//   backward/pe  dPE[r] = sum_{b,i} dS[b,i,r-(Lq-1)+i]·(q_i+v)   (one workgroup per (32-row tile of the table, head, clip) + reduce)
// Operands whose contraction index is contiguous in memory (Q, K, V for dP, dctx) are loaded from HBM/L2 straight into
// MFMA fragments (lane l: row l&31, k = (l>>5)*8..+7 = one 16-byte load); operands that are "k-major" (V for P·V, K for
// dS·K, the transposed P/dS) are staged through LDS and gathered.  Sequences here are short (<= ~600 frames), so the
// whole score row of a tile lives in LDS and the softmax is exact (no online rescaling).
#include <stdlib.h>

#include "common.h"

#define MHA_DH 64
#define MHA_VP 66          // LDS pitch (bf16 elements) of staged 64-wide rows
#define MHA_TP 34          // LDS pitch (bf16 elements) of staged 32-wide rows

struct MhaArgs {
    const bf16_t* q; int q_pitch;
    const bf16_t* k; const bf16_t* v; int kv_pitch;
    const bf16_t* pe; int pe_pitch;           // [2*Lq-1][pe_pitch] projected position table, or null
    const float* bias_u; const float* bias_v; // [H*64] (rel-pos only)
    const int* klen;                          // [B] number of valid keys, or null
    int causal;
    int B, H, Lq, Lk, ldp;
    float scale;
    bf16_t* ctx; int ctx_pitch;
    bf16_t* probs;                            // [B*H][Lq][ldp]
    // backward
    const bf16_t* dctx; int dctx_pitch;
    bf16_t* ds;                               // [B*H][Lq][ldp]
    bf16_t* dq; int dq_pitch;
    bf16_t* dq_ac; bf16_t* dq_bd; int aux_pitch;   // rel-pos only: the two summands of dq (for the pos_bias_u / pos_bias_v gradients)
    bf16_t* dk; bf16_t* dv; int dkv_pitch;
    bf16_t* dpe; int dpe_pitch;
    float* pe_part;                           // [B][2*Lq-1][dpe_pitch] fp32 per-batch-item partials of dpe
    DropArgs drop;                            // attention-probability dropout (seed == nullptr: off); element index = position in probs
};

__device__ __forceinline__ bf16x8 ld_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }

// q fragment plus a per-column fp32 bias, rounded back to bf16 (what the reference's bf16 autocast matmul consumes)
__device__ __forceinline__ bf16x8 add_bias_frag(bf16x8 f, const float* bias) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(bf2f((bf16_t)f[e]) + bias[e]);
    return o;
}

// k-major gather: fragment element e = tile[(k0 + (lane>>5)*8 + e) * pitch + col0 + (lane&31)]
__device__ __forceinline__ bf16x8 gather_frag(const bf16_t* tile, int pitch, int k0, int col0, int lane) {
    const bf16_t* base = tile + (k0 + (lane >> 5) * 8) * pitch + col0 + (lane & 31);
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (short)base[e * pitch];
    return f;
}

// 8 consecutive fp32 of an LDS row -> bf16 fragment (16-byte aligned address)
__device__ __forceinline__ bf16x8 f32row_frag(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    bf16x8 f;
    f[0] = (short)f2bf(a[0]); f[1] = (short)f2bf(a[1]); f[2] = (short)f2bf(a[2]); f[3] = (short)f2bf(a[3]);
    f[4] = (short)f2bf(b[0]); f[5] = (short)f2bf(b[1]); f[6] = (short)f2bf(b[2]); f[7] = (short)f2bf(b[3]);
    return f;
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

#include "mha_coop.h"
#include "mha_flash.h"

__global__ __launch_bounds__(256) void k_mha_pe_reduce(const float* __restrict__ part, bf16_t* __restrict__ dpe, int B, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float s = 0.f;
        int b = 0;
        for (; b + 7 < B; b += 8) {          // eight loads in flight, added in clip order
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = part[(long)(b + k) * n + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += t[k];
        }
        for (; b < B; ++b) s += part[(long)b * n + i];
        dpe[i] = f2bf(s);
    }
}

static inline size_t mha_lds_fwd4(int Lk) { return ((size_t)32 * (((Lk + 31) & ~31) + 4) + 4 * 32 * 64 + 64) * sizeof(float) + (size_t)32 * MHA_VP * 2; }
static inline size_t mha_lds_bwd4(int Lk) { return ((size_t)32 * (((Lk + 31) & ~31) + 4) + 2 * 1024) * sizeof(float) + (size_t)32 * MHA_VP * 2; }
#define MHA_MAX_LDS (150 * 1024)

template <typename K>
static inline void mha_allow_lds(K kern) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, MHA_MAX_LDS); }

extern "C" {

int svsr_mha_fwd(const void* q, int q_pitch, const void* k, const void* v, int kv_pitch, const void* pe, int pe_pitch,
                 const float* bias_u, const float* bias_v, const int* klen, int causal, int B, int H, int dh, int Lq, int Lk, int ldp,
                 float scale, void* ctx, int ctx_pitch, void* probs, const unsigned* drop_seed, unsigned drop_site, float drop_p,
                 hipStream_t stream) {
    if (dh != MHA_DH || Lq < 1 || Lk < 1 || ldp < Lk || (q_pitch | kv_pitch | pe_pitch) % 8 != 0) return SVSR_ERR_ARG;
    if (pe != nullptr && (bias_u == nullptr || bias_v == nullptr || Lq != Lk)) return SVSR_ERR_ARG;
    const size_t lds = mha_lds_fwd4(Lk);
    if (lds > MHA_MAX_LDS) return SVSR_ERR_ARG;
    MhaArgs a{};
    a.q = (const bf16_t*)q; a.q_pitch = q_pitch; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.kv_pitch = kv_pitch;
    a.pe = (const bf16_t*)pe; a.pe_pitch = pe_pitch; a.bias_u = bias_u; a.bias_v = bias_v; a.klen = klen; a.causal = causal;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.ldp = ldp; a.scale = scale; a.ctx = (bf16_t*)ctx; a.ctx_pitch = ctx_pitch; a.probs = (bf16_t*)probs;
    a.drop = svsr_make_drop(drop_seed, drop_site, drop_p);
    static bool attr = false;
    if (!attr) { mha_allow_lds(k_mha_fwd4<true>); mha_allow_lds(k_mha_fwd4<false>); attr = true; }
    const dim3 grid((Lq + 31) / 32, B * H);
    if (pe != nullptr) hipLaunchKernelGGL(k_mha_fwd4<true>, grid, dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(k_mha_fwd4<false>, grid, dim3(256), lds, stream, a);
    return svsr_check_launch();
}

int svsr_mha_bwd(const void* dctx, int dctx_pitch, const void* q, int q_pitch, const void* k, const void* v, int kv_pitch,
                 const void* pe, int pe_pitch, const float* bias_u, const float* bias_v, const void* probs, void* ds, int B, int H,
                 int dh, int Lq, int Lk, int ldp, float scale, void* dq, int dq_pitch, void* dq_ac, void* dq_bd, int aux_pitch,
                 void* dk, void* dv, int dkv_pitch, void* dpe, int dpe_pitch, float* pe_part, const unsigned* drop_seed, unsigned drop_site,
                 float drop_p, hipStream_t stream) {
    if (dh != MHA_DH || Lq < 1 || Lk < 1 || ldp < Lk || (q_pitch | kv_pitch | pe_pitch | dctx_pitch) % 8 != 0) return SVSR_ERR_ARG;
    const bool rel = pe != nullptr;
    if (rel && (bias_u == nullptr || bias_v == nullptr || Lq != Lk || dq_ac == nullptr || dq_bd == nullptr || dpe == nullptr || pe_part == nullptr)) return SVSR_ERR_ARG;
    const size_t lds = mha_lds_bwd4(Lk);
    if (lds > MHA_MAX_LDS) return SVSR_ERR_ARG;
    MhaArgs a{};
    a.q = (const bf16_t*)q; a.q_pitch = q_pitch; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.kv_pitch = kv_pitch;
    a.pe = (const bf16_t*)pe; a.pe_pitch = pe_pitch; a.bias_u = bias_u; a.bias_v = bias_v;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.ldp = ldp; a.scale = scale; a.probs = (bf16_t*)const_cast<void*>(probs);
    a.dctx = (const bf16_t*)dctx; a.dctx_pitch = dctx_pitch; a.ds = (bf16_t*)ds; a.dq = (bf16_t*)dq; a.dq_pitch = dq_pitch;
    a.dq_ac = (bf16_t*)dq_ac; a.dq_bd = (bf16_t*)dq_bd; a.aux_pitch = aux_pitch; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.dkv_pitch = dkv_pitch;
    a.dpe = (bf16_t*)dpe; a.dpe_pitch = dpe_pitch; a.pe_part = pe_part;
    a.drop = svsr_make_drop(drop_seed, drop_site, drop_p);
    static bool attr = false;
    if (!attr) { mha_allow_lds(k_mha_bwd_q4<true>); mha_allow_lds(k_mha_bwd_q4<false>); attr = true; }
    const dim3 gq((Lq + 31) / 32, B * H), gk((Lk + 31) / 32, B * H);
    if (rel) {
        const long n = (long)(2 * Lq - 1) * dpe_pitch;
        long blocks = (n + 255) / 256; if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(k_mha_bwd_q4<true>, gq, dim3(256), lds, stream, a);
        hipLaunchKernelGGL(k_mha_bwd_kv4<true>, gk, dim3(256), 0, stream, a);
        hipLaunchKernelGGL(k_mha_bwd_pe4, dim3((2 * Lq - 1 + 31) / 32, H, B), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(k_mha_pe_reduce, dim3((int)blocks), dim3(256), 0, stream, pe_part, (bf16_t*)dpe, B, n);
    } else {
        hipLaunchKernelGGL(k_mha_bwd_q4<false>, gq, dim3(256), lds, stream, a);
        hipLaunchKernelGGL(k_mha_bwd_kv4<false>, gk, dim3(256), 0, stream, a);
    }
    return svsr_check_launch();
}

/* waves per workgroup and workgroups per (clip, head) of the flash kernels: one query tile per wave, at most eight, balanced */
static inline void mhaf_grid(int Lq, int& groups, int& waves, int max_waves = 8) {
    const int nq = (Lq + 31) / 32;
    groups = (nq + max_waves - 1) / max_waves;
    waves = (nq + groups - 1) / groups;
    if (waves < 4) waves = 4;          // (the staging of a key block is dealt out over 256 threads)
}

int64_t svsr_mha_flash_ws_bytes(int H, int Lq) {
    if (H < 1 || Lq < 1) return 0;
    const int LM = 64 + ((8 - Lq % 8) % 8), Rp = (2 * Lq + 32 + LM + 7) / 8 * 8;
    return (int64_t)H * MHA_DH * Rp * 2;
}

int svsr_mha_flash_fwd(const void* q, int q_pitch, const void* k, const void* v, int kv_pitch, const void* pe, int pe_pitch,
                       const float* bias_u, const float* bias_v, const int* klen, int causal, int B, int H, int dh, int Lq, int Lk, int ldp,
                       float scale, void* ctx, int ctx_pitch, float* lse, const unsigned* drop_seed, unsigned drop_site, float drop_p,
                       hipStream_t stream) {
    if (dh != MHA_DH || Lq < 1 || Lk < 1 || ldp < Lk || (q_pitch | kv_pitch | pe_pitch | ctx_pitch) % 8 != 0 || lse == nullptr || ctx == nullptr) return SVSR_ERR_ARG;
    if (pe != nullptr && (bias_u == nullptr || bias_v == nullptr || Lq != Lk)) return SVSR_ERR_ARG;
    MhafArgs p{};
    MhaArgs& a = p.m;
    a.q = (const bf16_t*)q; a.q_pitch = q_pitch; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.kv_pitch = kv_pitch;
    a.pe = (const bf16_t*)pe; a.pe_pitch = pe_pitch; a.bias_u = bias_u; a.bias_v = bias_v; a.klen = klen; a.causal = causal;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.ldp = ldp; a.scale = scale; a.ctx = (bf16_t*)ctx; a.ctx_pitch = ctx_pitch;
    a.drop = svsr_make_drop(drop_seed, drop_site, drop_p);
    p.lse = lse;
    static bool attr = false;
    if (!attr) { mha_allow_lds(k_mhaf_fwd<true>); mha_allow_lds(k_mhaf_fwd<false>); attr = true; }
    int groups, waves;
    mhaf_grid(Lq, groups, waves);
    const dim3 grid(groups, B * H);
    if (pe != nullptr) hipLaunchKernelGGL(k_mhaf_fwd<true>, grid, dim3(64 * waves), mhaf_fwd_lds(waves), stream, p);
    else hipLaunchKernelGGL(k_mhaf_fwd<false>, grid, dim3(64 * waves), mhaf_fwd_lds(waves), stream, p);
    return svsr_check_launch();
}

/* the transposed position table the query pass of svsr_mha_flash_bwd reads, made ahead of time (it depends on pe alone: the forward, or another
 * stream, can make it): ws of svsr_mha_flash_ws_bytes(H, Lq) bytes; hand the same ws to svsr_mha_flash_bwd_parts with bit 2 set */
int svsr_mha_pe_transpose(const void* pe, int pe_pitch, int H, int Lq, void* ws, int64_t ws_bytes, hipStream_t stream) {
    if (pe == nullptr || ws == nullptr || H < 1 || Lq < 1 || pe_pitch % 8 != 0 || ws_bytes < svsr_mha_flash_ws_bytes(H, Lq)) return SVSR_ERR_ARG;
    const int LM = 64 + ((8 - Lq % 8) % 8), Rp = (2 * Lq + 32 + LM + 7) / 8 * 8;
    hipLaunchKernelGGL(k_mhaf_pe_transpose, dim3((Rp + 31) / 32, H * MHA_DH / 32), dim3(256), 0, stream, (const bf16_t*)pe, pe_pitch, 2 * Lq - 1, (bf16_t*)ws, Rp, LM);
    return svsr_check_launch();
}

/* parts: bit 2 = ws already holds the transposed table (svsr_mha_pe_transpose); bit 0 = the query and key passes (dq, dq_ac, dq_bd, dk, dv; probs / ds workspace), bit 1 = the position-table pass (dpe from ds and q:
 * nothing but the weight gradient of linear_pos reads it, so the caller may issue it on another stream once bit 0's launches are done there) */
int svsr_mha_flash_bwd_parts(const void* dctx, int dctx_pitch, const void* ctx, int ctx_pitch, const float* lse, const void* q, int q_pitch, const void* k,
                             const void* v, int kv_pitch, const void* pe, int pe_pitch, const float* bias_u, const float* bias_v, const int* klen, int causal,
                             void* probs, void* ds, int B, int H, int dh, int Lq, int Lk, int ldp, float scale, void* dq, int dq_pitch, void* dq_ac,
                             void* dq_bd, int aux_pitch, void* dk, void* dv, int dkv_pitch, void* dpe, int dpe_pitch, float* pe_part, void* ws,
                             int64_t ws_bytes, const unsigned* drop_seed, unsigned drop_site, float drop_p, int parts, hipStream_t stream) {
    if ((parts & 3) == 0 || (parts & ~7) != 0 || ((parts & 4) && !(parts & 1))) return SVSR_ERR_ARG;
    if (dh != MHA_DH || Lq < 1 || Lk < 1 || ldp < Lk || ldp % 8 != 0 || (q_pitch | kv_pitch | pe_pitch | dctx_pitch | ctx_pitch | dq_pitch | aux_pitch) % 8 != 0) return SVSR_ERR_ARG;
    if (dctx == nullptr || ctx == nullptr || lse == nullptr || probs == nullptr || ds == nullptr || dq == nullptr || dk == nullptr || dv == nullptr) return SVSR_ERR_ARG;
    const bool rel = pe != nullptr;
    if (rel && (bias_u == nullptr || bias_v == nullptr || Lq != Lk || dq_ac == nullptr || dq_bd == nullptr || dpe == nullptr || pe_part == nullptr ||
                ws == nullptr || ws_bytes < svsr_mha_flash_ws_bytes(H, Lq) || dpe_pitch != H * MHA_DH)) return SVSR_ERR_ARG;
    MhafArgs p{};
    MhaArgs& a = p.m;
    a.q = (const bf16_t*)q; a.q_pitch = q_pitch; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.kv_pitch = kv_pitch;
    a.pe = (const bf16_t*)pe; a.pe_pitch = pe_pitch; a.bias_u = bias_u; a.bias_v = bias_v; a.klen = klen; a.causal = causal;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.ldp = ldp; a.scale = scale; a.probs = (bf16_t*)probs;
    a.dctx = (const bf16_t*)dctx; a.dctx_pitch = dctx_pitch; a.ds = (bf16_t*)ds; a.dq = (bf16_t*)dq; a.dq_pitch = dq_pitch;
    a.dq_ac = (bf16_t*)dq_ac; a.dq_bd = (bf16_t*)dq_bd; a.aux_pitch = aux_pitch; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.dkv_pitch = dkv_pitch;
    a.dpe = (bf16_t*)dpe; a.dpe_pitch = dpe_pitch; a.pe_part = pe_part;
    a.drop = svsr_make_drop(drop_seed, drop_site, drop_p);
    p.lse = const_cast<float*>(lse); p.ctx = (const bf16_t*)ctx; p.ctx_in_pitch = ctx_pitch;
    static bool attr = false;
    if (!attr) { mha_allow_lds(k_mhaf_bwd_q<true>); mha_allow_lds(k_mhaf_bwd_q<false>); attr = true; }
    int groups, waves;
    mhaf_grid(Lq, groups, waves);
    const dim3 gq(groups, B * H), gk((Lk + 31) / 32, B * H);
    if (rel) {
        const int LM = 64 + ((8 - Lq % 8) % 8), Rp = (2 * Lq + 32 + LM + 7) / 8 * 8;
        p.pet = (const bf16_t*)ws; p.pet_pitch = Rp; p.pet_lm = LM;
        if (parts & 1) {
            if (!(parts & 4))
                hipLaunchKernelGGL(k_mhaf_pe_transpose, dim3((Rp + 31) / 32, H * MHA_DH / 32), dim3(256), 0, stream, a.pe, pe_pitch, 2 * Lq - 1, (bf16_t*)ws, Rp, LM);
            hipLaunchKernelGGL(k_mhaf_bwd_q<true>, gq, dim3(64 * waves), mhaf_bwd_lds(waves), stream, p);
            hipLaunchKernelGGL(k_mha_bwd_kv4<true>, gk, dim3(256), 0, stream, a);
        }
        if (parts & 2) {
            const long n = (long)(2 * Lq - 1) * dpe_pitch;
            long blocks = (n + 255) / 256; if (blocks > 1024) blocks = 1024;
            hipLaunchKernelGGL(k_mha_bwd_pe4, dim3((2 * Lq - 1 + 31) / 32, H, B), dim3(256), 0, stream, a);
            hipLaunchKernelGGL(k_mha_pe_reduce, dim3((int)blocks), dim3(256), 0, stream, pe_part, (bf16_t*)dpe, B, n);
        }
    } else if (parts & 1) {
        hipLaunchKernelGGL(k_mhaf_bwd_q<false>, gq, dim3(64 * waves), mhaf_bwd_lds(waves), stream, p);
        hipLaunchKernelGGL(k_mha_bwd_kv4<false>, gk, dim3(256), 0, stream, a);
    }
    return svsr_check_launch();
}

int svsr_mha_flash_bwd(const void* dctx, int dctx_pitch, const void* ctx, int ctx_pitch, const float* lse, const void* q, int q_pitch, const void* k,
                       const void* v, int kv_pitch, const void* pe, int pe_pitch, const float* bias_u, const float* bias_v, const int* klen, int causal,
                       void* probs, void* ds, int B, int H, int dh, int Lq, int Lk, int ldp, float scale, void* dq, int dq_pitch, void* dq_ac,
                       void* dq_bd, int aux_pitch, void* dk, void* dv, int dkv_pitch, void* dpe, int dpe_pitch, float* pe_part, void* ws,
                       int64_t ws_bytes, const unsigned* drop_seed, unsigned drop_site, float drop_p, hipStream_t stream) {
    return svsr_mha_flash_bwd_parts(dctx, dctx_pitch, ctx, ctx_pitch, lse, q, q_pitch, k, v, kv_pitch, pe, pe_pitch, bias_u, bias_v, klen, causal, probs, ds, B, H, dh,
                                    Lq, Lk, ldp, scale, dq, dq_pitch, dq_ac, dq_bd, aux_pitch, dk, dv, dkv_pitch, dpe, dpe_pitch, pe_part, ws, ws_bytes, drop_seed,
                                    drop_site, drop_p, 3, stream);
}

}  // extern "C"
