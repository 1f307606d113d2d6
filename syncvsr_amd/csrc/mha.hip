// Multi-head attention for the LRS model (gfx950, wave64, MFMA 32x32x16 bf16): one kernel family serves
//   * the Conformer's relative-position self-attention   (reference LRS/video/espnet/nets/pytorch_backend/transformer/
//     attention.py:191-278, rel_shift :216-236)          scores[i,j] = ((q_i+u)·k_j + (q_i+v)·p[Lq-1+j-i]) / sqrt(dk)
//   * the decoder's causal self-attention and source attention (attention.py:38-108, decoder_layer.py:60-121)
// with the reference's mask semantics (attention.py:71-78): masked scores are excluded from the softmax and their
// probabilities are zero.  Head width is 64.  One wave owns a 32-row tile of one (batch, head):
//   forward      S = Q·K^T (+ rel-pos term) -> LDS, softmax, P -> HBM (bf16, kept for the backward), ctx = P·V
//   backward/q   dP = dctx·V^T, dS = P∘(dP - rowsum(dP∘P))·scale -> HBM, dq = dS·K (+ dS_shifted·PE)
//   backward/kv  dV = P^T·dctx, dK = dS^T·(q+u)         (one wave per 32-key tile)
//   backward/pe  dPE[r] = sum_{b,i} dS[b,i,r-(Lq-1)+i]·(q_i+v)   (one wave per 32-row tile of the position table)
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

// stage 32 rows x 64 columns (bf16) of a row-major matrix into LDS [32][MHA_VP]; rows >= nrows are zero
__device__ __forceinline__ void stage_rows64(bf16_t* dst, const bf16_t* src, long row0, int nrows_total, int pitch, int col0, int lane) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = it * 64 + lane, r = idx >> 3, c8 = idx & 7;
        u32x4 v{0u, 0u, 0u, 0u};
        if (row0 + r < nrows_total && row0 + r >= 0) v = *reinterpret_cast<const u32x4*>(src + (row0 + r) * (long)pitch + col0 + c8 * 8);
        unsigned* d = reinterpret_cast<unsigned*>(dst + r * MHA_VP + c8 * 8);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
template <bool REL>
__global__ __launch_bounds__(64) void k_mha_fwd(const MhaArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int LkS = ((a.Lk + 31) & ~31) + 4;
    float* sS = reinterpret_cast<float*>(smem_raw);                  // [32][LkS]
    float* sBD = sS + 32 * LkS;                                      // [32][64]
    float* sStat = sBD + 32 * 64;                                    // [32][2]
    bf16_t* sV = reinterpret_cast<bf16_t*>(sStat + 64);              // [32][MHA_VP]
    const int lane = threadIdx.x, row = lane & 31, half = lane >> 5;
    const int i0 = blockIdx.x * 32, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int klen = a.klen != nullptr ? a.klen[b] : a.Lk;
    const int qi = min(i0 + row, a.Lq - 1);
    const bf16_t* qrow = a.q + ((long)b * a.Lq + qi) * a.q_pitch + h * MHA_DH + half * 8;
    bf16x8 qu[4], qv[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 f = ld_frag(qrow + kk * 16);
        if (REL) {
            qu[kk] = add_bias_frag(f, a.bias_u + h * MHA_DH + kk * 16 + half * 8);
            qv[kk] = add_bias_frag(f, a.bias_v + h * MHA_DH + kk * 16 + half * 8);
        } else {
            qu[kk] = f;
        }
    }
    // ---- scores -> LDS ----
    for (int j0 = 0; j0 < a.Lk; j0 += 32) {
        const int kj = min(j0 + row, a.Lk - 1);
        const bf16_t* krow = a.k + ((long)b * a.Lk + kj) * a.kv_pitch + h * MHA_DH + half * 8;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qu[kk], ld_frag(krow + kk * 16), acc, 0, 0, 0);
        if (REL) {
            // (q+v)·PE^T over the 63 relative offsets this 32x32 block touches, then read back along the diagonals
            const int rbase = (a.Lq - 1) + j0 - i0 - 31;
            __syncthreads();
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                int pr = rbase + nb * 32 + row;
                pr = pr < 0 ? 0 : (pr > 2 * a.Lq - 2 ? 2 * a.Lq - 2 : pr);
                const bf16_t* prow = a.pe + (long)pr * a.pe_pitch + h * MHA_DH + half * 8;
                f32x16 accb;
#pragma unroll
                for (int r = 0; r < 16; ++r) accb[r] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qv[kk], ld_frag(prow + kk * 16), accb, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) sBD[acc_row(r, lane) * 64 + nb * 32 + row] = accb[r];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = acc_row(r, lane);
                acc[r] += sBD[il * 64 + row - il + 31];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int il = acc_row(r, lane), i = i0 + il, j = j0 + row;
            const bool ok = j < klen && j < a.Lk && (!a.causal || j <= i);
            sS[il * LkS + j0 + row] = ok ? acc[r] * a.scale : -INFINITY;
        }
    }
    __syncthreads();
    // ---- softmax: two lanes per row compute the row statistics, then the whole wave normalises row by row ----
    {
        float m = -INFINITY;
        for (int j = half; j < a.Lk; j += 2) m = fmaxf(m, sS[row * LkS + j]);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float s = 0.f;
        if (m > -INFINITY)
            for (int j = half; j < a.Lk; j += 2) s += __expf(sS[row * LkS + j] - m);
        s += __shfl_xor(s, 32, 64);
        if (half == 0) { sStat[row * 2] = m; sStat[row * 2 + 1] = s > 0.f ? 1.f / s : 0.f; }
    }
    __syncthreads();
    const int LkR = (a.Lk + 31) & ~31;
    const bool drop_on = a.drop.seed != nullptr;
    const unsigned dkey = drop_on ? drop_key(a.drop) : 0u;
    for (int il = 0; il < 32; ++il) {
        const float m = sStat[il * 2], inv = sStat[il * 2 + 1];
        const int i = i0 + il;
        for (int j = lane; j < LkR; j += 64) {
            float p = 0.f;
            if (j < a.Lk && inv > 0.f) p = __expf(sS[il * LkS + j] - m) * inv;
            if (i < a.Lq && j < a.ldp) a.probs[((long)bh * a.Lq + i) * a.ldp + j] = f2bf(p);
            if (drop_on) p = drop_keep(dkey, a.drop.thresh, (unsigned)(((long)bh * a.Lq + i) * a.ldp + j)) ? p * a.drop.scale : 0.f;
            sS[il * LkS + j] = p;
        }
    }
    // ---- ctx = P·V ----
    f32x16 o[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
    for (int j0 = 0; j0 < a.Lk; j0 += 32) {
        __syncthreads();
        stage_rows64(sV, a.v, (long)b * a.Lk + j0, b * a.Lk + a.Lk, a.kv_pitch, h * MHA_DH, lane);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 fp = f32row_frag(sS + row * LkS + j0 + kk * 16 + half * 8);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp, gather_frag(sV, MHA_VP, kk * 16, nb * 32, lane), o[nb], 0, 0, 0);
        }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + acc_row(r, lane);
            if (i < a.Lq) a.ctx[((long)b * a.Lq + i) * a.ctx_pitch + h * MHA_DH + nb * 32 + row] = f2bf(o[nb][r]);
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, query side
// ---------------------------------------------------------------------------------------------------------------------
template <bool REL>
__global__ __launch_bounds__(64) void k_mha_bwd_q(const MhaArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int LkS = ((a.Lk + 31) & ~31) + 4;
    float* sS = reinterpret_cast<float*>(smem_raw);                  // [32][LkS]: dP, then dS
    bf16_t* sV = reinterpret_cast<bf16_t*>(sS + 32 * LkS);           // [32][MHA_VP]
    const int lane = threadIdx.x, row = lane & 31, half = lane >> 5;
    const int i0 = blockIdx.x * 32, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int qi = min(i0 + row, a.Lq - 1);
    const bf16_t* drow = a.dctx + ((long)b * a.Lq + qi) * a.dctx_pitch + h * MHA_DH + half * 8;
    bf16x8 fd[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fd[kk] = ld_frag(drow + kk * 16);
    for (int j0 = 0; j0 < a.Lk; j0 += 32) {
        const int kj = min(j0 + row, a.Lk - 1);
        const bf16_t* vrow = a.v + ((long)b * a.Lk + kj) * a.kv_pitch + h * MHA_DH + half * 8;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd[kk], ld_frag(vrow + kk * 16), acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) sS[acc_row(r, lane) * LkS + j0 + row] = acc[r];
    }
    __syncthreads();
    // dS = P ∘ (dP - sum_j dP∘P) * scale, row by row (coalesced P reads, dS writes); with dropout dP = mask/(1-p) ∘ d(dropped P)
    const int LkR = (a.Lk + 31) & ~31;
    const bool drop_on = a.drop.seed != nullptr;
    const unsigned dkey = drop_on ? drop_key(a.drop) : 0u;
    for (int il = 0; il < 32; ++il) {
        const int i = i0 + il;
        const bool live = i < a.Lq;
        const bf16_t* prow = a.probs + ((long)bh * a.Lq + (live ? i : 0)) * a.ldp;
        float part = 0.f;
        if (live)
            for (int j = lane; j < a.Lk; j += 64) {
                float dp = sS[il * LkS + j];
                if (drop_on) {
                    dp = drop_keep(dkey, a.drop.thresh, (unsigned)(((long)bh * a.Lq + i) * a.ldp + j)) ? dp * a.drop.scale : 0.f;
                    sS[il * LkS + j] = dp;
                }
                part += bf2f(prow[j]) * dp;
            }
        const float dsum = wave_sum(part);
        for (int j = lane; j < LkR; j += 64) {
            float d = 0.f;
            if (live && j < a.Lk) d = bf2f(prow[j]) * (sS[il * LkS + j] - dsum) * a.scale;
            sS[il * LkS + j] = d;
            if (live && j < a.ldp) a.ds[((long)bh * a.Lq + i) * a.ldp + j] = f2bf(d);
        }
    }
    // dq_ac = dS·K
    f32x16 oac[2], obd[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { oac[nb][r] = 0.f; obd[nb][r] = 0.f; }
    for (int j0 = 0; j0 < a.Lk; j0 += 32) {
        __syncthreads();
        stage_rows64(sV, a.k, (long)b * a.Lk + j0, b * a.Lk + a.Lk, a.kv_pitch, h * MHA_DH, lane);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 fs = f32row_frag(sS + row * LkS + j0 + kk * 16 + half * 8);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                oac[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fs, gather_frag(sV, MHA_VP, kk * 16, nb * 32, lane), oac[nb], 0, 0, 0);
        }
    }
    if (REL) {
        // dq_bd[i] = sum_r dS[i][r - (Lq-1) + i] · PE[r]   over the rows r this tile can reach
        int r_lo = (a.Lq - 1) - (i0 + 31), r_hi = (a.Lq - 1) + (a.Lk - 1) - i0;
        if (r_lo < 0) r_lo = 0;
        if (r_hi > 2 * a.Lq - 2) r_hi = 2 * a.Lq - 2;
        for (int r0 = r_lo & ~31; r0 <= r_hi; r0 += 32) {
            __syncthreads();
            stage_rows64(sV, a.pe, r0, 2 * a.Lq - 1, a.pe_pitch, h * MHA_DH, lane);
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 fs;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = r0 + kk * 16 + half * 8 + e - (a.Lq - 1) + i0 + row;
                    fs[e] = (short)f2bf((j >= 0 && j < a.Lk) ? sS[row * LkS + j] : 0.f);
                }
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    obd[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fs, gather_frag(sV, MHA_VP, kk * 16, nb * 32, lane), obd[nb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + acc_row(r, lane);
            if (i >= a.Lq) continue;
            const int col = h * MHA_DH + nb * 32 + row;
            const long rr = (long)b * a.Lq + i;
            a.dq[rr * a.dq_pitch + col] = f2bf(oac[nb][r] + obd[nb][r]);
            if (REL) {
                a.dq_ac[rr * a.aux_pitch + col] = f2bf(oac[nb][r]);
                a.dq_bd[rr * a.aux_pitch + col] = f2bf(obd[nb][r]);
            }
        }
}

// stage a 32 x 32 bf16 block  dst[r][c] = src[(row0 + r) * pitch + col0 + c]  (zero outside [0,nrows) x [0,ncols))
__device__ __forceinline__ void stage_block32(bf16_t* dst, const bf16_t* src, int row0, int nrows, int col0, int ncols, int pitch, int lane) {
    for (int idx = lane; idx < 32 * 32; idx += 64) {
        const int r = idx >> 5, c = idx & 31;
        const int rr = row0 + r, cc = col0 + c;
        dst[r * MHA_TP + c] = (rr >= 0 && rr < nrows && cc >= 0 && cc < ncols) ? src[(long)rr * pitch + cc] : (bf16_t)0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, key/value side: one wave per 32-key tile
// ---------------------------------------------------------------------------------------------------------------------
template <bool REL>
__global__ __launch_bounds__(64) void k_mha_bwd_kv(const MhaArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t sP[32 * MHA_TP], sD[32 * MHA_TP], sDC[32 * MHA_VP], sQ[32 * MHA_VP];
    const int lane = threadIdx.x, row = lane & 31;
    const int j0 = blockIdx.x * 32, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    f32x16 odv[2], odk[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { odv[nb][r] = 0.f; odk[nb][r] = 0.f; }
    for (int i0 = 0; i0 < a.Lq; i0 += 32) {
        __syncthreads();
        stage_block32(sP, a.probs + (long)bh * a.Lq * a.ldp, i0, a.Lq, j0, a.Lk, a.ldp, lane);
        if (a.drop.seed != nullptr) {       // dV = (dropped P)^T · dctx
            const unsigned dkey = drop_key(a.drop);
            for (int idx = lane; idx < 32 * 32; idx += 64) {
                const int r = idx >> 5, c = idx & 31;
                const unsigned e = (unsigned)(((long)bh * a.Lq + i0 + r) * a.ldp + j0 + c);
                sP[r * MHA_TP + c] = drop_keep(dkey, a.drop.thresh, e) ? f2bf(bf2f(sP[r * MHA_TP + c]) * a.drop.scale) : (bf16_t)0;
            }
        }
        stage_block32(sD, a.ds + (long)bh * a.Lq * a.ldp, i0, a.Lq, j0, a.Lk, a.ldp, lane);
        stage_rows64(sDC, a.dctx, (long)b * a.Lq + i0, b * a.Lq + a.Lq, a.dctx_pitch, h * MHA_DH, lane);
        stage_rows64(sQ, a.q, (long)b * a.Lq + i0, b * a.Lq + a.Lq, a.q_pitch, h * MHA_DH, lane);
        __syncthreads();
        if (REL) {      // (q + u), rounded to bf16 as in the forward
            for (int idx = lane; idx < 32 * 64; idx += 64) {
                const int r = idx >> 6, c = idx & 63;
                if (i0 + r < a.Lq) sQ[r * MHA_VP + c] = f2bf(bf2f(sQ[r * MHA_VP + c]) + a.bias_u[h * MHA_DH + c]);
            }
            __syncthreads();
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 fp = gather_frag(sP, MHA_TP, kk * 16, 0, lane);     // A[row = key j][k = query i]
            const bf16x8 fs = gather_frag(sD, MHA_TP, kk * 16, 0, lane);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                odv[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp, gather_frag(sDC, MHA_VP, kk * 16, nb * 32, lane), odv[nb], 0, 0, 0);
                odk[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fs, gather_frag(sQ, MHA_VP, kk * 16, nb * 32, lane), odk[nb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + acc_row(r, lane);
            if (j >= a.Lk) continue;
            const long o = ((long)b * a.Lk + j) * a.dkv_pitch + h * MHA_DH + nb * 32 + row;
            a.dk[o] = f2bf(odk[nb][r]);
            a.dv[o] = f2bf(odv[nb][r]);
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, position table: dPE[r][h*64+d] = sum_b sum_i dS[b,h,i, r-(Lq-1)+i] * (q[b,i,h,d] + v_bias[h,d])
// one wave per (32-row tile of the table, head, batch item) writes an fp32 partial; k_mha_pe_reduce sums over the batch.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_mha_bwd_pe(const MhaArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t sA[32 * MHA_TP], sQ[32 * MHA_VP];
    const int lane = threadIdx.x, row = lane & 31;
    const int r0 = blockIdx.x * 32, h = blockIdx.y;
    f32x16 o[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
    {
        const int b = blockIdx.z;
        const bf16_t* dsb = a.ds + ((long)b * a.H + h) * a.Lq * a.ldp;
        for (int i0 = 0; i0 < a.Lq; i0 += 32) {
            // keys reachable from this (row-tile, query-tile): j = r - (Lq-1) + i
            const int jmin = r0 - (a.Lq - 1) + i0, jmax = r0 + 31 - (a.Lq - 1) + i0 + 31;
            if (jmax < 0 || jmin >= a.Lk) continue;
            __syncthreads();
            for (int idx = lane; idx < 32 * 32; idx += 64) {
                const int il = idx >> 5, rl = idx & 31;
                const int i = i0 + il, j = r0 + rl - (a.Lq - 1) + i;
                sA[il * MHA_TP + rl] = (i < a.Lq && j >= 0 && j < a.Lk) ? dsb[(long)i * a.ldp + j] : (bf16_t)0;
            }
            stage_rows64(sQ, a.q, (long)b * a.Lq + i0, b * a.Lq + a.Lq, a.q_pitch, h * MHA_DH, lane);
            __syncthreads();
            for (int idx = lane; idx < 32 * 64; idx += 64) {
                const int r = idx >> 6, c = idx & 63;
                if (i0 + r < a.Lq) sQ[r * MHA_VP + c] = f2bf(bf2f(sQ[r * MHA_VP + c]) + a.bias_v[h * MHA_DH + c]);
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 fa = gather_frag(sA, MHA_TP, kk * 16, 0, lane);      // A[row = r][k = i]
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, gather_frag(sQ, MHA_VP, kk * 16, nb * 32, lane), o[nb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = r0 + acc_row(r, lane);
            if (rr < 2 * a.Lq - 1) a.pe_part[((long)blockIdx.z * (2 * a.Lq - 1) + rr) * a.dpe_pitch + h * MHA_DH + nb * 32 + row] = o[nb][r];
        }
}

__global__ __launch_bounds__(256) void k_mha_pe_reduce(const float* __restrict__ part, bf16_t* __restrict__ dpe, int B, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += part[(long)b * n + i];
        dpe[i] = f2bf(s);
    }
}

#include "mha_coop.h"

static inline size_t mha_lds_fwd4(int Lk) { return ((size_t)32 * (((Lk + 31) & ~31) + 4) + 4 * 32 * 64 + 64) * sizeof(float) + (size_t)32 * MHA_VP * 2; }
static inline size_t mha_lds_bwd4(int Lk) { return ((size_t)32 * (((Lk + 31) & ~31) + 4) + 2 * 1024) * sizeof(float) + (size_t)32 * MHA_VP * 2; }
static inline bool mha_coop() {
    static const bool v = [] { const char* e = getenv("SVSR_MHA_COOP"); return !(e != nullptr && e[0] == '0'); }();
    return v;
}
static inline size_t mha_lds_fwd(int Lk) { return ((size_t)32 * (((Lk + 31) & ~31) + 4) + 32 * 64 + 64) * sizeof(float) + (size_t)32 * MHA_VP * 2; }
static inline size_t mha_lds_bwd(int Lk) { return ((size_t)32 * (((Lk + 31) & ~31) + 4)) * sizeof(float) + (size_t)32 * MHA_VP * 2; }
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
    const bool coop = mha_coop();
    const size_t lds = coop ? mha_lds_fwd4(Lk) : mha_lds_fwd(Lk);
    if (lds > MHA_MAX_LDS) return SVSR_ERR_ARG;
    MhaArgs a{};
    a.q = (const bf16_t*)q; a.q_pitch = q_pitch; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.kv_pitch = kv_pitch;
    a.pe = (const bf16_t*)pe; a.pe_pitch = pe_pitch; a.bias_u = bias_u; a.bias_v = bias_v; a.klen = klen; a.causal = causal;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.ldp = ldp; a.scale = scale; a.ctx = (bf16_t*)ctx; a.ctx_pitch = ctx_pitch; a.probs = (bf16_t*)probs;
    a.drop = svsr_make_drop(drop_seed, drop_site, drop_p);
    static bool attr = false;
    if (!attr) {
        mha_allow_lds(k_mha_fwd<true>); mha_allow_lds(k_mha_fwd<false>); mha_allow_lds(k_mha_fwd4<true>); mha_allow_lds(k_mha_fwd4<false>);
        attr = true;
    }
    const dim3 grid((Lq + 31) / 32, B * H);
    if (coop) {
        if (pe != nullptr) hipLaunchKernelGGL(k_mha_fwd4<true>, grid, dim3(256), lds, stream, a);
        else hipLaunchKernelGGL(k_mha_fwd4<false>, grid, dim3(256), lds, stream, a);
    } else {
        if (pe != nullptr) hipLaunchKernelGGL(k_mha_fwd<true>, grid, dim3(64), lds, stream, a);
        else hipLaunchKernelGGL(k_mha_fwd<false>, grid, dim3(64), lds, stream, a);
    }
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
    const bool coop = mha_coop();
    const size_t lds = coop ? mha_lds_bwd4(Lk) : mha_lds_bwd(Lk);
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
    if (!attr) {
        mha_allow_lds(k_mha_bwd_q<true>); mha_allow_lds(k_mha_bwd_q<false>); mha_allow_lds(k_mha_bwd_q4<true>); mha_allow_lds(k_mha_bwd_q4<false>);
        attr = true;
    }
    const dim3 gq((Lq + 31) / 32, B * H), gk((Lk + 31) / 32, B * H);
    if (coop) {
        const long n = (long)(2 * Lq - 1) * dpe_pitch;
        long blocks = (n + 255) / 256; if (blocks > 1024) blocks = 1024;
        if (rel) {
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
    if (rel) {
        hipLaunchKernelGGL(k_mha_bwd_q<true>, gq, dim3(64), lds, stream, a);
        hipLaunchKernelGGL(k_mha_bwd_kv<true>, gk, dim3(64), 0, stream, a);
        hipLaunchKernelGGL(k_mha_bwd_pe, dim3((2 * Lq - 1 + 31) / 32, H, B), dim3(64), 0, stream, a);
        const long n = (long)(2 * Lq - 1) * dpe_pitch;
        long blocks = (n + 255) / 256; if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(k_mha_pe_reduce, dim3((int)blocks), dim3(256), 0, stream, pe_part, (bf16_t*)dpe, B, n);
    } else {
        hipLaunchKernelGGL(k_mha_bwd_q<false>, gq, dim3(64), lds, stream, a);
        hipLaunchKernelGGL(k_mha_bwd_kv<false>, gk, dim3(64), 0, stream, a);
    }
    return svsr_check_launch();
}

}  // extern "C"
