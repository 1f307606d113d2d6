// Transformer-encoder passes that are not plain contractions (gfx950, wave64):
//   embeddings (+position +token-type) -> LayerNorm; residual add -> LayerNorm; softmax attention for short
//   sequences (one workgroup per (batch, head), everything LDS-resident); bias/GELU backward with column sums.
// Replaces HF BertModel's BertEmbeddings / BertSelfAttention / BertSelfOutput / BertOutput non-GEMM work as
// reached from reference LRW/video/src/lightning.py:92,152-156.  LayerNorm eps is
// 1e-12 (BertConfig default), far below bf16 resolution, so all statistics are fp32.
#include "common.h"

#define LN_MAXV 4   // up to 4 x (64 lanes x 8) = 2048 columns per row

// ---------------------------------------------------------------------------------------------------------
// y = LayerNorm(a + r) * gamma + beta ; one wave per row.  mean/rstd saved for the backward.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_add_ln_fwd(const bf16_t* __restrict__ a, const bf16_t* __restrict__ r,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    bf16_t* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                    int R, int D, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane owns columns (i*64 + lane)*8 .. +7 for i < LN_MAXV; any D % 8 == 0 up to 2048 (512 BERT, 768 Conformer)
    for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
        float v[LN_MAXV][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            if ((i * 64 + lane) * 8 < D) {
                const long o = ((long)row * D + (i * 64 + lane) * 8) >> 3;
                float fa[8], fr[8];
                unpack8(reinterpret_cast<const u32x4*>(a)[o], fa);
                if (r != nullptr) unpack8(reinterpret_cast<const u32x4*>(r)[o], fr);
#pragma unroll
                for (int k = 0; k < 8; ++k) { v[i][k] = fa[k] + (r != nullptr ? fr[k] : 0.f); s += v[i][k]; }
            }
        }
        const float mu = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i)
            if ((i * 64 + lane) * 8 < D)
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mu; q += d * d; }
        const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            if ((i * 64 + lane) * 8 < D) {
                const int c0 = (i * 64 + lane) * 8;
                float o[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = (v[i][k] - mu) * rs * gamma[c0 + k] + beta[c0 + k];
                reinterpret_cast<u32x4*>(y)[((long)row * D + c0) >> 3] = pack8(o);
            }
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// backward: ds = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)) for both a and r; dgamma += dy*xhat, dbeta += dy
__global__ __launch_bounds__(256) void k_add_ln_bwd(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ a,
                                                    const bf16_t* __restrict__ r, const float* __restrict__ gamma,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    bf16_t* __restrict__ ds, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                    int R, int D, const bf16_t* __restrict__ addend) {
    extern __shared__ float sred_dyn[];          // [4 waves][2][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane owns columns (i*64 + lane)*8 .. +7 for i < LN_MAXV; any D % 8 == 0 up to 2048 (512 BERT, 768 Conformer)
    float ag[LN_MAXV][8], ab[LN_MAXV][8];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) { ag[i][k] = 0.f; ab[i][k] = 0.f; }
    for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float xh[LN_MAXV][8], gd[LN_MAXV][8];
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            if ((i * 64 + lane) * 8 < D) {
                const int c0 = (i * 64 + lane) * 8;
                const long o = ((long)row * D + c0) >> 3;
                float fa[8], fr[8], fd[8];
                unpack8(reinterpret_cast<const u32x4*>(a)[o], fa);
                if (r != nullptr) unpack8(reinterpret_cast<const u32x4*>(r)[o], fr);
                unpack8(reinterpret_cast<const u32x4*>(dy)[o], fd);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    xh[i][k] = (fa[k] + (r != nullptr ? fr[k] : 0.f) - mu) * rs;
                    gd[i][k] = fd[k] * gamma[c0 + k];
                    m1 += gd[i][k];
                    m2 += gd[i][k] * xh[i][k];
                    ag[i][k] += fd[k] * xh[i][k];
                    ab[i][k] += fd[k];
                }
            }
        }
        m1 = wave_sum(m1) / (float)D;
        m2 = wave_sum(m2) / (float)D;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            if ((i * 64 + lane) * 8 < D) {
                float o[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = rs * (gd[i][k] - m1 - xh[i][k] * m2);
                if (addend != nullptr) {       // pre-LN residual stream: grad(x) = grad through LN + grad of the skip path
                    float ad[8];
                    unpack8(reinterpret_cast<const u32x4*>(addend)[((long)row * D + (i * 64 + lane) * 8) >> 3], ad);
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] += ad[k];
                }
                reinterpret_cast<u32x4*>(ds)[((long)row * D + (i * 64 + lane) * 8) >> 3] = pack8(o);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
        if ((i * 64 + lane) * 8 < D)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                sred_dyn[(wave * 2 + 0) * D + (i * 64 + lane) * 8 + k] = ag[i][k];
                sred_dyn[(wave * 2 + 1) * D + (i * 64 + lane) * 8 + k] = ab[i][k];
            }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        atomicAdd(dgamma + c, sred_dyn[0 * D + c] + sred_dyn[2 * D + c] + sred_dyn[4 * D + c] + sred_dyn[6 * D + c]);
        atomicAdd(dbeta + c, sred_dyn[1 * D + c] + sred_dyn[3 * D + c] + sred_dyn[5 * D + c] + sred_dyn[7 * D + c]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// BERT embeddings on inputs_embeds:  e[b,0] = cls, e[b,1+t] = feats[b,t];  y = LN(e + pos[s] + type[0])
// `sum_out` keeps the pre-norm sum in bf16 so the backward shares k_add_ln_bwd's arithmetic.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_embed_ln_fwd(const bf16_t* __restrict__ feats, const float* __restrict__ cls,
                                                      const float* __restrict__ pos, const float* __restrict__ type0,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      bf16_t* __restrict__ sum_out, bf16_t* __restrict__ y,
                                                      float* __restrict__ mean, float* __restrict__ rstd, int B, int S, int D, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane owns columns (i*64 + lane)*8 .. +7 for i < LN_MAXV; any D % 8 == 0 up to 2048 (512 BERT, 768 Conformer)
    const int R = B * S;
    for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
        const int b = row / S, s_ = row - b * S;
        float v[LN_MAXV][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            if ((i * 64 + lane) * 8 < D) {
                const int c0 = (i * 64 + lane) * 8;
                float e[8];
                if (s_ == 0) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) e[k] = bf2f(f2bf(cls[c0 + k]));   // the encoder input is a bf16 tensor
                } else {
                    unpack8(reinterpret_cast<const u32x4*>(feats)[(((long)b * (S - 1) + s_ - 1) * D + c0) >> 3], e);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    // keep the bf16-rounded sum: the backward recomputes xhat from it
                    v[i][k] = bf2f(f2bf(e[k] + pos[(long)s_ * D + c0 + k] + type0[c0 + k]));
                    s += v[i][k];
                }
                reinterpret_cast<u32x4*>(sum_out)[((long)row * D + c0) >> 3] = pack8(v[i]);
            }
        }
        const float mu = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i)
            if ((i * 64 + lane) * 8 < D)
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mu; q += d * d; }
        const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            if ((i * 64 + lane) * 8 < D) {
                const int c0 = (i * 64 + lane) * 8;
                float o[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = (v[i][k] - mu) * rs * gamma[c0 + k] + beta[c0 + k];
                reinterpret_cast<u32x4*>(y)[((long)row * D + c0) >> 3] = pack8(o);
            }
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// scatter of the embedding-sum gradient ds [B*S][D] (bf16):  dfeats = ds[:,1:], dcls += sum_b ds[b,0],
// dpos[s] += sum_b ds[b,s], dtype0 += sum ds.   One thread per 8 columns, loops over the batch.
__global__ __launch_bounds__(256) void k_embed_bwd_scatter(const bf16_t* __restrict__ ds, bf16_t* __restrict__ dfeats,
                                                           float* __restrict__ dcls, float* __restrict__ dpos,
                                                           float* __restrict__ dtype0, int B, int S, int D) {
    const int cv = D >> 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;       // over S * cv
    if (idx >= S * cv) return;
    const int s_ = idx / cv, c0 = (idx - s_ * cv) * 8;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int b = 0; b < B; ++b) {
        const u32x4 raw = reinterpret_cast<const u32x4*>(ds)[(((long)b * S + s_) * D + c0) >> 3];
        float f[8];
        unpack8(raw, f);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += f[k];
        if (s_ > 0) reinterpret_cast<u32x4*>(dfeats)[(((long)b * (S - 1) + s_ - 1) * D + c0) >> 3] = raw;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        dpos[(long)s_ * D + c0 + k] += acc[k];              // each (s, c) is owned by exactly one thread
        atomicAdd(dtype0 + c0 + k, acc[k]);
        if (s_ == 0) dcls[c0 + k] += acc[k];
    }
}

// ---------------------------------------------------------------------------------------------------------
// softmax attention, S <= 64, head dim 64: one workgroup per (b, h).  qkv: [B*S][3*D] (q | k | v), ctx: [B*S][D].
// ---------------------------------------------------------------------------------------------------------
#define AT_DH 64
#define AT_LD 65

__global__ __launch_bounds__(256) void k_attn_fwd(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                  bf16_t* __restrict__ probs, int B, int S, int H, float scale) {
    extern __shared__ float sm[];
    float* sQ = sm;
    float* sK = sQ + S * AT_LD;
    float* sV = sK + S * AT_LD;
    float* sP = sV + S * AT_LD;          // [S][S+1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int D = H * AT_DH;
    for (int e = tid; e < S * 8 * 3; e += 256) {
        const int which = e / (S * 8), rem = e - which * (S * 8);
        const int i = rem >> 3, c = (rem & 7) * 8;
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(qkv + ((long)(b * S + i) * 3 + which) * D + h * AT_DH + c), f);
        float* dst = (which == 0 ? sQ : which == 1 ? sK : sV) + i * AT_LD + c;
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = f[k];
    }
    __syncthreads();
    for (int e = tid; e < S * S; e += 256) {
        const int i = e / S, j = e - i * S;
        float acc = 0.f;
#pragma unroll 16
        for (int d = 0; d < AT_DH; ++d) acc += sQ[i * AT_LD + d] * sK[j * AT_LD + d];
        sP[i * (S + 1) + j] = acc * scale;
    }
    __syncthreads();
    for (int i = wave; i < S; i += 4) {
        const float v = lane < S ? sP[i * (S + 1) + lane] : -INFINITY;
        const float m = wave_max(v);
        const float ex = lane < S ? __expf(v - m) : 0.f;
        const float sum = wave_sum(ex);
        const float pr = ex / sum;
        if (lane < S) {
            sP[i * (S + 1) + lane] = pr;
            if (probs != nullptr) probs[((long)blockIdx.x * S + i) * S + lane] = f2bf(pr);
        }
    }
    __syncthreads();
    for (int e = tid; e < S * AT_DH; e += 256) {
        const int i = e >> 6, d = e & 63;
        float acc = 0.f;
        for (int j = 0; j < S; ++j) acc += sP[i * (S + 1) + j] * sV[j * AT_LD + d];
        ctx[(long)(b * S + i) * D + h * AT_DH + d] = f2bf(acc);
    }
}

__global__ __launch_bounds__(256) void k_attn_bwd(const bf16_t* __restrict__ dctx, const bf16_t* __restrict__ qkv,
                                                  const bf16_t* __restrict__ probs, bf16_t* __restrict__ dqkv,
                                                  int B, int S, int H, float scale) {
    extern __shared__ float sm[];
    float* sQ = sm;
    float* sK = sQ + S * AT_LD;
    float* sV = sK + S * AT_LD;
    float* sO = sV + S * AT_LD;          // dO
    float* sP = sO + S * AT_LD;          // [S][S+1]
    float* sD = sP + S * (S + 1);        // dS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int D = H * AT_DH;
    for (int e = tid; e < S * 8 * 4; e += 256) {
        const int which = e / (S * 8), rem = e - which * (S * 8);
        const int i = rem >> 3, c = (rem & 7) * 8;
        float f[8];
        if (which < 3) unpack8(*reinterpret_cast<const u32x4*>(qkv + ((long)(b * S + i) * 3 + which) * D + h * AT_DH + c), f);
        else unpack8(*reinterpret_cast<const u32x4*>(dctx + (long)(b * S + i) * D + h * AT_DH + c), f);
        float* dst = (which == 0 ? sQ : which == 1 ? sK : which == 2 ? sV : sO) + i * AT_LD + c;
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = f[k];
    }
    for (int e = tid; e < S * S; e += 256) {
        const int i = e / S, j = e - i * S;
        sP[i * (S + 1) + j] = bf2f(probs[((long)blockIdx.x * S + i) * S + j]);
    }
    __syncthreads();
    for (int e = tid; e < S * S; e += 256) {       // dP = dO V^T
        const int i = e / S, j = e - i * S;
        float acc = 0.f;
#pragma unroll 16
        for (int d = 0; d < AT_DH; ++d) acc += sO[i * AT_LD + d] * sV[j * AT_LD + d];
        sD[i * (S + 1) + j] = acc;
    }
    __syncthreads();
    for (int i = wave; i < S; i += 4) {            // dS = P * (dP - sum_j dP*P) * scale
        const float pr = lane < S ? sP[i * (S + 1) + lane] : 0.f;
        const float dp = lane < S ? sD[i * (S + 1) + lane] : 0.f;
        const float rs = wave_sum(pr * dp);
        if (lane < S) sD[i * (S + 1) + lane] = pr * (dp - rs) * scale;
    }
    __syncthreads();
    for (int e = tid; e < S * AT_DH; e += 256) {
        const int i = e >> 6, d = e & 63;
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int j = 0; j < S; ++j) {
            dq += sD[i * (S + 1) + j] * sK[j * AT_LD + d];
            dk += sD[j * (S + 1) + i] * sQ[j * AT_LD + d];
            dv += sP[j * (S + 1) + i] * sO[j * AT_LD + d];
        }
        bf16_t* o = dqkv + (long)(b * S + i) * 3 * D + h * AT_DH + d;
        o[0] = f2bf(dq);
        o[D] = f2bf(dk);
        o[2 * D] = f2bf(dv);
    }
}

// ---------------------------------------------------------------------------------------------------------
// dz = dy * gelu'(z) (optional), db[n] += sum_rows dz      — thread owns 8 columns, block covers a row slab
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bias_act_bwd(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ z,
                                                      bf16_t* __restrict__ dz, float* __restrict__ db, int R, int N, int n_valid,
                                                      int ld, int rows_per_block, int act, float gscale) {
    // block = 8 row lanes x 32 column vectors: a wave reads 2 rows x 512 contiguous bytes per step
    __shared__ float sred[8][32][8];
    const int cv = N >> 3;
    const int cvi = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int v = blockIdx.x * 32 + cvi;
    const int r0 = blockIdx.y * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > R) r1 = R;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    if (v < cv) {
        for (int r = r0 + rl; r < r1; r += 8) {
            float g[8];
            unpack8(*reinterpret_cast<const u32x4*>(dy + (long)r * ld + v * 8), g);
            if (z != nullptr) {
                float zz[8];
                unpack8(*reinterpret_cast<const u32x4*>(z + (long)r * ld + v * 8), zz);
#pragma unroll
                for (int k = 0; k < 8; ++k) g[k] = act == 2 ? (zz[k] > 0.f ? g[k] * gscale : 0.f) : g[k] * gelu_erf_grad(zz[k]) * gscale;
                *reinterpret_cast<u32x4*>(dz + (long)r * ld + v * 8) = pack8(g);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += g[k];
        }
    }
    if (db == nullptr) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) sred[rl][cvi][k] = acc[k];
    __syncthreads();
    // 256 threads = 32 vectors x 8 columns: each sums the 8 row lanes of one column
    const int c = threadIdx.x;
    const int vv = c >> 3, kk = c & 7;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += sred[i][vv][kk];
    const int col = (blockIdx.x * 32 + vv) * 8 + kk;
    if (col < n_valid) atomicAdd(db + col, s);
}

extern "C" {

int svsr_add_ln_fwd(const void* a, const void* r, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                    int R, int D, float eps, hipStream_t stream) {
    if (D % 8 != 0 || D > 512 * LN_MAXV) return SVSR_ERR_ARG;
    int grid = (R + 3) / 4; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_add_ln_fwd, dim3(grid), dim3(256), 0, stream, (const bf16_t*)a, (const bf16_t*)r, gamma, beta, (bf16_t*)y,
                       mean, rstd, R, D, eps);
    return svsr_check_launch();
}

int svsr_add_ln_bwd(const void* dy, const void* a, const void* r, const float* gamma, const float* mean, const float* rstd,
                    void* ds, float* dgamma, float* dbeta, int R, int D, const void* addend, hipStream_t stream) {
    if (D % 8 != 0 || D > 512 * LN_MAXV) return SVSR_ERR_ARG;
    // every workgroup ends with 2*D fp32 atomics (dgamma/dbeta); measured flat between 4 and 16 rows per workgroup at
    // 2,400 x 768, slower at 64 (too few workgroups) — 16 keeps the atomics few
    static const int rpb = [] { const char* e = getenv("SVSR_LN_RPB"); return e ? atoi(e) : 16; }();
    int grid = (R + rpb - 1) / rpb; if (grid > 512) grid = 512;
    hipLaunchKernelGGL(k_add_ln_bwd, dim3(grid), dim3(256), (size_t)8 * D * sizeof(float), stream, (const bf16_t*)dy, (const bf16_t*)a, (const bf16_t*)r, gamma,
                       mean, rstd, (bf16_t*)ds, dgamma, dbeta, R, D, (const bf16_t*)addend);
    return svsr_check_launch();
}

int svsr_embed_ln_fwd(const void* feats, const float* cls, const float* pos, const float* type0, const float* gamma,
                      const float* beta, void* sum_out, void* y, float* mean, float* rstd, int B, int S, int D, float eps,
                      hipStream_t stream) {
    if (D % 512 != 0 || D > 512 * LN_MAXV || S < 2) return SVSR_ERR_ARG;
    int grid = (B * S + 3) / 4; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_embed_ln_fwd, dim3(grid), dim3(256), 0, stream, (const bf16_t*)feats, cls, pos, type0, gamma, beta,
                       (bf16_t*)sum_out, (bf16_t*)y, mean, rstd, B, S, D, eps);
    return svsr_check_launch();
}

int svsr_embed_bwd_scatter(const void* ds, void* dfeats, float* dcls, float* dpos, float* dtype0, int B, int S, int D,
                           hipStream_t stream) {
    if (D % 8 != 0) return SVSR_ERR_ARG;
    const int n = S * (D / 8);
    hipLaunchKernelGGL(k_embed_bwd_scatter, dim3((n + 255) / 256), dim3(256), 0, stream, (const bf16_t*)ds, (bf16_t*)dfeats, dcls,
                       dpos, dtype0, B, S, D);
    return svsr_check_launch();
}

int svsr_attn_fwd(const void* qkv, void* ctx, void* probs, int B, int S, int H, int dh, float scale, hipStream_t stream) {
    if (dh != AT_DH || S < 1 || S > 64) return SVSR_ERR_ARG;
    const size_t lds = ((size_t)3 * S * AT_LD + (size_t)S * (S + 1)) * sizeof(float);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); attr = true; }
    hipLaunchKernelGGL(k_attn_fwd, dim3(B * H), dim3(256), lds, stream, (const bf16_t*)qkv, (bf16_t*)ctx, (bf16_t*)probs, B, S, H, scale);
    return svsr_check_launch();
}

int svsr_attn_bwd(const void* dctx, const void* qkv, const void* probs, void* dqkv, int B, int S, int H, int dh, float scale,
                  hipStream_t stream) {
    if (dh != AT_DH || S < 1 || S > 64) return SVSR_ERR_ARG;
    const size_t lds = ((size_t)4 * S * AT_LD + (size_t)2 * S * (S + 1)) * sizeof(float);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); attr = true; }
    hipLaunchKernelGGL(k_attn_bwd, dim3(B * H), dim3(256), lds, stream, (const bf16_t*)dctx, (const bf16_t*)qkv, (const bf16_t*)probs,
                       (bf16_t*)dqkv, B, S, H, scale);
    return svsr_check_launch();
}

int svsr_bias_act_bwd(const void* dy, const void* z, void* dz, float* db, int R, int N, int n_valid, int ld, int act, float gscale,
                      hipStream_t stream) {
    if (N % 8 != 0 || ld % 8 != 0) return SVSR_ERR_ARG;
    const int cv = N / 8;
    const int col_blocks = (cv + 31) / 32;
    int splits = (R + 63) / 64;                                  // ~64 rows (8 per thread) per block ...
    while (splits > 1 && col_blocks * splits > 2048) splits = (splits + 1) / 2;
    const int rpb = (R + splits - 1) / splits;
    splits = (R + rpb - 1) / rpb;
    hipLaunchKernelGGL(k_bias_act_bwd, dim3(col_blocks, splits), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)z,
                       (bf16_t*)dz, db, R, N, n_valid, ld, rpb, act, gscale);
    return svsr_check_launch();
}

}  // extern "C"
