// Transformer-encoder passes that are not plain contractions (gfx950, wave64):
//   embeddings (+position +token-type) -> LayerNorm; residual add -> LayerNorm; bias/GELU backward with column sums.
// (Attention itself is csrc/mha.hip.)  Parameter gradients that are column sums over the rows (LayerNorm gamma/beta, biases,
// the token-type embedding) are written as one partial row per workgroup and added by svsr_colsum_rows in a fixed order.
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
                for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mu; q = __builtin_fmaf(d, d, q); }      // (explicit: csrc/enc_fused.hip repeats this arithmetic bit for bit)
        const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            if ((i * 64 + lane) * 8 < D) {
                const int c0 = (i * 64 + lane) * 8;
                float o[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = __builtin_fmaf((v[i][k] - mu) * rs, gamma[c0 + k], beta[c0 + k]);
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
                                                    bf16_t* __restrict__ ds, float* __restrict__ part,
                                                    int R, int D, const bf16_t* __restrict__ addend,
                                                    bf16_t* __restrict__ ds2, float alpha2, DropArgs drop2) {
    // ds2 (optional): the gradient handed to the residual BRANCH that ends in this sum — x' = x + alpha2 * dropout(branch(..)) has
    // d branch = alpha2 * mask / (1 - p) * d x' — written from the bf16-rounded ds exactly as svsr_scale_bf16 would compute it from ds
    // (same element indices for the mask): the separate pass over [R][D] that the sentence-level model ran 60 times per step
    extern __shared__ float sred_dyn[];          // [4 waves][2][D]
    const unsigned key2 = (ds2 != nullptr && drop2.seed != nullptr) ? drop_key(drop2) : 0u;
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
        u32x4 adv[LN_MAXV];           // the skip-path gradient: requested with the other operands, not behind the two row reductions
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            if ((i * 64 + lane) * 8 < D) {
                const int c0 = (i * 64 + lane) * 8;
                const long o = ((long)row * D + c0) >> 3;
                float fa[8], fr[8], fd[8];
                if (addend != nullptr) adv[i] = reinterpret_cast<const u32x4*>(addend)[o];
                unpack8(reinterpret_cast<const u32x4*>(a)[o], fa);
                if (r != nullptr) unpack8(reinterpret_cast<const u32x4*>(r)[o], fr);
                unpack8(reinterpret_cast<const u32x4*>(dy)[o], fd);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    // (multiply-adds spelled out: csrc/enc_fused.hip's ln_bwd8 must round the same way, see common.h normal_cdf_exp)
                    xh[i][k] = (fa[k] + (r != nullptr ? fr[k] : 0.f) - mu) * rs;
                    gd[i][k] = fd[k] * gamma[c0 + k];
                    asm volatile("" : "+v"(gd[i][k]));       // a rounded product: not to be fused into the sums below at the compiler's choice
                    m1 += gd[i][k];
                    m2 = __builtin_fmaf(gd[i][k], xh[i][k], m2);
                    ag[i][k] = __builtin_fmaf(fd[k], xh[i][k], ag[i][k]);
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
                for (int k = 0; k < 8; ++k) o[k] = rs * __builtin_fmaf(-xh[i][k], m2, gd[i][k] - m1);
                if (addend != nullptr) {       // pre-LN residual stream: grad(x) = grad through LN + grad of the skip path
                    float ad[8];
                    unpack8(adv[i], ad);
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] += ad[k];
                }
                const long vi = ((long)row * D + (i * 64 + lane) * 8) >> 3;
                const u32x4 packed = pack8(o);
                reinterpret_cast<u32x4*>(ds)[vi] = packed;
                if (ds2 != nullptr) {
                    float f[8];
                    unpack8(packed, f);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if (drop2.seed != nullptr) f[k] = drop_keep(key2, drop2.thresh, (unsigned)(vi * 8 + k)) ? f[k] * drop2.scale : 0.f;
                        f[k] *= alpha2;
                    }
                    reinterpret_cast<u32x4*>(ds2)[vi] = pack8(f);
                }
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
    float* row = part + (long)blockIdx.x * 2 * D;          // this workgroup's partial [dgamma | dbeta]
    for (int c = threadIdx.x; c < D; c += 256) {
        row[c] = ((sred_dyn[0 * D + c] + sred_dyn[2 * D + c]) + sred_dyn[4 * D + c]) + sred_dyn[6 * D + c];
        row[D + c] = ((sred_dyn[1 * D + c] + sred_dyn[3 * D + c]) + sred_dyn[5 * D + c]) + sred_dyn[7 * D + c];
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
                                                      float* __restrict__ mean, float* __restrict__ rstd, int B, int S, int D, float eps,
                                                      DropArgs din, DropArgs dout) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // emb_dropout on cat(cls, feats) (lightning.py:150) and BertEmbeddings' dropout on the LayerNorm output: element index =
    // position in the [B*S][D] tensor
    const bool on_in = din.seed != nullptr, on_out = dout.seed != nullptr;
    const unsigned key_in = on_in ? drop_key(din) : 0u, key_out = on_out ? drop_key(dout) : 0u;
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
                if (on_in) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) e[k] = drop_keep(key_in, din.thresh, (unsigned)((long)row * D + c0 + k)) ? e[k] * din.scale : 0.f;
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
                if (on_out) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = drop_keep(key_out, dout.thresh, (unsigned)((long)row * D + c0 + k)) ? o[k] * dout.scale : 0.f;
                }
                reinterpret_cast<u32x4*>(y)[((long)row * D + c0) >> 3] = pack8(o);
            }
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// scatter of the embedding-sum gradient ds [B*S][D] (bf16):  dfeats = ds[:,1:], dcls += sum_b ds[b,0],
// dpos[s] += sum_b ds[b,s]; part[s] = sum_b ds[b,s] (the token-type gradient is its column sum, added by svsr_colsum_rows).
// One thread per 8 columns, loops over the batch.
__global__ __launch_bounds__(256) void k_embed_bwd_scatter(const bf16_t* __restrict__ ds, bf16_t* __restrict__ dfeats,
                                                           float* __restrict__ dcls, float* __restrict__ dpos,
                                                           float* __restrict__ part, int B, int S, int D, DropArgs din) {
    const bool on_in = din.seed != nullptr;          // emb_dropout mask of the forward, regenerated: applies to dfeats / dcls only
    const unsigned key_in = on_in ? drop_key(din) : 0u;
    const int cv = D >> 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;       // over S * cv
    if (idx >= S * cv) return;
    const int s_ = idx / cv, c0 = (idx - s_ * cv) * 8;
    float acc[8], accm[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { acc[k] = 0.f; accm[k] = 0.f; }
    for (int b = 0; b < B; ++b) {
        const long e0 = ((long)b * S + s_) * D + c0;
        u32x4 raw = reinterpret_cast<const u32x4*>(ds)[e0 >> 3];
        float f[8];
        unpack8(raw, f);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += f[k];
        if (on_in) {
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = drop_keep(key_in, din.thresh, (unsigned)(e0 + k)) ? f[k] * din.scale : 0.f;
            raw = pack8(f);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) accm[k] += f[k];
        if (s_ > 0) reinterpret_cast<u32x4*>(dfeats)[(((long)b * (S - 1) + s_ - 1) * D + c0) >> 3] = raw;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        dpos[(long)s_ * D + c0 + k] += acc[k];              // each (s, c) is owned by exactly one thread
        part[(long)s_ * D + c0 + k] = acc[k];
        if (s_ == 0) dcls[c0 + k] += accm[k];
    }
}

// ---------------------------------------------------------------------------------------------------------
// dz = dy * gelu'(z) (optional), db[n] += sum_rows dz      — thread owns 8 columns, block covers a row slab
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bias_act_bwd(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ z,
                                                      bf16_t* __restrict__ dz, float* __restrict__ db, float* __restrict__ part, int R, int N,
                                                      int n_valid, int ld, int rows_per_block, int act, float gscale) {
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
        // two rows per trip, all four loads requested before the first is used (one row per trip left a thread with a chain of
        // dependent global round trips: 15 us for 960 x 2,048 values)
        for (int r = r0 + rl; r < r1; r += 16) {
            const int rb = r + 8 < r1 ? r + 8 : r;          // (clamped: the second row's result is dropped when it is out of range)
            const bool hb = r + 8 < r1;
            const u32x4 ga = *reinterpret_cast<const u32x4*>(dy + (long)r * ld + v * 8);
            const u32x4 gb = *reinterpret_cast<const u32x4*>(dy + (long)rb * ld + v * 8);
            float g0[8], g1[8];
            if (z != nullptr) {
                const u32x4 za = *reinterpret_cast<const u32x4*>(z + (long)r * ld + v * 8);
                const u32x4 zb = *reinterpret_cast<const u32x4*>(z + (long)rb * ld + v * 8);
                float z0[8], z1[8];
                unpack8(ga, g0); unpack8(gb, g1); unpack8(za, z0); unpack8(zb, z1);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    g0[k] = act == 2 ? (z0[k] > 0.f ? g0[k] * gscale : 0.f) : g0[k] * gelu_erf_grad(z0[k]) * gscale;
                    g1[k] = act == 2 ? (z1[k] > 0.f ? g1[k] * gscale : 0.f) : g1[k] * gelu_erf_grad(z1[k]) * gscale;
                }
                *reinterpret_cast<u32x4*>(dz + (long)r * ld + v * 8) = pack8(g0);
                if (hb) *reinterpret_cast<u32x4*>(dz + (long)rb * ld + v * 8) = pack8(g1);
            } else {
                unpack8(ga, g0); unpack8(gb, g1);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = (acc[k] + g0[k]) + (hb ? g1[k] : 0.f);
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
    if (col < N) {
        if (gridDim.y > 1) part[(long)blockIdx.y * N + col] = s;       // one partial row per row slab, added in a fixed order afterwards
        else if (col < n_valid) db[col] += s;
    }
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

static int ln_bwd_grid(int R) {
    // every workgroup ends with one partial row of 2*D floats; measured flat between 4 and 16 rows per workgroup at
    // 2,400 x 768, slower at 64 (too few workgroups) — 16 keeps the partials few
    int rpb = svsr_tune_get(SVSR_TUNE_LN_RPB);
    if (rpb < 1) rpb = 16;
    int grid = (R + rpb - 1) / rpb; if (grid > 512) grid = 512;
    return grid < 1 ? 1 : grid;
}

/* rows of [2][D] partials svsr_add_ln_bwd needs in `part` for R rows */
int svsr_add_ln_bwd_rows(int R) { return ln_bwd_grid(R); }

int svsr_colsum_rows(const float* ws, int nrows, int64_t ld, float* out0, int64_t n0, float* out1, int64_t n1, int accumulate, float scale,
                     hipStream_t stream);

/* the launch without its reduction: part [svsr_add_ln_bwd_rows(R)][2 * D] is left for the caller to add with svsr_colsum_rows(part, rows, 2 * D,
 * dgamma, D, dbeta, D, 1, 1.0f, any stream) — a parameter-gradient sum nothing in the backward chain waits for */
int svsr_add_ln_bwd_branch(const void* dy, const void* a, const void* r, const float* gamma, const float* mean, const float* rstd,
                           void* ds, int R, int D, const void* addend, float* part, void* ds2, float alpha2, const unsigned* drop_seed,
                           unsigned drop_site, float drop_p, hipStream_t stream) {
    if (D % 8 != 0 || D > 512 * LN_MAXV || part == nullptr) return SVSR_ERR_ARG;
    const int grid = ln_bwd_grid(R);
    hipLaunchKernelGGL(k_add_ln_bwd, dim3(grid), dim3(256), (size_t)8 * D * sizeof(float), stream, (const bf16_t*)dy, (const bf16_t*)a, (const bf16_t*)r, gamma,
                       mean, rstd, (bf16_t*)ds, part, R, D, (const bf16_t*)addend, (bf16_t*)ds2, alpha2, svsr_make_drop(drop_seed, drop_site, drop_p));
    return svsr_check_launch();
}

int svsr_add_ln_bwd_partials(const void* dy, const void* a, const void* r, const float* gamma, const float* mean, const float* rstd,
                             void* ds, int R, int D, const void* addend, float* part, hipStream_t stream) {
    return svsr_add_ln_bwd_branch(dy, a, r, gamma, mean, rstd, ds, R, D, addend, part, nullptr, 1.0f, nullptr, 0, 0.f, stream);
}

int svsr_add_ln_bwd(const void* dy, const void* a, const void* r, const float* gamma, const float* mean, const float* rstd,
                    void* ds, float* dgamma, float* dbeta, int R, int D, const void* addend, float* part, hipStream_t stream) {
    const int rc = svsr_add_ln_bwd_partials(dy, a, r, gamma, mean, rstd, ds, R, D, addend, part, stream);
    if (rc != SVSR_OK) return rc;
    return svsr_colsum_rows(part, ln_bwd_grid(R), 2 * D, dgamma, D, dbeta, D, 1, 1.0f, stream);
}

int svsr_embed_ln_fwd(const void* feats, const float* cls, const float* pos, const float* type0, const float* gamma,
                      const float* beta, void* sum_out, void* y, float* mean, float* rstd, int B, int S, int D, float eps,
                      const unsigned* drop_seed, unsigned site_in, float p_in, unsigned site_out, float p_out, hipStream_t stream) {
    if (D % 512 != 0 || D > 512 * LN_MAXV || S < 2) return SVSR_ERR_ARG;
    int grid = (B * S + 3) / 4; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_embed_ln_fwd, dim3(grid), dim3(256), 0, stream, (const bf16_t*)feats, cls, pos, type0, gamma, beta,
                       (bf16_t*)sum_out, (bf16_t*)y, mean, rstd, B, S, D, eps, svsr_make_drop(drop_seed, site_in, p_in),
                       svsr_make_drop(drop_seed, site_out, p_out));
    return svsr_check_launch();
}

int svsr_embed_bwd_scatter(const void* ds, void* dfeats, float* dcls, float* dpos, float* dtype0, int B, int S, int D, float* part,
                           const unsigned* drop_seed, unsigned site_in, float p_in, hipStream_t stream) {
    if (D % 8 != 0 || part == nullptr) return SVSR_ERR_ARG;       // part: [S][D] floats
    const int n = S * (D / 8);
    hipLaunchKernelGGL(k_embed_bwd_scatter, dim3((n + 255) / 256), dim3(256), 0, stream, (const bf16_t*)ds, (bf16_t*)dfeats, dcls,
                       dpos, part, B, S, D, svsr_make_drop(drop_seed, site_in, p_in));
    const int rc = svsr_check_launch();
    if (rc != SVSR_OK) return rc;
    return svsr_colsum_rows(part, S, D, dtype0, D, nullptr, 0, 1, 1.0f, stream);
}

static void bias_bwd_grid(int R, int N, int& col_blocks, int& splits, int& rpb) {
    const int cv = N / 8;
    col_blocks = (cv + 31) / 32;
    splits = (R + 15) / 16;                                  // 16 rows (2 per thread) per block: 480 workgroups at 960 x 2,048 ...
    while (splits > 1 && col_blocks * splits > 2048) splits = (splits + 1) / 2;
    rpb = (R + splits - 1) / splits;
    splits = (R + rpb - 1) / rpb;
}

/* rows of [N] partials svsr_bias_act_bwd needs in `part` (0: a single row slab adds into db directly) */
int svsr_bias_act_bwd_rows(int R, int N) {
    if (R < 1 || N < 8) return 0;
    int cb, sp, rpb;
    bias_bwd_grid(R, N, cb, sp, rpb);
    return sp > 1 ? sp : 0;
}

/* the launch without its reduction (only when svsr_bias_act_bwd_rows(R, N) > 0, i.e. the grid has several row slabs): part [rows][N] is left
 * for the caller to add with svsr_colsum_rows(part, rows, N, db, n_valid, null, 0, 1, 1.0f, any stream) */
int svsr_bias_act_bwd_partials(const void* dy, const void* z, void* dz, float* db, int R, int N, int n_valid, int ld, int act, float gscale,
                               float* part, hipStream_t stream) {
    if (N % 8 != 0 || ld % 8 != 0) return SVSR_ERR_ARG;
    int col_blocks, splits, rpb;
    bias_bwd_grid(R, N, col_blocks, splits, rpb);
    if (db != nullptr && splits > 1 && part == nullptr) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_bias_act_bwd, dim3(col_blocks, splits), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)z,
                       (bf16_t*)dz, db, part, R, N, n_valid, ld, rpb, act, gscale);
    return svsr_check_launch();
}

int svsr_bias_act_bwd(const void* dy, const void* z, void* dz, float* db, int R, int N, int n_valid, int ld, int act, float gscale,
                      float* part, hipStream_t stream) {
    const int rc = svsr_bias_act_bwd_partials(dy, z, dz, db, R, N, n_valid, ld, act, gscale, part, stream);
    const int rows = svsr_bias_act_bwd_rows(R, N);
    if (rc != SVSR_OK || db == nullptr || rows <= 0) return rc;
    return svsr_colsum_rows(part, rows, N, db, n_valid, nullptr, 0, 1, 1.0f, stream);
}

}  // extern "C"
