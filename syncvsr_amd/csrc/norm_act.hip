// HBM-bound passes of the visual front-end on NHWC bf16 tensors (gfx950):
//   train-mode BatchNorm statistics -> normalise + affine + (residual) + ReLU, and its backward;
//   the stem's BatchNorm3d + exact GELU + MaxPool3d((1,3,3),(1,2,2),(0,1,1)) fused pass and its backward;
//   the global spatial mean.
// Replaces nn.BatchNorm3d/2d, nn.GELU, nn.ReLU, nn.MaxPool3d, the residual add and hidden.mean((2,3))
// (reference LRW/video/src/lightning.py:49-54,118; tcn/models/resnet.py:59-72).
// Every thread owns 8 consecutive channels (one 16-byte vector) of a pixel; per-channel parameters stay in registers.
#include "common.h"

// ---------------------------------------------------------------------------------------------------------
// statistics finalisation.  part: [nrows][2][C] (sum, sum of squares), one row per workgroup of the producing conv's
// epilogue (plain stores).  Rows are added in a FIXED order — 32 row lanes each walking rows rl, rl+32, ... then the lane
// sums in lane order, in double — so the statistics, and with them the whole step, are reproducible run to run.
// Writes mean / rstd and updates the running statistics exactly as torch does (momentum 0.1, unbiased variance for the
// running estimate).  Block = 16 channels x 64 row lanes (1024 threads): with ~1000 partial rows every lane walks ~16 rows, four
// row pairs in flight — the launch is a few microseconds of pure latency, and there are 40 of them per step.
// ---------------------------------------------------------------------------------------------------------
#define BNF_CL 4      // 256-thread blocks: a 1,024-thread block needs a CU with 16 free wave slots, and under a side-stream weight gradient (two ~200-register workgroups per CU) no CU has them — the finaliser, 5 us of work on the critical path, then waited 25-35 us for a whole CU to drain
#define BNF_RL 64
__device__ __forceinline__ void bn_rows_sum(const float* __restrict__ part, int nrows, int C, int c, bool live, int rl, int cl,
                                            double (*sred)[2][BNF_CL], double& s, double& q) {
    double a = 0.0, b = 0.0;
    if (live) {
        int r = rl;
        for (; r + 7 * BNF_RL < nrows; r += 8 * BNF_RL) {           // eight independent row pairs in flight (one round trip for up to 512 rows), added in row order
            float av[8], bv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                av[k] = part[((long)(r + k * BNF_RL) * 2 + 0) * C + c];
                bv[k] = part[((long)(r + k * BNF_RL) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { a += (double)av[k]; b += (double)bv[k]; }
        }
        for (; r + 3 * BNF_RL < nrows; r += 4 * BNF_RL) {           // four
            const float a0 = part[((long)r * 2 + 0) * C + c], b0 = part[((long)r * 2 + 1) * C + c];
            const float a1 = part[((long)(r + BNF_RL) * 2 + 0) * C + c], b1 = part[((long)(r + BNF_RL) * 2 + 1) * C + c];
            const float a2 = part[((long)(r + 2 * BNF_RL) * 2 + 0) * C + c], b2 = part[((long)(r + 2 * BNF_RL) * 2 + 1) * C + c];
            const float a3 = part[((long)(r + 3 * BNF_RL) * 2 + 0) * C + c], b3 = part[((long)(r + 3 * BNF_RL) * 2 + 1) * C + c];
            a = (((a + (double)a0) + (double)a1) + (double)a2) + (double)a3;
            b = (((b + (double)b0) + (double)b1) + (double)b2) + (double)b3;
        }
        for (; r < nrows; r += BNF_RL) { a += (double)part[((long)r * 2 + 0) * C + c]; b += (double)part[((long)r * 2 + 1) * C + c]; }
    }
    sred[rl][0][cl] = a;
    sred[rl][1][cl] = b;
    __syncthreads();
    s = 0.0; q = 0.0;
    if (rl == 0) {
#pragma unroll 8
        for (int k = 0; k < BNF_RL; ++k) { s += sred[k][0][cl]; q += sred[k][1][cl]; }
    }
}

__global__ __launch_bounds__(256) void k_bn_finalize(const float* __restrict__ part, int nrows, int C, float count, float eps, float momentum,
                                                      float* mean, float* rstd, float* running_mean, float* running_var,
                                                      long* num_batches_tracked) {
    __shared__ double sred[BNF_RL][2][BNF_CL];
    const int cl = threadIdx.x % BNF_CL, rl = threadIdx.x / BNF_CL;
    const int c = blockIdx.x * BNF_CL + cl;
    double s, q;
    bn_rows_sum(part, nrows, C, c, c < C, rl, cl, sred, s, q);
    if (rl == 0 && c < C) {
        const double m = s / (double)count;
        double var = q / (double)count - m * m;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean != nullptr) {
            const double unbiased = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
    if (num_batches_tracked != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
}

// eval mode: mean = running_mean, rstd = 1/sqrt(running_var + eps)
__global__ void k_bn_eval_prepare(const float* running_mean, const float* running_var, int C, float eps, float* mean, float* rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { mean[c] = running_mean[c]; rstd[c] = rsqrtf(running_var[c] + eps); }
}

// ---------------------------------------------------------------------------------------------------------
// y = act(gamma * (x - mean) * rstd + beta [+ res])         act: 0 none, 1 relu, 2 swish
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bn_act_fwd(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                    bf16_t* __restrict__ y, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, long nvec, int C, int act) {
    const int cv = C >> 3;
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    if (idx >= nvec) return;
    const int c0 = (int)(idx % cv) * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = gamma[c0 + k] * rstd[c0 + k];
        sh[k] = __builtin_fmaf(-mean[c0 + k], sc[k], beta[c0 + k]);
    }
    for (; idx < nvec; idx += stride) {
        float f[8];
        unpack8(reinterpret_cast<const u32x4*>(x)[idx], f);
        // (explicit fused multiply-adds: the data-gradient epilogues that recompute the ReLU mask as bn(x) > 0 use the same two)
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = __builtin_fmaf(f[k], sc[k], sh[k]);
        if (res != nullptr) {
            float r[8];
            unpack8(reinterpret_cast<const u32x4*>(res)[idx], r);
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] += r[k];
        }
        if (act == 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = fmaxf(f[k], 0.f);
        } else if (act == 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = swish(f[k]);
        }
        reinterpret_cast<u32x4*>(y)[idx] = pack8(f);
    }
}

// block-level reduction of per-thread 8-channel partials into this workgroup's row of the partials [rows][2][C]
// (fixed order inside the block; k_bn_bwd_finalize adds the rows in a fixed order)
__device__ __forceinline__ void reduce_to_row(float* sred, const float* s1, const float* s2, int cv, int C, float* part,
                                              int first_group = 0) {
    // sred: [256][16]
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) { sred[tid * 16 + k] = s1[k]; sred[tid * 16 + 8 + k] = s2[k]; }
    __syncthreads();
    // thread t owns channel group (first_group + t) % cv; first_group = 0 whenever 256 % cv == 0
    const long row = (long)blockIdx.y * gridDim.x + blockIdx.x;
    for (int o = tid; o < cv * 16; o += 256) {
        const int g = o >> 4, k = o & 15;
        float acc = 0.f;
        int t0 = g - first_group;
        if (t0 < 0) t0 += cv;
        for (int t = t0; t < 256; t += cv) acc += sred[t * 16 + k];
        const int which = k >> 3, c = g * 8 + (k & 7);
        part[(row * 2 + which) * C + c] = acc;
    }
}

// pass 1 of the backward: slots += (sum g, sum g * xhat) with g = dy * act'(.)   (relu mask from saved y)
// (templated on the activation: the Swish variant's extra registers must not lower the occupancy of the ReLU passes,
// which run at HBM speed in the LRW trunk)
template <int act>
__global__ __launch_bounds__(256) void k_bn_act_bwd_reduce(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y,
                                                           const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, long nvec, int C,
                                                           float* slots, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const bf16_t* __restrict__ res) {
    __shared__ float sred[256 * 16];
    const int cv = C >> 3;
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;       // a multiple of cv (host): a thread keeps its channel group
    const int c0 = (int)(idx % cv) * 8;
    const int first_group = (int)(((long)blockIdx.x * 256) % cv);
    float mu[8], rs[8], ga[8], be[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu[k] = mean[c0 + k]; rs[k] = rstd[c0 + k]; s1[k] = 0.f; s2[k] = 0.f;
        if (act == 2) { ga[k] = gamma[c0 + k]; be[k] = beta[c0 + k]; }
    }
    for (; idx < nvec; idx += stride) {
        float g[8], xv[8];
        unpack8(reinterpret_cast<const u32x4*>(dy)[idx], g);
        unpack8(reinterpret_cast<const u32x4*>(x)[idx], xv);
        if (act == 1) {
            float yv[8];
            unpack8(reinterpret_cast<const u32x4*>(y)[idx], yv);
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = yv[k] > 0.f ? g[k] : 0.f;
        } else if (act == 2) {     // Swish: the pre-activation is recomputed from x (and the residual branch)
            float rv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) rv[k] = 0.f;
            if (res != nullptr) unpack8(reinterpret_cast<const u32x4*>(res)[idx], rv);
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] *= swish_grad((xv[k] - mu[k]) * rs[k] * ga[k] + be[k] + rv[k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1[k] += g[k]; s2[k] += g[k] * (xv[k] - mu[k]) * rs[k]; }
    }
    reduce_to_row(sred, s1, s2, cv, C, slots, first_group);
}

// finalise the backward statistics (rows added in the same fixed order as k_bn_finalize):
// dbeta += sum g, dgamma += sum g*xhat, coef = {gamma*rstd, sum g / n, sum g*xhat / n}
__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const float* __restrict__ part, int nrows, int C, float count, const float* gamma,
                                                          const float* rstd, float* dgamma, float* dbeta, float* coef) {
    __shared__ double sred[BNF_RL][2][BNF_CL];
    const int cl = threadIdx.x % BNF_CL, rl = threadIdx.x / BNF_CL;
    const int c = blockIdx.x * BNF_CL + cl;
    double s, q;
    bn_rows_sum(part, nrows, C, c, c < C, rl, cl, sred, s, q);
    if (rl == 0 && c < C) {
        dbeta[c] += (float)s;
        dgamma[c] += (float)q;
        coef[c] = gamma[c] * rstd[c];
        coef[C + c] = (float)(s / (double)count);
        coef[2 * C + c] = (float)(q / (double)count);
    }
}

// pass 2: dx = gamma*rstd * (g - mean(g) - xhat * mean(g*xhat));  dres = g (optional)
template <int act>
__global__ __launch_bounds__(256) void k_bn_act_bwd_apply(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y,
                                                          const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ coef,
                                                          bf16_t* __restrict__ dx, bf16_t* __restrict__ dres, long nvec,
                                                          int C, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const bf16_t* __restrict__ res) {
    const int cv = C >> 3;
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    if (idx >= nvec) return;
    const int c0 = (int)(idx % cv) * 8;
    float mu[8], rs[8], k0[8], k1[8], k2[8], ga[8], be[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu[k] = mean[c0 + k]; rs[k] = rstd[c0 + k];
        k0[k] = coef[c0 + k]; k1[k] = coef[C + c0 + k]; k2[k] = coef[2 * C + c0 + k];
        if (act == 2) { ga[k] = gamma[c0 + k]; be[k] = beta[c0 + k]; }
    }
    for (; idx < nvec; idx += stride) {
        float g[8], xv[8], o[8];
        unpack8(reinterpret_cast<const u32x4*>(dy)[idx], g);
        unpack8(reinterpret_cast<const u32x4*>(x)[idx], xv);
        if (act == 1) {
            float yv[8];
            unpack8(reinterpret_cast<const u32x4*>(y)[idx], yv);
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = yv[k] > 0.f ? g[k] : 0.f;
        } else if (act == 2) {
            float rv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) rv[k] = 0.f;
            if (res != nullptr) unpack8(reinterpret_cast<const u32x4*>(res)[idx], rv);
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] *= swish_grad((xv[k] - mu[k]) * rs[k] * ga[k] + be[k] + rv[k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = k0[k] * (g[k] - k1[k] - (xv[k] - mu[k]) * rs[k] * k2[k]);
        reinterpret_cast<u32x4*>(dx)[idx] = pack8(o);
        if (dres != nullptr) reinterpret_cast<u32x4*>(dres)[idx] = pack8(g);
    }
}

// ---------------------------------------------------------------------------------------------------------
// stem: y[n,ph,pw,:] = max_{3x3 window, stride 2, pad 1 (-inf)} gelu(bn(x[n,2ph-1+i,2pw-1+j,:])); argmax index i*3+j
// first maximum in row-major window order wins (torch CPU max_pool semantics: strictly-greater update).
// ---------------------------------------------------------------------------------------------------------
// Index decode shared by the stem passes: a block owns `rpb` consecutive rows of one frame (blockIdx.y = frame) and walks
// its (row, column, 8-channel group) items with 32-bit arithmetic and a float-reciprocal division — the first version's
// 64-bit divisions by run-time values cost more than the GELU.
struct StemRowIter {
    int W, cv, cshift, rpb;
    float inv_wcv;
    __device__ __forceinline__ void decode(int e, int& rl, int& w, int& c8) const {
        const int wcv = W << cshift;
        rl = (int)((float)e * inv_wcv);
        int rem = e - rl * wcv;
        if (rem < 0) { rl--; rem += wcv; } else if (rem >= wcv) { rl++; rem -= wcv; }
        w = rem >> cshift;
        c8 = rem & (cv - 1);
    }
};

template <int ACT>
__global__ __launch_bounds__(256) void k_stem_bn_act_pool_fwd(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                               unsigned char* __restrict__ amax, bf16_t* __restrict__ xwin,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               int N, int Hc, int Wc, int Hp, int Wp, int C, StemRowIter it) {
    const int n = blockIdx.y, row0 = blockIdx.x * it.rpb;
    const int c0 = (threadIdx.x & (it.cv - 1)) * 8;            // 256 % cv == 0: a thread keeps its channel group
    float sc[8], sh[8], rsc[8];
    bool any_zero_scale = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = gamma[c0 + k] * rstd[c0 + k]; sh[k] = beta[c0 + k] - mean[c0 + k] * sc[k];
        rsc[k] = sc[k] != 0.f ? 1.0f / sc[k] : 0.f;
        any_zero_scale |= sc[k] == 0.f;
    }
    int rows = Hp - row0; if (rows > it.rpb) rows = it.rpb;
    const int items = rows * (Wp << it.cshift);
    for (int e = threadIdx.x; e < items; e += 256) {
        int rl, pw, c8;
        it.decode(e, rl, pw, c8);
        const int ph = row0 + rl;
        // GELU and Swish both fall to a single minimum and rise from there (quasi-convex), so the maximum of act(z) over a window is
        // attained at the window's largest or smallest z: track those two (first occurrence each) and evaluate the activation twice
        // per channel instead of nine times — the pass was bound by the erf/exp VALU work, not by HBM.
        // (A fast path for windows whose largest z is >= 0 — the maximum is then there, one activation per channel — measured SLOWER,
        // 192 vs 150 us: with 8 channels x 64 lanes some lane of nearly every wave has an all-negative window, so both paths run.)
        float zhi[8], zlo[8], zw[8];
        int ihi[8], ilo[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { zhi[k] = -INFINITY; zlo[k] = INFINITY; ihi[k] = 0; ilo[k] = 0; }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int h = 2 * ph - 1 + i;
            if (h < 0 || h >= Hc) continue;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int w = 2 * pw - 1 + j;
                if (w < 0 || w >= Wc) continue;
                float f[8];
                unpack8(*reinterpret_cast<const u32x4*>(x + (((long)n * Hc + h) * Wc + w) * C + c0), f);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float z = f[k] * sc[k] + sh[k];
                    if (z > zhi[k]) { zhi[k] = z; ihi[k] = i * 3 + j; }
                    if (z < zlo[k]) { zlo[k] = z; ilo[k] = i * 3 + j; }
                }
            }
        }
        float best[8];
        int bi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float vh = ACT == 2 ? swish(zhi[k]) : gelu_erf(zhi[k]);
            const float vl = ACT == 2 ? swish(zlo[k]) : gelu_erf(zlo[k]);
            // the first maximum in window order wins (torch's max_pool: strictly-greater update)
            const bool take_lo = vl > vh || (vl == vh && ilo[k] < ihi[k]);
            best[k] = take_lo ? vl : vh;
            bi[k] = take_lo ? ilo[k] : ihi[k];
            zw[k] = take_lo ? zlo[k] : zhi[k];
        }
        const long o = (((long)n * Hp + ph) * Wp + pw) * C + c0;
        *reinterpret_cast<u32x4*>(y + o) = pack8(best);
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { lo |= (unsigned)bi[k] << (8 * k); hi |= (unsigned)bi[k + 4] << (8 * k); }
        *reinterpret_cast<uint2*>(amax + o) = make_uint2(lo, hi);
        if (xwin != nullptr) {
            // the winner's convolution output, for the backward's reduce pass (svsr_stem_bn_act_pool_bwd with xwin): recovered from its
            // z = x * sc + sh — x is a bf16 value, the few fp32 ulps of the round trip vanish in the rounding back to bf16 — or read
            // again where a channel's scale is exactly zero
            float xw[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) xw[k] = (zw[k] - sh[k]) * rsc[k];
            if (any_zero_scale) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (sc[k] == 0.f) {
                        const int i = (bi[k] * 11) >> 5, j = bi[k] - 3 * i;
                        xw[k] = bf2f(x[(((long)n * Hc + (2 * ph - 1 + i)) * Wc + (2 * pw - 1 + j)) * C + c0 + k]);
                    }
            }
            *reinterpret_cast<u32x4*>(xwin + o) = pack8(xw);
        }
    }
}

// gradient reaching the stem conv output element (n,h,w,c..c+7) through pool -> gelu:  g = (sum of dpool over the
// windows whose argmax is this element) * gelu'(bn(x)).
template <int ACT>
__device__ __forceinline__ void stem_gather_g(const bf16_t* __restrict__ dpool, const unsigned char* __restrict__ amax,
                                              int n, int h, int w, int c0, int Hp, int Wp, int C, const float* z, float* g) {
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    const int ph_lo = h >> 1, ph_hi = (h & 1) ? (h >> 1) + 1 : (h >> 1);
    const int pw_lo = w >> 1, pw_hi = (w & 1) ? (w >> 1) + 1 : (w >> 1);
    for (int ph = ph_lo; ph <= ph_hi; ++ph) {
        if (ph >= Hp) continue;
        const int i = h - (2 * ph - 1);
        for (int pw = pw_lo; pw <= pw_hi; ++pw) {
            if (pw >= Wp) continue;
            const int j = w - (2 * pw - 1);
            const long o = (((long)n * Hp + ph) * Wp + pw) * C + c0;
            const uint2 am = *reinterpret_cast<const uint2*>(amax + o);
            float d[8];
            unpack8(*reinterpret_cast<const u32x4*>(dpool + o), d);
            const unsigned want = (unsigned)(i * 3 + j);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (((am.x >> (8 * k)) & 0xffu) == want) acc[k] += d[k];
                if (((am.y >> (8 * k)) & 0xffu) == want) acc[k + 4] += d[k + 4];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = acc[k] * (ACT == 2 ? swish_grad(z[k]) : gelu_erf_grad(z[k]));
}

template <int ACT>
__global__ __launch_bounds__(256) void k_stem_pool_bwd_reduce(const bf16_t* __restrict__ dpool, const unsigned char* __restrict__ amax,
                                                              const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int N, int Hc, int Wc, int Hp, int Wp,
                                                              int C, float* slots, StemRowIter it) {
    __shared__ float sred[256 * 16];
    const int n = blockIdx.y, row0 = blockIdx.x * it.rpb;
    const int c0 = (threadIdx.x & (it.cv - 1)) * 8;
    float mu[8], rs[8], ga[8], be[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { mu[k] = mean[c0 + k]; rs[k] = rstd[c0 + k]; ga[k] = gamma[c0 + k]; be[k] = beta[c0 + k]; s1[k] = 0.f; s2[k] = 0.f; }
    int rows = Hc - row0; if (rows > it.rpb) rows = it.rpb;
    const int items = rows * (Wc << it.cshift);
    for (int e = threadIdx.x; e < items; e += 256) {
        int rl, w, c8;
        it.decode(e, rl, w, c8);
        const int h = row0 + rl;
        float xv[8], xh[8], z[8], g[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + (((long)n * Hc + h) * Wc + w) * C + c0), xv);
#pragma unroll
        for (int k = 0; k < 8; ++k) { xh[k] = (xv[k] - mu[k]) * rs[k]; z[k] = ga[k] * xh[k] + be[k]; }
        stem_gather_g<ACT>(dpool, amax, n, h, w, c0, Hp, Wp, C, z, g);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1[k] += g[k]; s2[k] += g[k] * xh[k]; }
    }
    reduce_to_row(sred, s1, s2, it.cv, C, slots);
}

template <int ACT>
__global__ __launch_bounds__(256) void k_stem_pool_bwd_apply(const bf16_t* __restrict__ dpool, const unsigned char* __restrict__ amax,
                                                             const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ coef,
                                                             bf16_t* __restrict__ dx, int N, int Hc, int Wc, int Hp, int Wp, int C,
                                                             StemRowIter it) {
    const int n = blockIdx.y, row0 = blockIdx.x * it.rpb;
    const int c0 = (threadIdx.x & (it.cv - 1)) * 8;
    float mu[8], rs[8], ga[8], be[8], k0[8], k1[8], k2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu[k] = mean[c0 + k]; rs[k] = rstd[c0 + k]; ga[k] = gamma[c0 + k]; be[k] = beta[c0 + k];
        k0[k] = coef[c0 + k]; k1[k] = coef[C + c0 + k]; k2[k] = coef[2 * C + c0 + k];
    }
    int rows = Hc - row0; if (rows > it.rpb) rows = it.rpb;
    const int items = rows * (Wc << it.cshift);
    for (int e = threadIdx.x; e < items; e += 256) {
        int rl, w, c8;
        it.decode(e, rl, w, c8);
        const int h = row0 + rl;
        const long o = (((long)n * Hc + h) * Wc + w) * C + c0;
        float xv[8], xh[8], z[8], g[8], ov[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + o), xv);
#pragma unroll
        for (int k = 0; k < 8; ++k) { xh[k] = (xv[k] - mu[k]) * rs[k]; z[k] = ga[k] * xh[k] + be[k]; }
        stem_gather_g<ACT>(dpool, amax, n, h, w, c0, Hp, Wp, C, z, g);
#pragma unroll
        for (int k = 0; k < 8; ++k) ov[k] = k0[k] * (g[k] - k1[k] - xh[k] * k2[k]);
        *reinterpret_cast<u32x4*>(dx + o) = pack8(ov);
    }
}


// ---------------------------------------------------------------------------------------------------------
// LDS-tiled variants of the stem passes (backward: default; forward: opt-in, see stem_lds_fwd()).  The first versions above evaluate the activation of every conv
// output once per pooling window that contains it (2.25x) and gather dpool/amax for every element from L2 (six times
// the bytes of the element itself); here a workgroup stages what it needs once:
//   forward : act(bn(x)) of the 2*PR+1 input rows behind PR pooled rows, fp32 (exact argmax), then 3x3 max from LDS
//   backward: dpool + amax of the R/2+1 pooled rows above R input rows, then the routing gather from LDS
// ---------------------------------------------------------------------------------------------------------
#define STEM_PR 2
#define STEM_BR 8

template <int ACT>
__global__ __launch_bounds__(256) void k_stem_fwd_lds(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, unsigned char* __restrict__ amax,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      int Hc, int Wc, int Hp, int Wp, int C) {
    extern __shared__ __attribute__((aligned(16))) float sAct[];      // [2*PR+1][Wc][C]
    const int cv = C >> 3;
    const int n = blockIdx.y, ph0 = blockIdx.x * STEM_PR, h_lo = 2 * ph0 - 1;
    const int c0 = (threadIdx.x % cv) * 8;                           // 256 % cv == 0: a thread keeps its channel group
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = gamma[c0 + k] * rstd[c0 + k]; sh[k] = beta[c0 + k] - mean[c0 + k] * sc[k]; }
    const int wcv = Wc * cv;
    const int items = (2 * STEM_PR + 1) * wcv;
    for (int e = threadIdx.x; e < items; e += 256) {
        const int r = e / wcv, rem = e - r * wcv, w = rem / cv, h = h_lo + r;
        float f[8];
        if (h >= 0 && h < Hc) {
            unpack8(*reinterpret_cast<const u32x4*>(x + (((long)n * Hc + h) * Wc + w) * C + c0), f);
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float z = f[k] * sc[k] + sh[k]; f[k] = ACT == 2 ? swish(z) : gelu_erf(z); }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = -INFINITY;
        }
        float* d = sAct + (long)(r * Wc + w) * C + c0;
        *reinterpret_cast<f32x4*>(d) = f32x4{f[0], f[1], f[2], f[3]};
        *reinterpret_cast<f32x4*>(d + 4) = f32x4{f[4], f[5], f[6], f[7]};
    }
    __syncthreads();
    const int pcv = Wp * cv;
    for (int e = threadIdx.x; e < STEM_PR * pcv; e += 256) {
        const int pl = e / pcv, rem = e - pl * pcv, pw = rem / cv, ph = ph0 + pl;
        if (ph >= Hp) break;
        float best[8];
        int bi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; bi[k] = 0; }
        bool first = true;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int h = 2 * ph - 1 + i;
            if (h < 0 || h >= Hc) continue;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int w = 2 * pw - 1 + j;
                if (w < 0 || w >= Wc) continue;
                const float* sp = sAct + (long)((2 * pl + i) * Wc + w) * C + c0;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(sp), hi = *reinterpret_cast<const f32x4*>(sp + 4);
                const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (first || v[k] > best[k]) { best[k] = v[k]; bi[k] = i * 3 + j; }
                first = false;
            }
        }
        const long o = (((long)n * Hp + ph) * Wp + pw) * C + c0;
        *reinterpret_cast<u32x4*>(y + o) = pack8(best);
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { lo |= (unsigned)bi[k] << (8 * k); hi |= (unsigned)bi[k + 4] << (8 * k); }
        *reinterpret_cast<uint2*>(amax + o) = make_uint2(lo, hi);
    }
}

// GP: `dpool` already is g = dpool * act'(z of the window's winner) (written by k_stem_bwd_reduce_win): no activation derivative here —
// evaluated per convolution element it was 4 of the pass's ~6 G vector operations
template <int ACT, bool APPLY, bool GP = false>
__global__ __launch_bounds__(256) void k_stem_bwd_lds(const bf16_t* __restrict__ dpool, const unsigned char* __restrict__ amax,
                                                      const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ coef,
                                                      bf16_t* __restrict__ dx, float* __restrict__ slots, int Hc, int Wc, int Hp, int Wp, int C) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_stem[];
    constexpr int PROWS = STEM_BR / 2 + 1;
    __shared__ float sred[APPLY ? 1 : 256 * 16];
    const int cv = C >> 3;
    bf16_t* sD = reinterpret_cast<bf16_t*>(smem_stem);                               // [PROWS][Wp][C]
    unsigned char* sM = smem_stem + (size_t)PROWS * Wp * C * sizeof(bf16_t);         // [PROWS][Wp][C]
    const int n = blockIdx.y, h0 = blockIdx.x * STEM_BR, p_lo = h0 >> 1;
    const int c0 = (threadIdx.x % cv) * 8;
    const int pcv = Wp * cv;
    for (int e = threadIdx.x; e < PROWS * pcv; e += 256) {
        const int pr = e / pcv, rem = e - pr * pcv, pw = rem / cv, ph = p_lo + pr;
        // (compiler vector types and no branch: a conditionally assigned u32x4 STRUCT is kept in scratch memory)
        typedef __attribute__((ext_vector_type(4))) unsigned v4u;
        typedef __attribute__((ext_vector_type(2))) unsigned v2u;
        const bool ok = ph < Hp;
        const long o = ok ? (((long)n * Hp + ph) * Wp + pw) * C + c0 : 0;
        v4u d = *reinterpret_cast<const v4u*>(dpool + o);
        v2u m = *reinterpret_cast<const v2u*>(amax + o);
        if (!ok) { d = v4u{0u, 0u, 0u, 0u}; m = v2u{0xffffffffu, 0xffffffffu}; }
        const int so = (pr * Wp + pw) * C + c0;
        *reinterpret_cast<v4u*>(sD + so) = d;
        *reinterpret_cast<v2u*>(sM + so) = m;
    }
    float mu[8], rs[8], ga[8], be[8], k0[8], k1[8], k2[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu[k] = mean[c0 + k]; rs[k] = rstd[c0 + k]; ga[k] = gamma[c0 + k]; be[k] = beta[c0 + k]; s1[k] = 0.f; s2[k] = 0.f;
        if (APPLY) { k0[k] = coef[c0 + k]; k1[k] = coef[C + c0 + k]; k2[k] = coef[2 * C + c0 + k]; }
    }
    __syncthreads();
    int rows = Hc - h0; if (rows > STEM_BR) rows = STEM_BR;
    const int wcv = Wc * cv;
    for (int e = threadIdx.x; e < rows * wcv; e += 256) {
        const int rl = e / wcv, rem = e - rl * wcv, w = rem / cv, h = h0 + rl;
        const long o = (((long)n * Hc + h) * Wc + w) * C + c0;
        float xv[8], xh[8], acc[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + o), xv);
#pragma unroll
        for (int k = 0; k < 8; ++k) { xh[k] = (xv[k] - mu[k]) * rs[k]; acc[k] = 0.f; }
        const int ph_lo = h >> 1, ph_hi = (h & 1) ? (h >> 1) + 1 : (h >> 1);
        const int pw_lo = w >> 1, pw_hi = (w & 1) ? (w >> 1) + 1 : (w >> 1);
        for (int ph = ph_lo; ph <= ph_hi; ++ph) {
            if (ph >= Hp) continue;
            const int i = h - (2 * ph - 1);
            for (int pw = pw_lo; pw <= pw_hi; ++pw) {
                if (pw >= Wp) continue;
                const int j = w - (2 * pw - 1);
                const int so = ((ph - p_lo) * Wp + pw) * C + c0;
                const uint2 am = *reinterpret_cast<const uint2*>(sM + so);
                float d[8];
                unpack8(*reinterpret_cast<const u32x4*>(sD + so), d);
                const unsigned want = (unsigned)(i * 3 + j);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (((am.x >> (8 * k)) & 0xffu) == want) acc[k] += d[k];
                    if (((am.y >> (8 * k)) & 0xffu) == want) acc[k + 4] += d[k + 4];
                }
            }
        }
        float g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (GP) g[k] = acc[k];
            else {
                const float z = ga[k] * xh[k] + be[k];
                g[k] = acc[k] * (ACT == 2 ? swish_grad(z) : gelu_erf_grad(z));
            }
        }
        if (APPLY) {
            float ov[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) ov[k] = k0[k] * (g[k] - k1[k] - xh[k] * k2[k]);
            *reinterpret_cast<u32x4*>(dx + o) = pack8(ov);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) { s1[k] += g[k]; s2[k] += g[k] * xh[k]; }
        }
    }
    if (!APPLY) reduce_to_row(sred, s1, s2, cv, C, slots);
}


// Reduce pass of the stem backward in GATHER form (C = 64).  sum g and sum g*xhat only need g at the window WINNERS: per pooled output
// and channel, g = dpool * act'(bn(x[winner])), four times fewer activation-derivative evaluations than the element-centric pass
// above (which evaluates it for every conv element although ~3/4 of them win no window) and no window-membership compares — that
// pass was VALU-bound at 1.7 TB/s.  A workgroup owns STEM_GP pooled rows of one frame: the 2*STEM_GP+1 conv rows under them are
// copied to LDS by DMA (16-byte pieces XOR-swizzled by pixel pair so that the per-channel 2-byte reads of lanes with equal channel
// group spread over the banks), then every thread walks pooled 8-channel vectors and reads its eight winners.
#define STEM_GP 4
__device__ unsigned g_stem_zero_page[64];
// Reduce pass of the stem's BatchNorm backward from the winners the FORWARD kept (xwin = convolution output at every window's
// arg-max): per pooled output one activation derivative, g = dpool * act'(z), written out (bf16) for the apply pass, and the sums of g
// and g * xhat over the values the apply pass will read back.  Streams 3 x 57 MB at B = 32 instead of re-reading the 230 MB
// convolution output through LDS tiles (k_stem_bwd_reduce_gather).  Same grid and partial rows as the gather form.
#define STEM_GPW 24      // pooled rows per workgroup of k_stem_bwd_reduce_win: a whole 22-row frame.  With STEM_GP (4) rows the pass wrote 5,568 (LRW) /
                         // 15,360 (LRS) partial rows and the finaliser behind it — 16 workgroups on the step's critical path — took 26 / ~70 us to add them
template <int ACT>
__global__ __launch_bounds__(256) void k_stem_bwd_reduce_win(const bf16_t* __restrict__ dpool, const bf16_t* __restrict__ xwin,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              bf16_t* __restrict__ gpool, float* __restrict__ slots, int Hp, int Wp) {
    constexpr int C = 64;
    __shared__ float sred[256 * 16];
    const int tid = threadIdx.x, n = blockIdx.y, p0 = blockIdx.x * STEM_GPW;
    const int c0 = (tid & 7) * 8;
    float mu[8], rs[8], ga[8], be[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu[k] = mean[c0 + k]; rs[k] = rstd[c0 + k]; ga[k] = gamma[c0 + k]; be[k] = beta[c0 + k]; s1[k] = 0.f; s2[k] = 0.f;
    }
    int rows = Hp - p0; if (rows > STEM_GPW) rows = STEM_GPW;
    const int pcv = Wp * 8;
    for (int v = tid; v < rows * pcv; v += 256) {               // 256 % 8 == 0: a thread keeps its channel group
        const int pr = v / pcv, pw = (v - pr * pcv) >> 3;
        const long o = (((long)n * Hp + p0 + pr) * Wp + pw) * C + c0;
        float d[8], xv[8], g[8];
        unpack8(*reinterpret_cast<const u32x4*>(dpool + o), d);
        unpack8(*reinterpret_cast<const u32x4*>(xwin + o), xv);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float xh = (xv[k] - mu[k]) * rs[k];
            const float z = ga[k] * xh + be[k];
            g[k] = bf2f(f2bf(d[k] * (ACT == 2 ? swish_grad(z) : gelu_erf_grad(z))));
            s1[k] += g[k];
            s2[k] += g[k] * xh;
        }
        *reinterpret_cast<u32x4*>(gpool + o) = pack8(g);
    }
    reduce_to_row(sred, s1, s2, 8, C, slots);
}

template <int ACT>
__global__ __launch_bounds__(256) void k_stem_bwd_reduce_gather(const bf16_t* __restrict__ dpool, const unsigned char* __restrict__ amax,
                                                                const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ slots, int Hc, int Wc,
                                                                int Hp, int Wp) {
    constexpr int C = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_stem[];
    bf16_t* sX = reinterpret_cast<bf16_t*>(smem_stem);            // [2*STEM_GP+1][Wc] pixels x 8 pieces of 8 channels, piece index swizzled
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.y, p0 = blockIdx.x * STEM_GP, h_lo = 2 * p0 - 1;
    const int npix = (2 * STEM_GP + 1) * Wc, total = npix * 8;
    for (int e0 = 0; e0 < total; e0 += 256) {
        const int e = e0 + tid, q = e >> 3, slot = e & 7;
        const int r = q / Wc, h = h_lo + r;
        const bf16_t* src = (e < total && h >= 0 && h < Hc)
                                ? x + (((long)n * Hc + h) * Wc + (q - r * Wc)) * C + ((slot ^ ((q >> 1) & 7)) << 3)
                                : reinterpret_cast<const bf16_t*>(g_stem_zero_page) + slot * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sX + (size_t)(e0 + wave * 64) * 8), 16, 0, 0);
    }
    const int c8 = tid & 7, c0 = c8 * 8;
    float mu[8], rs[8], ga[8], be[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu[k] = mean[c0 + k]; rs[k] = rstd[c0 + k]; ga[k] = gamma[c0 + k]; be[k] = beta[c0 + k]; s1[k] = 0.f; s2[k] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int pcv = Wp * 8;
    for (int v = tid; v < STEM_GP * pcv; v += 256) {           // 256 % 8 == 0: a thread keeps its channel group
        const int pr = v / pcv, pw = (v - pr * pcv) >> 3, ph = p0 + pr;
        if (ph >= Hp) break;
        const long o = (((long)n * Hp + ph) * Wp + pw) * C + c0;
        float d[8];
        unpack8(*reinterpret_cast<const u32x4*>(dpool + o), d);
        const uint2 am = *reinterpret_cast<const uint2*>(amax + o);
        const int qbase = (2 * pr) * Wc + 2 * pw - 1;          // tile-local pixel of window position (0,0): row 2*ph-1 - h_lo = 2*pr
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = (int)(((k < 4 ? am.x : am.y) >> (8 * (k & 3))) & 0xffu);
            const int i = (idx * 11) >> 5, j = idx - 3 * i;
            const int q = qbase + i * Wc + j;
            const float xv = bf2f(sX[q * C + (((c8 ^ ((q >> 1) & 7))) << 3) + k]);
            const float xh = (xv - mu[k]) * rs[k];
            const float z = ga[k] * xh + be[k];
            const float g = d[k] * (ACT == 2 ? swish_grad(z) : gelu_erf_grad(z));
            s1[k] += g;
            s2[k] += g * xh;
        }
    }
    __syncthreads();                                              // the tile is dead: its first 16 KiB become the reduction buffer
    reduce_to_row(reinterpret_cast<float*>(smem_stem), s1, s2, 8, C, slots);
}

// ---------------------------------------------------------------------------------------------------------
// global spatial mean  [N][HW][C] -> [N][C]  and its backward
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_avgpool_fwd(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long N, int HW, int C) {
    const int cv = C >> 3;
    const long nvec = N * cv;
    const float inv = 1.f / (float)HW;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < nvec; idx += (long)gridDim.x * 256) {
        const long n = idx / cv;
        const int c0 = (int)(idx % cv) * 8;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int p = 0; p < HW; ++p) {
            float f[8];
            unpack8(*reinterpret_cast<const u32x4*>(x + ((long)n * HW + p) * C + c0), f);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += f[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] *= inv;
        reinterpret_cast<u32x4*>(y)[idx] = pack8(acc);
    }
}

__global__ __launch_bounds__(256) void k_avgpool_bwd(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, long N, int HW, int C) {
    const int cv = C >> 3;
    const long nvec = N * HW * cv;
    const float inv = 1.f / (float)HW;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < nvec; idx += (long)gridDim.x * 256) {
        const long n = idx / ((long)HW * cv);
        const int c8 = (int)(idx % cv);
        float f[8];
        unpack8(reinterpret_cast<const u32x4*>(dy)[n * cv + c8], f);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] *= inv;
        reinterpret_cast<u32x4*>(dx)[idx] = pack8(f);
    }
}

// ---------------------------------------------------------------------------------------------------------
static inline int ew_grid(long nvec) {
    long b = (nvec + 255) / 256;
    if (b > 2048) b = 2048;     // 8 blocks x 256 CUs, grid-stride beyond that
    if (b < 1) b = 1;
    return (int)b;
}
// measured at 928 x 44 x 44 x 64: forward 191 us (direct) vs 225 us (LDS-tiled) — the pass is bound by the erf/exp VALU work,
// not by the redundant window reads; backward 411 us (direct) vs 360 us (LDS-tiled).  Defaults follow the measurement.
static inline bool stem_lds_fwd() { return svsr_tune_get(SVSR_TUNE_STEM_LDS_FWD) != 0; }
static inline bool stem_lds_bwd() { return svsr_tune_get(SVSR_TUNE_STEM_LDS_BWD) != 0; }
static inline bool chan_ok(int C) { return C >= 8 && C <= 2048 && (2048 % C) == 0; }
// any C % 8 == 0 up to 2048 (e.g. 768): the grid is rounded so that gridDim.x * 256 is a multiple of C/8 and every
// thread keeps one channel group across its grid-stride loop
static inline bool chan_ok_any(int C) { return C >= 8 && C <= 2048 && (C % 8) == 0; }
static inline int ew_grid_for(long nvec, int C, int cap = 2048) {
    const int cv = C / 8;
    int a = cv, b = 256;
    while (b) { const int t = a % b; a = b; b = t; }
    const int m = cv / a;                 // grid must be a multiple of cv / gcd(cv, 256)
    int g = ew_grid(nvec);
    if (g > cap) g = cap;
    g = (g + m - 1) / m * m;
    return g;
}
// reduction passes write one row of partials per workgroup: 768 workgroups (3 per CU) still stream at HBM speed and keep the
// fixed-order finalisation short
static inline int bn_bwd_grid(long nvec, int C) { return ew_grid_for(nvec, C, 768); }

// rows-per-block iteration for the stem passes: needs C/8 a power of two <= 256; ~1024 items per block
static inline bool stem_iter(StemRowIter& it, int C, int W, int H) {
    const int cv = C / 8;
    if (C % 8 || cv < 1 || cv > 256 || (cv & (cv - 1))) return false;
    it.W = W; it.cv = cv; it.cshift = 0;
    while ((1 << it.cshift) < cv) ++it.cshift;
    int rpb = 1024 / (W * cv);
    if (rpb < 1) rpb = 1;
    if (rpb > H) rpb = H;
    it.rpb = rpb;
    it.inv_wcv = 1.0f / (float)(W * cv);
    return true;
}

extern "C" {

int svsr_bn_finalize(const float* part, int nrows, int C, float count, float eps, float momentum, float* mean, float* rstd,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked, hipStream_t stream) {
    if (nrows < 1 || C < 1) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_bn_finalize, dim3((C + BNF_CL - 1) / BNF_CL), dim3(BNF_CL * BNF_RL), 0, stream, part, nrows, C, count, eps, momentum, mean, rstd,
                       running_mean, running_var, (long*)num_batches_tracked);
    return svsr_check_launch();
}

int svsr_bn_eval_prepare(const float* running_mean, const float* running_var, int C, float eps, float* mean, float* rstd,
                         hipStream_t stream) {
    hipLaunchKernelGGL(k_bn_eval_prepare, dim3((C + 127) / 128), dim3(128), 0, stream, running_mean, running_var, C, eps, mean, rstd);
    return svsr_check_launch();
}

int svsr_bn_act_fwd(const void* x, const void* res, void* y, const float* mean, const float* rstd, const float* gamma,
                    const float* beta, int64_t npix, int C, int act, hipStream_t stream) {
    if (!chan_ok_any(C)) return SVSR_ERR_ARG;
    const long nvec = npix * (C / 8);
    hipLaunchKernelGGL(k_bn_act_fwd, dim3(ew_grid_for(nvec, C)), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)res, (bf16_t*)y,
                       mean, rstd, gamma, beta, nvec, C, act);
    return svsr_check_launch();
}

/* rows of [2][C] partials svsr_bn_act_bwd needs in its workspace for this shape */
int svsr_bn_act_bwd_rows(int64_t npix, int C) { return chan_ok_any(C) && npix > 0 ? bn_bwd_grid(npix * (C / 8), C) : 0; }

int svsr_bn_act_bwd(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, const float* gamma,
                    float* slots, float* coef, float* dgamma, float* dbeta, void* dx, void* dres, int64_t npix, int C, int act,
                    const float* beta, const void* res, hipStream_t stream) {
    if (!chan_ok_any(C)) return SVSR_ERR_ARG;
    if (act == 2 && beta == nullptr) return SVSR_ERR_ARG;
    const long nvec = npix * (C / 8);
    const int grid = bn_bwd_grid(nvec, C);
    const int grid_apply = ew_grid_for(nvec, C);
#define SVSR_BN_BWD_REDUCE(A) hipLaunchKernelGGL(k_bn_act_bwd_reduce<A>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)y, \
                       (const bf16_t*)x, mean, rstd, nvec, C, slots, gamma, beta, (const bf16_t*)res)
#define SVSR_BN_BWD_APPLY(A) hipLaunchKernelGGL(k_bn_act_bwd_apply<A>, dim3(grid_apply), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)y, \
                       (const bf16_t*)x, mean, rstd, coef, (bf16_t*)dx, (bf16_t*)dres, nvec, C, gamma, beta, (const bf16_t*)res)
    if (act < 0 || act > 2) return SVSR_ERR_ARG;
    if (act == 2) SVSR_BN_BWD_REDUCE(2); else if (act == 1) SVSR_BN_BWD_REDUCE(1); else SVSR_BN_BWD_REDUCE(0);
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((C + BNF_CL - 1) / BNF_CL), dim3(BNF_CL * BNF_RL), 0, stream, slots, grid, C, (float)npix, gamma, rstd, dgamma, dbeta, coef);
    if (act == 2) SVSR_BN_BWD_APPLY(2); else if (act == 1) SVSR_BN_BWD_APPLY(1); else SVSR_BN_BWD_APPLY(0);
    return svsr_check_launch();
}

/* svsr_bn_bwd_from_stats: second half of a BatchNorm backward whose first pass was taken by the producing data-gradient launch
 * (svsr_igemm_dgrad_bn / svsr_conv3x3_c64_dgrad_bn): g = the masked gradient that launch stored, stats = its nrows x [2][C] partial sums
 * of {g, g * xhat}.  Adds the rows in a fixed order (dgamma += , dbeta +=, coef) and writes dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)). */
int svsr_bn_bwd_from_stats(const void* g, const void* x, const float* mean, const float* rstd, const float* gamma, const float* stats,
                           int nrows, float* coef, float* dgamma, float* dbeta, void* dx, int64_t npix, int C, hipStream_t stream) {
    if (!chan_ok_any(C) || nrows < 1 || npix < 1 || g == nullptr || x == nullptr || stats == nullptr || dx == nullptr) return SVSR_ERR_ARG;
    const long nvec = npix * (C / 8);
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((C + BNF_CL - 1) / BNF_CL), dim3(BNF_CL * BNF_RL), 0, stream, stats, nrows, C, (float)npix, gamma, rstd, dgamma, dbeta, coef);
    hipLaunchKernelGGL(k_bn_act_bwd_apply<0>, dim3(ew_grid_for(nvec, C)), dim3(256), 0, stream, (const bf16_t*)g, (const bf16_t*)nullptr,
                       (const bf16_t*)x, mean, rstd, coef, (bf16_t*)dx, (bf16_t*)nullptr, nvec, C, gamma, (const float*)nullptr, (const bf16_t*)nullptr);
    return svsr_check_launch();
}

int svsr_stem_bn_act_pool_fwd(const void* x, void* y, void* amax, const float* mean, const float* rstd, const float* gamma,
                              const float* beta, int N, int Hc, int Wc, int Hp, int Wp, int C, int act, void* xwin, hipStream_t stream) {
    if (!chan_ok(C) || (act != SVSR_ACT_GELU && act != SVSR_ACT_SWISH)) return SVSR_ERR_ARG;
    StemRowIter it;
    if (!stem_iter(it, C, Wp, Hp)) return SVSR_ERR_ARG;
    if (stem_lds_fwd() && xwin == nullptr) {
        const size_t lds = (size_t)(2 * STEM_PR + 1) * Wc * C * sizeof(float);
        if (lds <= 150 * 1024) {
            static bool attr = false;
            if (!attr) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_stem_fwd_lds<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_stem_fwd_lds<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                attr = true;
            }
            const dim3 g2((Hp + STEM_PR - 1) / STEM_PR, N);
            if (act == SVSR_ACT_SWISH)
                hipLaunchKernelGGL(k_stem_fwd_lds<2>, g2, dim3(256), lds, stream, (const bf16_t*)x, (bf16_t*)y, (unsigned char*)amax, mean, rstd,
                                   gamma, beta, Hc, Wc, Hp, Wp, C);
            else
                hipLaunchKernelGGL(k_stem_fwd_lds<1>, g2, dim3(256), lds, stream, (const bf16_t*)x, (bf16_t*)y, (unsigned char*)amax, mean, rstd,
                                   gamma, beta, Hc, Wc, Hp, Wp, C);
            return svsr_check_launch();
        }
    }
    const dim3 grid((Hp + it.rpb - 1) / it.rpb, N);
    if (act == SVSR_ACT_SWISH)
        hipLaunchKernelGGL(k_stem_bn_act_pool_fwd<2>, grid, dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, (unsigned char*)amax, (bf16_t*)xwin, mean,
                           rstd, gamma, beta, N, Hc, Wc, Hp, Wp, C, it);
    else
        hipLaunchKernelGGL(k_stem_bn_act_pool_fwd<1>, grid, dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, (unsigned char*)amax, (bf16_t*)xwin, mean,
                           rstd, gamma, beta, N, Hc, Wc, Hp, Wp, C, it);
    return svsr_check_launch();
}

static inline bool stem_bwd_uses_lds(int Wp, int C) { return stem_lds_bwd() && (size_t)(STEM_BR / 2 + 1) * Wp * C * 3 <= 60 * 1024; }
// gather-form reduce pass (k_stem_bwd_reduce_gather): tuning value 2, 64 channels, tile of 2*STEM_GP+1 conv rows within 64 KiB
static inline size_t stem_gather_lds(int Wc) { const size_t t = (size_t)(((2 * STEM_GP + 1) * Wc * 8 + 255) / 256 * 256) * 16; return t > 16384 ? t : 16384; }
static inline bool stem_bwd_gathers(int Wc, int Wp, int C) {
    return svsr_tune_get(SVSR_TUNE_STEM_LDS_BWD) >= 2 && C == 64 && stem_bwd_uses_lds(Wp, C) && stem_gather_lds(Wc) <= 64 * 1024;
}

/* rows of [2][C] partials svsr_stem_bn_act_pool_bwd needs in its workspace for this shape */
int svsr_stem_bn_act_pool_bwd_rows(int N, int Hc, int Wc, int C) {
    StemRowIter it;
    if (!chan_ok(C) || !stem_iter(it, C, Wc, Hc) || N < 1) return 0;
    const int Wp = (Wc - 1) / 2 + 1, Hp = (Hc - 1) / 2 + 1;
    if (stem_bwd_gathers(Wc, Wp, C)) return N * ((Hp + STEM_GP - 1) / STEM_GP);
    return N * (stem_bwd_uses_lds(Wp, C) ? (Hc + STEM_BR - 1) / STEM_BR : (Hc + it.rpb - 1) / it.rpb);
}

int svsr_stem_bn_act_pool_bwd(const void* dpool, const void* amax, const void* x, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, float* slots, float* coef, float* dgamma, float* dbeta,
                              void* dx, int N, int Hc, int Wc, int Hp, int Wp, int C, int act, const void* xwin, void* gpool, hipStream_t stream) {
    if (!chan_ok(C) || (act != SVSR_ACT_GELU && act != SVSR_ACT_SWISH)) return SVSR_ERR_ARG;
    StemRowIter it;
    if (!stem_iter(it, C, Wc, Hc)) return SVSR_ERR_ARG;
    const size_t lds_b = (size_t)(STEM_BR / 2 + 1) * Wp * C * 3;
    const int nrows = svsr_stem_bn_act_pool_bwd_rows(N, Hc, Wc, C);
    if (dx == nullptr && !(stem_bwd_uses_lds(Wp, C) && stem_bwd_gathers(Wc, Wp, C) && xwin != nullptr && gpool != nullptr)) return SVSR_ERR_ARG;
    if (stem_bwd_uses_lds(Wp, C)) {
        const dim3 g2((Hc + STEM_BR - 1) / STEM_BR, N);
#define SVSR_STEM_BWD(A, AP) hipLaunchKernelGGL((k_stem_bwd_lds<A, AP>), g2, dim3(256), lds_b, stream, (const bf16_t*)dpool, (const unsigned char*)amax, \
                       (const bf16_t*)x, mean, rstd, gamma, beta, coef, (bf16_t*)dx, slots, Hc, Wc, Hp, Wp, C)
        if (stem_bwd_gathers(Wc, Wp, C) && xwin != nullptr && gpool != nullptr) {
            // winners kept by the forward: reduce pass over pooled outputs only, apply pass without activation derivatives
            if (Hp != (Hc - 1) / 2 + 1) return SVSR_ERR_ARG;
            const dim3 gg((Hp + STEM_GPW - 1) / STEM_GPW, N);
            const int nrows_w = (int)(gg.x * gg.y);            // (<= nrows, the gather form's count the workspace is sized for)
            if (act == SVSR_ACT_SWISH)
                hipLaunchKernelGGL(k_stem_bwd_reduce_win<2>, gg, dim3(256), 0, stream, (const bf16_t*)dpool, (const bf16_t*)xwin, mean, rstd, gamma, beta,
                                   (bf16_t*)gpool, slots, Hp, Wp);
            else
                hipLaunchKernelGGL(k_stem_bwd_reduce_win<1>, gg, dim3(256), 0, stream, (const bf16_t*)dpool, (const bf16_t*)xwin, mean, rstd, gamma, beta,
                                   (bf16_t*)gpool, slots, Hp, Wp);
            hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((C + BNF_CL - 1) / BNF_CL), dim3(BNF_CL * BNF_RL), 0, stream, slots, nrows_w, C, (float)((long)N * Hc * Wc), gamma, rstd,
                               dgamma, dbeta, coef);
            if (dx == nullptr) return svsr_check_launch();      // the apply pass runs inside svsr_stem_bwd_wgrad (stem.hip): gpool and coef are its inputs
            if (act == SVSR_ACT_SWISH)
                hipLaunchKernelGGL((k_stem_bwd_lds<2, true, true>), g2, dim3(256), lds_b, stream, (const bf16_t*)gpool, (const unsigned char*)amax, (const bf16_t*)x,
                                   mean, rstd, gamma, beta, coef, (bf16_t*)dx, slots, Hc, Wc, Hp, Wp, C);
            else
                hipLaunchKernelGGL((k_stem_bwd_lds<1, true, true>), g2, dim3(256), lds_b, stream, (const bf16_t*)gpool, (const unsigned char*)amax, (const bf16_t*)x,
                                   mean, rstd, gamma, beta, coef, (bf16_t*)dx, slots, Hc, Wc, Hp, Wp, C);
            return svsr_check_launch();
        }
        if (stem_bwd_gathers(Wc, Wp, C)) {
            if (Hp != (Hc - 1) / 2 + 1) return SVSR_ERR_ARG;
            const dim3 gg((Hp + STEM_GP - 1) / STEM_GP, N);
            const size_t lds_g = stem_gather_lds(Wc);
            if (act == SVSR_ACT_SWISH)
                hipLaunchKernelGGL(k_stem_bwd_reduce_gather<2>, gg, dim3(256), lds_g, stream, (const bf16_t*)dpool, (const unsigned char*)amax,
                                   (const bf16_t*)x, mean, rstd, gamma, beta, slots, Hc, Wc, Hp, Wp);
            else
                hipLaunchKernelGGL(k_stem_bwd_reduce_gather<1>, gg, dim3(256), lds_g, stream, (const bf16_t*)dpool, (const unsigned char*)amax,
                                   (const bf16_t*)x, mean, rstd, gamma, beta, slots, Hc, Wc, Hp, Wp);
        } else if (act == SVSR_ACT_SWISH) SVSR_STEM_BWD(2, false); else SVSR_STEM_BWD(1, false);
        hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((C + BNF_CL - 1) / BNF_CL), dim3(BNF_CL * BNF_RL), 0, stream, slots, nrows, C, (float)((long)N * Hc * Wc), gamma, rstd,
                           dgamma, dbeta, coef);
        if (act == SVSR_ACT_SWISH) SVSR_STEM_BWD(2, true); else SVSR_STEM_BWD(1, true);
        return svsr_check_launch();
    }
    const dim3 grid((Hc + it.rpb - 1) / it.rpb, N);
    if (act == SVSR_ACT_SWISH)
        hipLaunchKernelGGL(k_stem_pool_bwd_reduce<2>, grid, dim3(256), 0, stream, (const bf16_t*)dpool, (const unsigned char*)amax,
                           (const bf16_t*)x, mean, rstd, gamma, beta, N, Hc, Wc, Hp, Wp, C, slots, it);
    else
        hipLaunchKernelGGL(k_stem_pool_bwd_reduce<1>, grid, dim3(256), 0, stream, (const bf16_t*)dpool, (const unsigned char*)amax,
                           (const bf16_t*)x, mean, rstd, gamma, beta, N, Hc, Wc, Hp, Wp, C, slots, it);
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((C + BNF_CL - 1) / BNF_CL), dim3(BNF_CL * BNF_RL), 0, stream, slots, nrows, C, (float)((long)N * Hc * Wc), gamma, rstd,
                       dgamma, dbeta, coef);
    if (act == SVSR_ACT_SWISH)
        hipLaunchKernelGGL(k_stem_pool_bwd_apply<2>, grid, dim3(256), 0, stream, (const bf16_t*)dpool, (const unsigned char*)amax,
                           (const bf16_t*)x, mean, rstd, gamma, beta, coef, (bf16_t*)dx, N, Hc, Wc, Hp, Wp, C, it);
    else
        hipLaunchKernelGGL(k_stem_pool_bwd_apply<1>, grid, dim3(256), 0, stream, (const bf16_t*)dpool, (const unsigned char*)amax,
                           (const bf16_t*)x, mean, rstd, gamma, beta, coef, (bf16_t*)dx, N, Hc, Wc, Hp, Wp, C, it);
    return svsr_check_launch();
}

int svsr_avgpool_fwd(const void* x, void* y, int64_t N, int HW, int C, hipStream_t stream) {
    if (C % 8) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_avgpool_fwd, dim3(ew_grid(N * (C / 8))), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, (long)N, HW, C);
    return svsr_check_launch();
}

int svsr_avgpool_bwd(const void* dy, void* dx, int64_t N, int HW, int C, hipStream_t stream) {
    if (C % 8) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_avgpool_bwd, dim3(ew_grid(N * HW * (C / 8))), dim3(256), 0, stream, (const bf16_t*)dy, (bf16_t*)dx, (long)N, HW, C);
    return svsr_check_launch();
}

}  // extern "C"
