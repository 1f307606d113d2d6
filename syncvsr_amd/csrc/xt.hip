// Row passes of the LRW model's `type: x-transformers` encoder (gfx950): RMSNorm, rotary position embedding, the GEGLU gate,
// and the [CLS] / word-boundary concatenation in front of the encoder.  All are HBM-bound streaming passes over [rows][ld]
// bf16 activations; the contractions around them are svsr_igemm_fwd / svsr_igemm_wgrad.
//
// The reference builds this encoder from the third-party `x_transformers.Encoder` (reference LRW/video/src/lightning.py:93-105;
// shipped configs LRW/video/config/bert-12l-512d_LRW_96_bf16_rrc_{WB,noWB}.yaml:18-29).  That package is not vendored under the
// reference tree and cannot be imported offline, so the arithmetic below restates its published algorithm (pre-norm residual
// blocks: RMSNorm -> attention with rotary embedding on the first 32 of 64 head dims / RMSNorm -> GEGLU feed-forward) and its
// parity is UNPINNED; oracle/lrw_oracle.py::xt_encoder is the restatement the kernels are tested against.
//
// Widths that are not multiples of 64 (513 = 512 + the word-boundary channel; 2052 = 4 * 513) live in rows padded to the
// next multiple of 64 whose pad columns are kept at exactly zero by every kernel here, so that the contraction kernels can
// treat the padded width as the real one.
#include "common.h"

extern "C" int svsr_colsum_rows(const float* ws, int nrows, int64_t ld, float* out0, int64_t n0, float* out1, int64_t n1, int accumulate,
                                float scale, hipStream_t stream);

// ---------------------------------------------------------------------------------------------------------------------
// RMSNorm (x-transformers' `RMSNorm`: y = x / max(||x||_2 * D^-0.5, eps) * g).  One wave per row.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rmsnorm_fwd(const bf16_t* __restrict__ x, const float* __restrict__ g, bf16_t* __restrict__ y,
                                                     float* __restrict__ inv_out, int R, int D, int ld, float scale, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const bf16_t* xr = x + (long)row * ld;
    float ss = 0.f;
    for (int c0 = lane * 8; c0 < ld; c0 += 512) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(xr + c0), f);
#pragma unroll
        for (int k = 0; k < 8; ++k) ss += (c0 + k < D) ? f[k] * f[k] : 0.f;
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / fmaxf(sqrtf(ss) * scale, eps);
    if (lane == 0) inv_out[row] = inv;
    bf16_t* yr = y + (long)row * ld;
    for (int c0 = lane * 8; c0 < ld; c0 += 512) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(xr + c0), f);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = (c0 + k < D) ? f[k] * inv * g[c0 + k] : 0.f;
        *reinterpret_cast<u32x4*>(yr + c0) = pack8(f);
    }
}

// dx = g * inv * dy - x * inv^3 * scale^2 * sum_c(dy_c g_c x_c)  (+ addend: the gradient that reaches x past the block);
// part[block][c] = sum over the block's rows of dy * x * inv (the gain gradient), rows added in increasing order.
// Block = 4 waves, RPB rows: wave w takes rows w, w + 4, ...
#define RMS_MAXV 2          // 8-column vectors per lane: ld <= 1024
__global__ __launch_bounds__(256) void k_rmsnorm_bwd(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ g,
                                                     const float* __restrict__ inv_in, const bf16_t* __restrict__ addend,
                                                     bf16_t* __restrict__ dx, float* __restrict__ part, int R, int D, int ld, float scale,
                                                     int rpb) {
    __shared__ float sred[4][RMS_MAXV * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rpb;
    int r1 = r0 + rpb;
    if (r1 > R) r1 = R;
    float dg[RMS_MAXV][8];
#pragma unroll
    for (int v = 0; v < RMS_MAXV; ++v)
#pragma unroll
        for (int k = 0; k < 8; ++k) dg[v][k] = 0.f;
    for (int row = r0 + wave; row < r1; row += 4) {
        const long base = (long)row * ld;
        const float inv = inv_in[row];
        float fx[RMS_MAXV][8], fd[RMS_MAXV][8];
        float dot = 0.f;
#pragma unroll
        for (int v = 0; v < RMS_MAXV; ++v) {
            const int c0 = lane * 8 + v * 512;
            if (c0 < ld) {
                unpack8(*reinterpret_cast<const u32x4*>(x + base + c0), fx[v]);
                unpack8(*reinterpret_cast<const u32x4*>(dy + base + c0), fd[v]);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float gk = (c0 + k < D) ? g[c0 + k] : 0.f;
                    dot += fd[v][k] * gk * fx[v][k];
                    dg[v][k] += fd[v][k] * fx[v][k] * inv;
                    fd[v][k] *= gk * inv;
                }
            }
        }
        dot = wave_sum(dot);
        const float coef = inv * inv * inv * scale * scale * dot;
#pragma unroll
        for (int v = 0; v < RMS_MAXV; ++v) {
            const int c0 = lane * 8 + v * 512;
            if (c0 < ld) {
                float a[8];
                if (addend != nullptr) unpack8(*reinterpret_cast<const u32x4*>(addend + base + c0), a);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float o = fd[v][k] - fx[v][k] * coef + (addend != nullptr ? a[k] : 0.f);
                    fd[v][k] = (c0 + k < D) ? o : 0.f;
                }
                *reinterpret_cast<u32x4*>(dx + base + c0) = pack8(fd[v]);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < RMS_MAXV; ++v)
#pragma unroll
        for (int k = 0; k < 8; ++k) sred[wave][v * 512 + lane * 8 + k] = dg[v][k];
    __syncthreads();
    for (int c = threadIdx.x; c < ld; c += 256)
        part[(long)blockIdx.x * ld + c] = ((sred[0][c] + sred[1][c]) + sred[2][c]) + sred[3][c];
}

// ---------------------------------------------------------------------------------------------------------------------
// Rotary position embedding on the first `rot` dims of every 64-wide head of q, k (and v), in place:
//   out[j] = a cos - b sin, out[j + rot/2] = b cos + a sin   (a = x[j], b = x[j + rot/2]; rotate_half convention)
// tab = fp32 [S][rot]: cos(s * f_j) for j < rot/2, then sin(s * f_j).  sign = -1 applies the transpose (backward).
// One thread per (row, head of one of the `ntens` tensors); row pitch `ld`, tensor t at column t * E.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rotary(bf16_t* __restrict__ qkv, const float* __restrict__ tab, int R, int S, int heads_total, int ld,
                                                float sign) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)R * heads_total) return;
    const int row = (int)(idx / heads_total), h = (int)(idx - (long)row * heads_total);
    const int s = row % S;
    bf16_t* p = qkv + (long)row * ld + h * 64;
    const float* cs = tab + (long)s * 32;
    float a[16], b[16];
    unpack8(*reinterpret_cast<const u32x4*>(p), a);
    unpack8(*reinterpret_cast<const u32x4*>(p + 8), a + 8);
    unpack8(*reinterpret_cast<const u32x4*>(p + 16), b);
    unpack8(*reinterpret_cast<const u32x4*>(p + 24), b + 8);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float c = cs[j], sn = cs[16 + j] * sign;
        const float oa = a[j] * c - b[j] * sn, ob = b[j] * c + a[j] * sn;
        a[j] = oa; b[j] = ob;
    }
    *reinterpret_cast<u32x4*>(p) = pack8(a);
    *reinterpret_cast<u32x4*>(p + 8) = pack8(a + 8);
    *reinterpret_cast<u32x4*>(p + 16) = pack8(b);
    *reinterpret_cast<u32x4*>(p + 24) = pack8(b + 8);
}

// ---------------------------------------------------------------------------------------------------------------------
// GEGLU gate (x-transformers' `GLU` with GELU + the feed-forward's dropout):  u = [value | gate] [R][ldu], halves I wide.
//   y[r][j] = dropout(value * bf16(gelu(gate)))  for j < I, 0 for I <= j < ldy.     One thread per 4 columns.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load4(const bf16_t* p, float* f) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
}
__device__ __forceinline__ void store4(bf16_t* p, const float* f) {
    uint2 r;
    r.x = pack2bf(f[0], f[1]); r.y = pack2bf(f[2], f[3]);
    *reinterpret_cast<uint2*>(p) = r;
}

__global__ __launch_bounds__(256) void k_geglu_fwd(const bf16_t* __restrict__ u, bf16_t* __restrict__ y, long R, int I, int ldu, int ldy,
                                                   DropArgs drop) {
    const int vy = ldy >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= R * vy) return;
    const long row = idx / vy;
    const int j = (int)(idx - row * vy) * 4;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (j < I) {
        float a[4], gt[4];
        load4(u + row * ldu + j, a);
        load4(u + row * ldu + I + j, gt);
        const bool on = drop.seed != nullptr;
        const unsigned key = on ? drop_key(drop) : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = bf2f(f2bf(a[k] * bf2f(f2bf(gelu_erf(gt[k])))));          // both products round to bf16 under autocast
            if (on) v = drop_keep(key, drop.thresh, (unsigned)(row * ldy + j + k)) ? v * drop.scale : 0.f;
            o[k] = v;
        }
    }
    store4(y + row * ldy + j, o);
}

// du[:, c] for c < I: dy * gelu(gate);  I <= c < 2I: dy * value * gelu'(gate);  0 beyond.  dy passes the forward's mask first.
__global__ __launch_bounds__(256) void k_geglu_bwd(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ u, bf16_t* __restrict__ du, long R,
                                                   int I, int ldu, int ldy, DropArgs drop) {
    const int vu = ldu >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= R * vu) return;
    const long row = idx / vu;
    const int c = (int)(idx - row * vu) * 4;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < 2 * I) {
        const int j = c < I ? c : c - I;
        float d[4], a[4], gt[4];
        load4(dy + row * ldy + j, d);
        load4(u + row * ldu + j, a);
        load4(u + row * ldu + I + j, gt);
        const bool on = drop.seed != nullptr;
        const unsigned key = on ? drop_key(drop) : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float dk = d[k];
            if (on) dk = drop_keep(key, drop.thresh, (unsigned)(row * ldy + j + k)) ? dk * drop.scale : 0.f;
            o[k] = c < I ? dk * bf2f(f2bf(gelu_erf(gt[k]))) : dk * a[k] * gelu_erf_grad(gt[k]);
        }
    }
    store4(du + row * ldu + c, o);
}

// ---------------------------------------------------------------------------------------------------------------------
// x0[b, 0] = cls; x0[b, 1 + t] = [feats[b, t] (F wide) | word_mask[b, t]] ; emb_dropout on all of it; pad columns zero.
// (reference lightning.py:145-150).  One thread per 8 columns of a row.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_xt_embed_fwd(const bf16_t* __restrict__ feats, const float* __restrict__ wmask, const float* __restrict__ cls,
                                                      bf16_t* __restrict__ x0, int B, int S, int F, int D, int ld, DropArgs drop) {
    const int cv = ld >> 3;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * S * cv) return;
    const long row = idx / cv;
    const int c0 = (int)(idx - row * cv) * 8;
    const int b = (int)(row / S), s = (int)(row - (long)b * S);
    float f[8];
    if (s == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = (c0 + k < D) ? cls[c0 + k] : 0.f;
    } else if (c0 + 8 <= F) {
        unpack8(*reinterpret_cast<const u32x4*>(feats + ((long)b * (S - 1) + s - 1) * F + c0), f);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = (c0 + k == F && F < D && wmask != nullptr) ? wmask[(long)b * (S - 1) + s - 1] : 0.f;
    }
    if (drop.seed != nullptr) {
        const unsigned key = drop_key(drop);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = drop_keep(key, drop.thresh, (unsigned)(row * ld + c0 + k)) ? f[k] * drop.scale : 0.f;
    }
    *reinterpret_cast<u32x4*>(x0 + row * ld + c0) = pack8(f);
}

// dfeats[b, t] = mask * dx0[b, 1 + t, :F];  dcls[c] += sum_b mask * dx0[b, 0, c] (b in increasing order).
__global__ __launch_bounds__(256) void k_xt_embed_bwd(const bf16_t* __restrict__ dx0, bf16_t* __restrict__ dfeats, float* __restrict__ dcls, int B,
                                                      int S, int F, int D, int ld, DropArgs drop) {
    const bool on = drop.seed != nullptr;
    const unsigned key = on ? drop_key(drop) : 0u;
    const int cvf = F >> 3, cv = ld >> 3;
    const long nfeat = (long)B * (S - 1) * cvf;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < nfeat) {
        const long frow = idx / cvf;
        const int c0 = (int)(idx - frow * cvf) * 8;
        const int b = (int)(frow / (S - 1)), t = (int)(frow - (long)b * (S - 1));
        const long row = (long)b * S + 1 + t;
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(dx0 + row * ld + c0), f);
        if (on) {
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = drop_keep(key, drop.thresh, (unsigned)(row * ld + c0 + k)) ? f[k] * drop.scale : 0.f;
        }
        *reinterpret_cast<u32x4*>(dfeats + frow * F + c0) = pack8(f);
    } else if (idx < nfeat + cv) {
        const int c0 = (int)(idx - nfeat) * 8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) {
            const long row = (long)b * S;
            float f[8];
            unpack8(*reinterpret_cast<const u32x4*>(dx0 + row * ld + c0), f);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (on) f[k] = drop_keep(key, drop.thresh, (unsigned)(row * ld + c0 + k)) ? f[k] * drop.scale : 0.f;
                acc[k] += f[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (c0 + k < D) dcls[c0 + k] += acc[k];
    }
}

extern "C" {

int svsr_rmsnorm_fwd(const void* x, const float* g, void* y, float* inv, int R, int D, int ld, float eps, hipStream_t stream) {
    if (R < 1 || D < 1 || ld < D || ld % 8 != 0 || ld > 512 * RMS_MAXV) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_rmsnorm_fwd, dim3((R + 3) / 4), dim3(256), 0, stream, (const bf16_t*)x, g, (bf16_t*)y, inv, R, D, ld,
                       1.0f / sqrtf((float)D), eps);
    return svsr_check_launch();
}

static int rms_rpb(int R) { return R >= 4096 ? 32 : 16; }

/* rows of the gain-gradient workspace svsr_rmsnorm_bwd needs: part = [rows][ld] floats */
int svsr_rmsnorm_bwd_rows(int R) { return R < 1 ? -SVSR_ERR_ARG : (R + rms_rpb(R) - 1) / rms_rpb(R); }

int svsr_rmsnorm_bwd(const void* dy, const void* x, const float* g, const float* inv, const void* addend, void* dx, float* dg, float* part,
                     int R, int D, int ld, hipStream_t stream) {
    if (R < 1 || D < 1 || ld < D || ld % 8 != 0 || ld > 512 * RMS_MAXV || part == nullptr) return SVSR_ERR_ARG;
    const int rpb = rms_rpb(R), grid = (R + rpb - 1) / rpb;
    hipLaunchKernelGGL(k_rmsnorm_bwd, dim3(grid), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, g, inv, (const bf16_t*)addend,
                       (bf16_t*)dx, part, R, D, ld, 1.0f / sqrtf((float)D), rpb);
    int rc = svsr_check_launch();
    if (rc != SVSR_OK) return rc;
    return svsr_colsum_rows(part, grid, ld, dg, D, nullptr, 0, 1, 1.0f, stream);
}

/* qkv: [R][ld] bf16 holding `heads_total` consecutive 64-wide heads from column 0 (q heads, k heads, v heads ...); the first 32
 * dims of every head are rotated in place by position s = row % S.  tab: fp32 [S][32] (16 cosines, 16 sines).  sign: +1 / -1. */
int svsr_rotary(void* qkv, const float* tab, int R, int S, int heads_total, int ld, int sign, hipStream_t stream) {
    if (R < 1 || S < 1 || R % S != 0 || heads_total < 1 || ld % 8 != 0 || ld < heads_total * 64 || (sign != 1 && sign != -1)) return SVSR_ERR_ARG;
    const long n = (long)R * heads_total;
    hipLaunchKernelGGL(k_rotary, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (bf16_t*)qkv, tab, R, S, heads_total, ld, (float)sign);
    return svsr_check_launch();
}

int svsr_geglu_fwd(const void* u, void* y, int R, int I, int ldu, int ldy, const unsigned* drop_seed, unsigned drop_site, float drop_p,
                   hipStream_t stream) {
    if (R < 1 || I < 4 || I % 4 != 0 || ldu < 2 * I || ldy < I || ldu % 4 != 0 || ldy % 4 != 0 || (long)R * ldy >= (1L << 32)) return SVSR_ERR_ARG;
    const long n = (long)R * (ldy / 4);
    hipLaunchKernelGGL(k_geglu_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)u, (bf16_t*)y, (long)R, I, ldu, ldy,
                       svsr_make_drop(drop_seed, drop_site, drop_p));
    return svsr_check_launch();
}

int svsr_geglu_bwd(const void* dy, const void* u, void* du, int R, int I, int ldu, int ldy, const unsigned* drop_seed, unsigned drop_site,
                   float drop_p, hipStream_t stream) {
    if (R < 1 || I < 4 || I % 4 != 0 || ldu < 2 * I || ldy < I || ldu % 4 != 0 || ldy % 4 != 0 || (long)R * ldy >= (1L << 32)) return SVSR_ERR_ARG;
    const long n = (long)R * (ldu / 4);
    hipLaunchKernelGGL(k_geglu_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)u, (bf16_t*)du,
                       (long)R, I, ldu, ldy, svsr_make_drop(drop_seed, drop_site, drop_p));
    return svsr_check_launch();
}

/* feats bf16 [B*(S-1)][F]; wmask fp32 [B][S-1] or NULL; cls fp32 [>= D]; x0 bf16 [B*S][ld].  D = F (+ 1 with the word boundary). */
int svsr_xt_embed_fwd(const void* feats, const float* wmask, const float* cls, void* x0, int B, int S, int F, int D, int ld,
                      const unsigned* drop_seed, unsigned drop_site, float drop_p, hipStream_t stream) {
    if (B < 1 || S < 2 || F % 8 != 0 || (D != F && D != F + 1) || ld < D || ld % 8 != 0 || (D == F + 1 && wmask == nullptr)) return SVSR_ERR_ARG;
    const long n = (long)B * S * (ld / 8);
    hipLaunchKernelGGL(k_xt_embed_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)feats, wmask, cls, (bf16_t*)x0, B, S,
                       F, D, ld, svsr_make_drop(drop_seed, drop_site, drop_p));
    return svsr_check_launch();
}

int svsr_xt_embed_bwd(const void* dx0, void* dfeats, float* dcls, int B, int S, int F, int D, int ld, const unsigned* drop_seed,
                      unsigned drop_site, float drop_p, hipStream_t stream) {
    if (B < 1 || S < 2 || F % 8 != 0 || (D != F && D != F + 1) || ld < D || ld % 8 != 0) return SVSR_ERR_ARG;
    const long n = (long)B * (S - 1) * (F / 8) + ld / 8;
    hipLaunchKernelGGL(k_xt_embed_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)dx0, (bf16_t*)dfeats, dcls, B, S, F, D,
                       ld, svsr_make_drop(drop_seed, drop_site, drop_p));
    return svsr_check_launch();
}

}  // extern "C"
