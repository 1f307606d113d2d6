// Cross-entropy heads, accuracy metric, and the optimiser-side passes (gfx950).
//   * row cross-entropy with class-index or probability targets and label smoothing — the word loss and the
//     SyncVSR audio-token loss over (B*T*A*G) rows x V classes (reference LRW/video/src/lightning.py:161-171,
//     README.md:47-53)
//   * top-1 / top-5 accuracy (lightning.py:177-183)
//   * global-norm clip + AdamW + cosine schedule + bf16 shadow refresh (lightning.py:216-223)
//   * fp32 -> bf16 shadow casts / transposes of the weights the MFMA kernels consume
#include "common.h"

// one wave per row; logits may be bf16 or f32; V <= 64*CE_MAXPER
#define CE_MAXPER 16

struct CeArgs {
    const void* logits; int logits_f32; int ld;   // row pitch in elements
    const long* target_idx;                        // hard targets [R] or null
    const float* target_prob; int ldt;             // soft targets [R][V] or null
    int R, V;
    float smoothing;
    float* row_loss;       // [R] per-row losses (svsr_colsum_rows adds them in a fixed order: loss = mean)
    float* lse;            // [R] saved for the backward
};

__device__ __forceinline__ float ce_load(const CeArgs& a, long off) {
    return a.logits_f32 ? reinterpret_cast<const float*>(a.logits)[off] : bf2f(reinterpret_cast<const bf16_t*>(a.logits)[off]);
}

__global__ __launch_bounds__(256) void k_ce_fwd(const CeArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < a.R; row += gridDim.x * 4) {
        float z[CE_MAXPER];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < CE_MAXPER; ++i) {
            const int v = i * 64 + lane;
            z[i] = v < a.V ? ce_load(a, (long)row * a.ld + v) : -INFINITY;
            m = fmaxf(m, z[i]);
        }
        m = wave_max(m);
        float se = 0.f, sz = 0.f;
#pragma unroll
        for (int i = 0; i < CE_MAXPER; ++i) {
            const int v = i * 64 + lane;
            if (v < a.V) { se += expf(z[i] - m); sz += z[i]; }
        }
        se = wave_sum(se);
        const float lse = m + logf(se);
        float loss;
        if (a.target_idx != nullptr) {
            sz = wave_sum(sz);
            const long t = a.target_idx[row];
            // a target outside [0, V) (torch raises a device assert) poisons the loss instead of reading out of bounds
            const float zt = (t >= 0 && t < a.V) ? ce_load(a, (long)row * a.ld + t) : __uint_as_float(0x7fc00000u);
            loss = (1.f - a.smoothing) * (lse - zt) + a.smoothing * (lse - sz / (float)a.V);
        } else {
            float st = 0.f, stz = 0.f;
#pragma unroll
            for (int i = 0; i < CE_MAXPER; ++i) {
                const int v = i * 64 + lane;
                if (v < a.V) {
                    const float t = a.target_prob[(long)row * a.ldt + v] * (1.f - a.smoothing) + a.smoothing / (float)a.V;
                    st += t; stz += t * z[i];
                }
            }
            st = wave_sum(st); stz = wave_sum(stz);
            loss = lse * st - stz;
        }
        if (lane == 0) { a.lse[row] = lse; a.row_loss[row] = loss; }
    }
}

// dlogits = gout * (softmax * sum(t') - t') / R      (bf16 out, pitch ldo)
__global__ __launch_bounds__(256) void k_ce_bwd(const CeArgs a, const float* gout, bf16_t* dlogits, int ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float g = gout[0] / (float)a.R;
    for (int row = blockIdx.x * 4 + wave; row < a.R; row += gridDim.x * 4) {
        const float lse = a.lse[row];
        if (a.target_idx != nullptr) {
            const long t = a.target_idx[row];
            for (int v = lane; v < a.V; v += 64) {
                const float p = expf(ce_load(a, (long)row * a.ld + v) - lse);
                const float tt = (v == t ? 1.f - a.smoothing : 0.f) + a.smoothing / (float)a.V;
                dlogits[(long)row * ldo + v] = f2bf(g * (p - tt));
            }
        } else {
            float st = 0.f;
            for (int v = lane; v < a.V; v += 64)
                st += a.target_prob[(long)row * a.ldt + v] * (1.f - a.smoothing) + a.smoothing / (float)a.V;
            st = wave_sum(st);
            for (int v = lane; v < a.V; v += 64) {
                const float p = expf(ce_load(a, (long)row * a.ld + v) - lse);
                const float tt = a.target_prob[(long)row * a.ldt + v] * (1.f - a.smoothing) + a.smoothing / (float)a.V;
                dlogits[(long)row * ldo + v] = f2bf(g * (p * st - tt));
            }
        }
        for (int v = a.V + lane; v < ldo; v += 64) dlogits[(long)row * ldo + v] = (bf16_t)0;      // pad columns of the row (ldo > V): zeros, no memset in front of the launch
    }
}

// top-1 / top-5 accuracy of f32 logits [B][C]; labels are class indices, or (soft) the argmax of [B][C] probabilities
__global__ __launch_bounds__(256) void k_topk_acc(const float* __restrict__ logits, const long* __restrict__ labels,
                                                  const float* __restrict__ soft, int B, int C, float* rows2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < B; row += gridDim.x * 4) {
        long lab;
        if (labels != nullptr) lab = labels[row];
        else {
            float best = -INFINITY; int bi = 0;
            for (int v = lane; v < C; v += 64) { const float p = soft[(long)row * C + v]; if (p > best) { best = p; bi = v; } }
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            lab = bi;
        }
        if (lab < 0 || lab >= C) lab = 0;
        const float zl = logits[(long)row * C + lab];
        float cnt = 0.f;
        for (int v = lane; v < C; v += 64) {
            const float z = logits[(long)row * C + v];
            if (z > zl || (z == zl && v < lab)) cnt += 1.f;
        }
        cnt = wave_sum(cnt);
        if (lane == 0) { rows2[row * 2 + 0] = cnt < 0.5f ? 1.f : 0.f; rows2[row * 2 + 1] = cnt < 4.5f ? 1.f : 0.f; }
    }
}

// ---------------------------------------------------------------------------------------------------------
// optimiser
// ---------------------------------------------------------------------------------------------------------
#define OPT_PARTS 1024      // workgroups of k_grad_sumsq = partial sums of squares kept in the device state
struct OptState { int step; int skipped; float lr_last; float gnorm_last; float part[OPT_PARTS]; };   // device-resident; skipped: steps whose gradient norm was not finite (no update, step not advanced)

// sum of the OPT_PARTS partials in a fixed order (every thread of a 256-thread block gets the same value)
__device__ __forceinline__ float opt_sumsq(const OptState* st, float* s4) {
    const int t = threadIdx.x;
    float v = ((st->part[t] + st->part[t + 256]) + st->part[t + 512]) + st->part[t + 768];
    v = wave_sum(v);                       // xor-butterfly: the same association on every lane and every launch
    if ((t & 63) == 0) s4[t >> 6] = v;
    __syncthreads();
    return ((s4[0] + s4[1]) + s4[2]) + s4[3];
}

__global__ __launch_bounds__(256) void k_grad_sumsq(const float* __restrict__ g, long n, OptState* st, int part0) {
    __shared__ float spart[4];
    float acc = 0.f;
    const long nv = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (long i = (nv << 2) + threadIdx.x; i < n; i += 256) acc += g[i] * g[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) spart[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) st->part[part0 + blockIdx.x] = ((spart[0] + spart[1]) + spart[2]) + spart[3];
}

struct AdamArgs {
    float* p; const float* g; float* m; float* v; bf16_t* shadow;
    long n, decay_end;       // elements [0, decay_end) get weight decay (ndim >= 2 parameters)
    float lr, beta1, beta2, eps, weight_decay, max_norm;
    int warmup, total_steps;  // cosine schedule with linear warm-up; total_steps <= 0 -> constant lr
    OptState* st;
};

__device__ __forceinline__ float sched_lr(const AdamArgs& a, int step /* 0-based optimiser step index */) {
    if (a.total_steps <= 0) return a.lr;
    if (step < a.warmup) return a.lr * (float)step / (float)max(1, a.warmup);
    const float prog = (float)(step - a.warmup) / (float)max(1, a.total_steps - a.warmup);
    return a.lr * fmaxf(0.f, 0.5f * (1.f + cosf(3.14159265358979323846f * prog)));
}

__global__ __launch_bounds__(256) void k_adamw(const AdamArgs a) {
    __shared__ float s4[4];
    const int step = a.st->step;             // steps already taken
    const float gnorm = sqrtf(opt_sumsq(a.st, s4));
    // A gradient norm that is not finite (a fused-encoder launch whose cluster wait gave up poisons its outputs with NaN: csrc/enc_fused.hip;
    // an overflow) would turn every parameter and both moments into NaN for good.  Such a step is SKIPPED: nothing is written, the step counter
    // does not advance (k_opt_advance counts it in `skipped`, TrainStep.state() reports it).  The reference (AdamW behind
    // clip_grad_norm_, lightning.py:216-223) would keep training on NaN parameters; there is nothing to be identical to.
    if (!(gnorm <= 3.0e38f)) return;
    const float clip = a.max_norm > 0.f ? fminf(1.f, a.max_norm / (gnorm + 1e-6f)) : 1.f;
    const float lr = sched_lr(a, step);
    const float t = (float)(step + 1);
    const float bc1 = 1.f - powf(a.beta1, t), bc2 = 1.f - powf(a.beta2, t);
    const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256) {
        const float g = a.g[i] * clip;
        float p = a.p[i];
        if (i < a.decay_end) p *= 1.f - lr * a.weight_decay;
        const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * g;
        const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
        p -= step_size * m / (sqrtf(v) * inv_sqrt_bc2 + a.eps);
        a.m[i] = m; a.v[i] = v; a.p[i] = p;
        if (a.shadow != nullptr) a.shadow[i] = f2bf(p);
    }
}

__global__ __launch_bounds__(256) void k_opt_advance(OptState* st, float lr_base, int warmup, int total_steps) {
    __shared__ float s4[4];
    const float sumsq = opt_sumsq(st, s4);
    if (threadIdx.x != 0) return;
    AdamArgs a; a.lr = lr_base; a.warmup = warmup; a.total_steps = total_steps;
    st->lr_last = sched_lr(a, st->step);
    st->gnorm_last = sqrtf(sumsq);
    if (st->gnorm_last <= 3.0e38f) st->step += 1;
    else st->skipped += 1;
}

// ---------------------------------------------------------------------------------------------------------
// shadow casts
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cast_bf16(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = f2bf(src[i]);
}

// table entry: src fp32 [A][T][Bd] at src_off  ->  dst bf16 [Bd][T][Apad] at dst_off   (T taps kept in the middle)
struct TransEntry { long src_off, dst_off; int A, T, Bd, Apad; };

__global__ __launch_bounds__(256) void k_transpose_cast_multi(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                              const TransEntry* __restrict__ table) {
    // 32x32 tiles through LDS: reads are contiguous along Bd, writes contiguous along A (pad columns stay zero)
    __shared__ float tile[32][33];
    const TransEntry e = table[blockIdx.y];
    const int ta = (e.A + 31) / 32, tb = (e.Bd + 31) / 32;
    const int ntiles = e.T * ta * tb;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {
        const int t = tile_id / (ta * tb);
        const int rem = tile_id - t * (ta * tb);
        const int a0 = (rem / tb) * 32, b0 = (rem % tb) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int a = a0 + ty + 8 * i, bd = b0 + tx;
            tile[ty + 8 * i][tx] = (a < e.A && bd < e.Bd) ? src[e.src_off + ((long)a * e.T + t) * e.Bd + bd] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int bd = b0 + ty + 8 * i, a = a0 + tx;
            if (a < e.A && bd < e.Bd) dst[e.dst_off + ((long)bd * e.T + t) * e.Apad + a] = f2bf(tile[tx][ty + 8 * i]);
        }
        __syncthreads();
    }
}

// the same from the bf16 shadow (same layout as the fp32 buffer, refreshed by the optimiser kernel just before): half the bytes read,
// identical result.  64 (A) x 64 (Bd) tiles: a thread reads two neighbouring Bd elements as one dword and writes two neighbouring A
// elements as one dword.
__global__ __launch_bounds__(256) void k_transpose_bf16_multi(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                              const TransEntry* __restrict__ table) {
    __shared__ bf16_t tile[64][66];
    const TransEntry e = table[blockIdx.y];
    const int ta = (e.A + 63) / 64, tb = (e.Bd + 63) / 64;
    const int ntiles = e.T * ta * tb;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 pairs x 8 rows
    if ((e.Bd & 7) == 0 && (e.Apad & 7) == 0 && (e.src_off & 7) == 0 && (e.dst_off & 7) == 0) {
        // 16-byte global accesses (every trunk / encoder weight): a thread loads two 8-element row pieces, the tile turns in LDS
        // (4-byte writes, 2-byte column reads), and it stores two 8-element pieces of transposed rows.  4-byte accesses ran the
        // refresh at 2.4 TB/s.
        const int vx = threadIdx.x & 7, vy = threadIdx.x >> 3;    // 8 pieces x 32 rows
        for (int tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {
            const int t = tile_id / (ta * tb);
            const int rem = tile_id - t * (ta * tb);
            const int a0 = (rem / tb) * 64, b0 = (rem % tb) * 64;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int a = a0 + vy + 32 * i, bd = b0 + 8 * vx;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (a < e.A && bd < e.Bd) v = *reinterpret_cast<const u32x4*>(src + e.src_off + ((long)a * e.T + t) * e.Bd + bd);
                unsigned* row = reinterpret_cast<unsigned*>(&tile[vy + 32 * i][8 * vx]);
                row[0] = v.x; row[1] = v.y; row[2] = v.z; row[3] = v.w;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int bd = b0 + vy + 32 * i, a = a0 + 8 * vx;
                if (bd < e.Bd && a < e.Apad) {
                    unsigned w[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        w[k] = (unsigned)tile[8 * vx + 2 * k][vy + 32 * i] | ((unsigned)tile[8 * vx + 2 * k + 1][vy + 32 * i] << 16);
                    u32x4 o; o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
                    *reinterpret_cast<u32x4*>(dst + e.dst_off + ((long)bd * e.T + t) * e.Apad + a) = o;
                }
            }
            __syncthreads();
        }
        return;
    }
    const bool pair_src = (e.Bd & 1) == 0 && (e.src_off & 1) == 0, pair_dst = (e.Apad & 1) == 0 && (e.dst_off & 1) == 0;
    for (int tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {
        const int t = tile_id / (ta * tb);
        const int rem = tile_id - t * (ta * tb);
        const int a0 = (rem / tb) * 64, b0 = (rem % tb) * 64;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int a = a0 + ty + 8 * i, bd = b0 + 2 * tx;
            bf16_t v0 = 0, v1 = 0;
            if (a < e.A) {
                const bf16_t* p = src + e.src_off + ((long)a * e.T + t) * e.Bd + bd;
                if (pair_src && bd + 1 < e.Bd) { const unsigned u = *reinterpret_cast<const unsigned*>(p); v0 = (bf16_t)(u & 0xffffu); v1 = (bf16_t)(u >> 16); }
                else { if (bd < e.Bd) v0 = p[0]; if (bd + 1 < e.Bd) v1 = p[1]; }
            }
            tile[ty + 8 * i][2 * tx] = v0;
            tile[ty + 8 * i][2 * tx + 1] = v1;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int bd = b0 + ty + 8 * i, a = a0 + 2 * tx;
            if (bd >= e.Bd || a >= e.A) continue;
            bf16_t* q = dst + e.dst_off + ((long)bd * e.T + t) * e.Apad + a;
            const bf16_t v0 = tile[2 * tx][ty + 8 * i], v1 = tile[2 * tx + 1][ty + 8 * i];
            if (pair_dst && a + 1 < e.A) *reinterpret_cast<unsigned*>(q) = (unsigned)v0 | ((unsigned)v1 << 16);
            else { q[0] = v0; if (a + 1 < e.A) q[1] = v1; }
        }
        __syncthreads();
    }
}

__global__ void k_fill_f32(float* p, long n, float v) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = v;
}

__global__ void k_word_add(int* w, int delta) { if (threadIdx.x == 0 && blockIdx.x == 0) w[0] += delta; }
__global__ void k_lincomb2(const float* a, const float* b, float wb, float* out) {
    // two roundings, as torch's `a + b * w` (a mul kernel, then an add kernel): no fused multiply-add here
    // (the build's -ffp-contract=fast disregards contraction pragmas: the product goes through an opaque register instead)
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float t = wb * b[0];
        asm volatile("" : "+v"(t));
        out[0] = a[0] + t;
    }
}

// out0 = (wa * a + wb * b) + wc * c with torch's roundings (three mul kernels, two add kernels); out1 = n / d (the sentence-level model's
// loss and token accuracy: e2e_asr_transformer.py:217-224)
__global__ void k_lincomb3_ratio(const float* a, float wa, const float* b, float wb, const float* c, float wc, float* out0, const float* n, const float* d, float* out1) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float t1 = wa * a[0], t2 = wb * b[0], t3 = wc * c[0];
        asm volatile("" : "+v"(t1), "+v"(t2), "+v"(t3));
        float s = t1 + t2;
        asm volatile("" : "+v"(s));
        out0[0] = s + t3;
        if (out1 != nullptr) out1[0] = n[0] / d[0];
    }
}

__global__ __launch_bounds__(256) void k_video_cast(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = f2bf(src[i]);
}

// ---------------------------------------------------------------------------------------------------------
// Device-side input pipeline: stored uint8 mouth crops -> model input.  Per clip: crop window (top, left, h, w) resized to
// H x W with bilinear sampling (align_corners = False, no antialias), horizontal flip, x / 255, (x - mean) / std — the chain
// x/255 -> RandomHorizontalFlip -> RandomResizedCrop | CenterCrop -> Normalize of reference LRW/video/src/data.py:150,157-171
// in one pass; the random decisions are made on the host and arrive as params [B][5] = {top, left, h, w, flip}.
// A window of exactly H x W is a plain crop (the sampling weights degenerate to 1 / 0).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_clip_prep(const unsigned char* __restrict__ src, const int* __restrict__ params,
                                                   float* __restrict__ dst, int B, int T, int Hs, int Ws, int H, int W, float mean,
                                                   float inv_std) {
    const long total = (long)B * T * H * W;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int x = (int)(idx % W);
        long r = idx / W;
        const int y = (int)(r % H);
        r /= H;
        const int t = (int)(r % T), b = (int)(r / T);
        int top = params[b * 5 + 0], left = params[b * 5 + 1], ch = params[b * 5 + 2], cw = params[b * 5 + 3];
        const int flip = params[b * 5 + 4];
        // the window is clamped into the stored frame: no read outside it whatever the host drew
        top = top < 0 ? 0 : (top > Hs - 1 ? Hs - 1 : top);
        left = left < 0 ? 0 : (left > Ws - 1 ? Ws - 1 : left);
        ch = ch < 1 ? 1 : (ch > Hs - top ? Hs - top : ch);
        cw = cw < 1 ? 1 : (cw > Ws - left ? Ws - left : cw);
        const int xo = flip ? W - 1 - x : x;
        // source coordinates inside the window (pixel centres), clamped to the window like torch's upsample_bilinear2d
        float fy = ((float)y + 0.5f) * ((float)ch / (float)H) - 0.5f;
        float fx = ((float)xo + 0.5f) * ((float)cw / (float)W) - 0.5f;
        fy = fmaxf(fy, 0.f); fx = fmaxf(fx, 0.f);
        int y0 = (int)fy, x0 = (int)fx;
        if (y0 > ch - 1) y0 = ch - 1;
        if (x0 > cw - 1) x0 = cw - 1;
        const int y1 = y0 + 1 < ch ? y0 + 1 : ch - 1, x1 = x0 + 1 < cw ? x0 + 1 : cw - 1;
        const float wy = fy - (float)y0, wx = fx - (float)x0;
        const unsigned char* f = src + ((long)b * T + t) * Hs * Ws;
        const float v00 = f[(top + y0) * Ws + left + x0], v01 = f[(top + y0) * Ws + left + x1];
        const float v10 = f[(top + y1) * Ws + left + x0], v11 = f[(top + y1) * Ws + left + x1];
        const float v = (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
        dst[idx] = (v * (1.0f / 255.0f) - mean) * inv_std;
    }
}

static inline int grid_for(long n) { long b = (n + 255) / 256; if (b > 4096) b = 4096; if (b < 1) b = 1; return (int)b; }

extern "C" {

int svsr_colsum_rows(const float* ws, int nrows, int64_t ld, float* out0, int64_t n0, float* out1, int64_t n1, int accumulate, float scale,
                     hipStream_t stream);

int svsr_ce_fwd(const void* logits, int logits_f32, int ld, const int64_t* target_idx, const float* target_prob, int ldt,
                int R, int V, float smoothing, float* loss, float* lse, float* row_loss, hipStream_t stream) {
    if (V > 64 * CE_MAXPER || V < 1 || R < 1 || (target_idx == nullptr) == (target_prob == nullptr) || row_loss == nullptr) return SVSR_ERR_ARG;
    CeArgs a{logits, logits_f32, ld, (const long*)target_idx, target_prob, ldt, R, V, smoothing, row_loss, lse};
    int grid = (R + 3) / 4; if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_ce_fwd, dim3(grid), dim3(256), 0, stream, a);
    const int rc = svsr_check_launch();
    if (rc != SVSR_OK) return rc;
    return svsr_colsum_rows(row_loss, R, 1, loss, 1, nullptr, 0, 0, 1.0f / (float)R, stream);     // *loss = mean of the row losses
}

int svsr_ce_bwd(const void* logits, int logits_f32, int ld, const int64_t* target_idx, const float* target_prob, int ldt,
                int R, int V, float smoothing, const float* lse, const float* gout, void* dlogits, int ldo, hipStream_t stream) {
    if (V < 1 || (target_idx == nullptr) == (target_prob == nullptr)) return SVSR_ERR_ARG;
    CeArgs a{logits, logits_f32, ld, (const long*)target_idx, target_prob, ldt, R, V, smoothing, nullptr, const_cast<float*>(lse)};
    int grid = (R + 3) / 4; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_ce_bwd, dim3(grid), dim3(256), 0, stream, a, gout, (bf16_t*)dlogits, ldo);
    return svsr_check_launch();
}

int svsr_topk_acc(const float* logits, const int64_t* labels, const float* soft_labels, int B, int C, float* out2, float* rows2,
                  hipStream_t stream) {
    if ((labels == nullptr) == (soft_labels == nullptr) || rows2 == nullptr || B < 1) return SVSR_ERR_ARG;      // rows2: [B][2] floats
    int grid = (B + 3) / 4; if (grid > 256) grid = 256;
    hipLaunchKernelGGL(k_topk_acc, dim3(grid), dim3(256), 0, stream, logits, (const long*)labels, soft_labels, B, C, rows2);
    const int rc = svsr_check_launch();
    if (rc != SVSR_OK) return rc;
    return svsr_colsum_rows(rows2, B, 2, out2, 2, nullptr, 0, 0, 1.0f / (float)B, stream);
}

int svsr_grad_sumsq(const float* g, int64_t n, void* opt_state, hipStream_t stream) {
    if (((uintptr_t)g & 15) != 0) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_grad_sumsq, dim3(OPT_PARTS), dim3(256), 0, stream, g, (long)n, (OptState*)opt_state, 0);
    return svsr_check_launch();
}

/* the sum of squares of one RANGE of the gradient into partial sums [part0, part0 + nparts) of the 1024: the ranges of a step together
 * must cover the buffer once and the partial sums once.  engine.TrainStep sums everything behind the stem convolution's weight into
 * partials 0..1022 on the side stream while the stem's weight gradient — the last one of the backward — is still being computed, and
 * that weight into partial 1023 afterwards (the whole-buffer pass sat alone between the backward and AdamW: 23 us LRW, 164 us LRS). */
int svsr_grad_sumsq_parts(const float* g, int64_t n, void* opt_state, int part0, int nparts, hipStream_t stream) {
    if (((uintptr_t)g & 15) != 0 || part0 < 0 || nparts < 1 || part0 + nparts > OPT_PARTS || n < 0) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_grad_sumsq, dim3(nparts), dim3(256), 0, stream, g, (long)n, (OptState*)opt_state, part0);
    return svsr_check_launch();
}

/* one RANGE of the flat buffers (pointers already offset; decay_end relative to the range) with the step counter advanced only when
 * `advance` != 0: a step may update the range the next forward needs first on the main stream and the rest on another stream beside that
 * forward (engine.TrainStep), the LAST range launched — behind every other range — advancing the counter. */
int svsr_adamw_range(float* p, const float* g, float* m, float* v, void* shadow, int64_t n, int64_t decay_end, float lr,
                     float beta1, float beta2, float eps, float weight_decay, float max_norm, int warmup, int total_steps,
                     void* opt_state, int advance, hipStream_t stream) {
    if (n > 0) {
        AdamArgs a{p, g, m, v, (bf16_t*)shadow, (long)n, (long)decay_end, lr, beta1, beta2, eps, weight_decay, max_norm, warmup,
                   total_steps, (OptState*)opt_state};
        hipLaunchKernelGGL(k_adamw, dim3(grid_for(n)), dim3(256), 0, stream, a);
    }
    if (advance) hipLaunchKernelGGL(k_opt_advance, dim3(1), dim3(256), 0, stream, (OptState*)opt_state, lr, warmup, total_steps);
    return svsr_check_launch();
}

int svsr_adamw_step(float* p, const float* g, float* m, float* v, void* shadow, int64_t n, int64_t decay_end, float lr,
                    float beta1, float beta2, float eps, float weight_decay, float max_norm, int warmup, int total_steps,
                    void* opt_state, hipStream_t stream) {
    return svsr_adamw_range(p, g, m, v, shadow, n, decay_end, lr, beta1, beta2, eps, weight_decay, max_norm, warmup, total_steps, opt_state, 1, stream);
}

int svsr_cast_bf16(const float* src, void* dst, int64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(k_cast_bf16, dim3(grid_for(n)), dim3(256), 0, stream, src, (bf16_t*)dst, (long)n);
    return svsr_check_launch();
}

// table: device array of n_entries {int64 src_off, int64 dst_off, int32 A, T, Bd, Apad}
int svsr_transpose_cast_multi(const float* src, void* dst, const void* table, int n_entries, hipStream_t stream) {
    if (n_entries < 1) return SVSR_OK;
    hipLaunchKernelGGL(k_transpose_cast_multi, dim3(128, n_entries), dim3(256), 0, stream, src, (bf16_t*)dst, (const TransEntry*)table);
    return svsr_check_launch();
}

/* the same refresh from the bf16 shadow of the parameters (same offsets as the fp32 buffer): half the bytes read, identical result */
int svsr_transpose_bf16_multi(const void* src16, void* dst, const void* table, int n_entries, hipStream_t stream) {
    if (n_entries < 1) return SVSR_OK;
    hipLaunchKernelGGL(k_transpose_bf16_multi, dim3(128, n_entries), dim3(256), 0, stream, (const bf16_t*)src16, (bf16_t*)dst, (const TransEntry*)table);
    return svsr_check_launch();
}

/* uint8 clips [B][T][Hs][Ws] -> fp32 model input [B][1][T][H][W]; params: device int32 [B][5] = {top, left, h, w, flip}, every
 * window inside the stored frame. */
int svsr_clip_prep(const void* src_u8, const int* params, float* dst, int B, int T, int Hs, int Ws, int H, int W, float mean, float std,
                   hipStream_t stream) {
    if (B < 1 || T < 1 || Hs < 1 || Ws < 1 || H < 1 || W < 1 || !(std > 0.f)) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_clip_prep, dim3(grid_for((long)B * T * H * W)), dim3(256), 0, stream, (const unsigned char*)src_u8, params, dst,
                       B, T, Hs, Ws, H, W, mean, 1.0f / std);
    return svsr_check_launch();
}

int svsr_fill_f32(float* p, int64_t n, float v, hipStream_t stream) {
    hipLaunchKernelGGL(k_fill_f32, dim3(grid_for(n)), dim3(256), 0, stream, p, (long)n, v);
    return svsr_check_launch();
}

int svsr_word_add(int* word, int delta, hipStream_t stream) {
    if (word == nullptr) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_word_add, dim3(1), dim3(64), 0, stream, word, delta);
    return svsr_check_launch();
}

int svsr_lincomb3_ratio(const float* a, float wa, const float* b, float wb, const float* c, float wc, float* out0, const float* num, const float* den,
                        float* out1, hipStream_t stream) {
    if (a == nullptr || b == nullptr || c == nullptr || out0 == nullptr || (out1 != nullptr && (num == nullptr || den == nullptr))) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_lincomb3_ratio, dim3(1), dim3(64), 0, stream, a, wa, b, wb, c, wc, out0, num, den, out1);
    return svsr_check_launch();
}

int svsr_lincomb2(const float* a, const float* b, float wb, float* out, hipStream_t stream) {
    if (a == nullptr || b == nullptr || out == nullptr) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_lincomb2, dim3(1), dim3(64), 0, stream, a, b, wb, out);
    return svsr_check_launch();
}

}  // extern "C"
