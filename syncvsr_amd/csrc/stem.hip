// 3-D stem convolution Conv3d(1, 64, (5,7,7), stride (1,2,2), pad (2,3,3), bias=False) on MFMA (gfx950).
// Replaces the ATen/MIOpen conv3d reached by `stem3d[0]` (reference LRW/video/src/lightning.py:50; LRS twin
// conv3d_extractor.py:32-34) and its weight gradient.  C_in = 1, so the contraction is a
// 245-tap stencil: the taps are laid out as K = 36 rows (kt,kh; 35 real) x 8 columns (kw; 7 real) = 288 and the
// im2col operand is never materialised — A fragments are read straight out of an LDS copy of the input rows
// (5 frames x a few rows, bf16), where 8 consecutive kw of one (kt,kh) row are 8 consecutive pixels.
#include "common.h"

#define STEM_C 64
#define STEM_KROWS 36          // 35 (kt,kh) rows + 1 zero row
#define STEM_WPITCH 296        // 288 + 8 pad: 592-byte rows -> conflict-free ds_read_b128 of B fragments

typedef __attribute__((ext_vector_type(4))) unsigned v4u;
typedef __attribute__((ext_vector_type(2))) unsigned v2u;

struct StemFwdArgs {
    const float* vid;   // [B][T][H][W] fp32 (C = 1)
    const float* w;     // [64][5][7][7] fp32
    bf16_t* out;        // [B*T][Ho][Wo][64] bf16
    float* stats;       // optional BatchNorm partials [gridDim.x][2][64]: one row per (persistent) workgroup
    int B, T, H, W, Ho, Wo;
    int tiles_per_frame, total_tiles;
    int rows_in_max, WP;   // LDS input tile: [5][rows_in_max][WP]
};

__global__ __launch_bounds__(256) void k_stem_conv_fwd(const StemFwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sW = reinterpret_cast<bf16_t*>(smem_raw);                 // [64][STEM_WPITCH]
    bf16_t* sIn = sW + STEM_C * STEM_WPITCH;                          // [5][rows_in_max][WP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HoWo = p.Ho * p.Wo;

    // weights -> LDS, padded K layout k = (kt*7+kh)*8 + kw
    for (int e = tid; e < STEM_C * STEM_KROWS * 8; e += 256) {
        const int c = e / (STEM_KROWS * 8), rem = e - c * (STEM_KROWS * 8);
        const int r = rem >> 3, kw = rem & 7;
        float v = 0.f;
        if (r < 35 && kw < 7) v = p.w[c * 245 + r * 7 + kw];
        sW[c * STEM_WPITCH + r * 8 + kw] = f2bf(v);
    }

    float st_s[2] = {0.f, 0.f}, st_q[2] = {0.f, 0.f};   // this lane's channel (jt*32 + lane&31), over the rows this lane holds

    for (int tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
        const int f = tile / p.tiles_per_frame, tp = tile - f * p.tiles_per_frame;
        const int b = f / p.T, t = f - b * p.T;
        const int p0 = tp * 128;
        int plast = p0 + 127;
        if (plast > HoWo - 1) plast = HoWo - 1;
        const int y0 = p0 / p.Wo, y1 = plast / p.Wo;
        const int nrows = 2 * (y1 - y0) + 7;
        const int row_base = 2 * y0 - 3;

        __syncthreads();   // previous tile's reads of sIn are complete (also orders the weight fill)
        // fill the input tile: one 32-lane group per (kt, row)
        for (int rr = tid >> 5; rr < 5 * nrows; rr += 8) {
            const int kt = rr / nrows, r = rr - kt * nrows;
            const int tt = t + kt - 2, iy = row_base + r;
            const bool row_ok = tt >= 0 && tt < p.T && iy >= 0 && iy < p.H;
            const float* src = p.vid + (((long)b * p.T + tt) * p.H + iy) * p.W;
            bf16_t* dst = sIn + ((long)kt * p.rows_in_max + r) * p.WP;
            if ((p.W & 3) == 0) {
                // 16-byte global loads; column ix lands at dst[ix + 3]; the 3 + 3 pad columns are zeroed by lanes 0..5
                const int l32 = tid & 31;
                if (l32 < 6) dst[l32 < 3 ? l32 : p.W + l32] = 0;
                for (int v4 = l32; v4 < (p.W >> 2); v4 += 32) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row_ok) v = *reinterpret_cast<const float4*>(src + 4 * v4);
                    bf16_t* d = dst + 4 * v4 + 3;
                    d[0] = f2bf(v.x); d[1] = f2bf(v.y); d[2] = f2bf(v.z); d[3] = f2bf(v.w);
                }
            } else {
                for (int c = tid & 31; c < p.WP; c += 32) {
                    const int ix = c - 3;
                    float v = 0.f;
                    if (row_ok && ix >= 0 && ix < p.W) v = src[ix];
                    dst[c] = f2bf(v);
                }
            }
        }
        __syncthreads();

        int pp = p0 + wave * 32 + (lane & 31);
        if (pp > HoWo - 1) pp = HoWo - 1;          // clamp: rows beyond the frame are masked at the store
        const int y = pp / p.Wo, x = pp - y * p.Wo;
        const int abase = (2 * (y - y0)) * p.WP + 2 * x;
        const int kg = lane >> 5;

        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }

#pragma unroll
        for (int ks = 0; ks < STEM_KROWS / 2; ++ks) {
            const int re = 2 * ks, ro = (2 * ks + 1 < 35) ? 2 * ks + 1 : 34;   // the zero row reads row 34's (finite) pixels
            const int kte = re / 7, khe = re % 7, kto = ro / 7, kho = ro % 7;
            const int kt = kg ? kto : kte, kh = kg ? kho : khe;
            const unsigned* src = reinterpret_cast<const unsigned*>(sIn + (kt * p.rows_in_max + kh) * p.WP + abase);
            union { bf16x8 v; unsigned u[4]; } fa;
            fa.u[0] = src[0]; fa.u[1] = src[1]; fa.u[2] = src[2]; fa.u[3] = src[3];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const bf16x8 fb = *reinterpret_cast<const bf16x8*>(sW + (jt * 32 + (lane & 31)) * STEM_WPITCH + ks * 16 + kg * 8);
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, fb, acc[jt], 0, 0, 0);
            }
        }

        bf16_t* obase = p.out + (long)f * HoWo * STEM_C;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int c = jt * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int po = p0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (po < HoWo) {
                    const float v = acc[jt][r];
                    obase[(long)po * STEM_C + c] = f2bf(v);
                    st_s[jt] += v;
                    st_q[jt] += v * v;
                }
            }
        }
    }
    if (p.stats != nullptr) {
        // the four waves hold disjoint positions of the same 64 channels: add them in wave order through LDS, then one plain
        // store per channel and statistic into this workgroup's row (svsr_bn_finalize adds the rows in a fixed order)
        __syncthreads();
        float* sred = reinterpret_cast<float*>(smem_raw);          // [4 waves][2][64]
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const float s = st_s[jt] + __shfl_xor(st_s[jt], 32, 64);
            const float q = st_q[jt] + __shfl_xor(st_q[jt], 32, 64);
            if (lane < 32) {
                sred[(wave * 2 + 0) * STEM_C + jt * 32 + lane] = s;
                sred[(wave * 2 + 1) * STEM_C + jt * 32 + lane] = q;
            }
        }
        __syncthreads();
        if (tid < 2 * STEM_C) {
            const int which = tid >> 6, c = tid & 63;
            const float v = ((sred[(0 * 2 + which) * STEM_C + c] + sred[(1 * 2 + which) * STEM_C + c]) + sred[(2 * 2 + which) * STEM_C + c]) +
                            sred[(3 * 2 + which) * STEM_C + c];
            p.stats[((long)blockIdx.x * 2 + which) * STEM_C + c] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// The same convolution fed by LDS-DMA (the path 88 x 88 clips take).  k_stem_conv_fwd above converts fp32 pixels and weights to bf16 on
// its way into LDS with one dependent global round trip per row (9 per tile) and per weight element (72 per workgroup): measured,
// 106 us of its 233 us were the input fill and 60 us the weight fill, for 70 us of MFMA work.  Here a prep pass (k_stem_prep, ~12 us)
// writes the clip once as bf16 rows in exactly the LDS layout — [3 zeros | W pixels | 5 zeros], pitch W + 8 — and the weights in the
// padded K layout [64][296]; tile rows and weights then go global -> LDS as verbatim 16-byte DMA copies.
// ------------------------------------------------------------------------------------------------------------
#define STEM_W_CHUNKS (STEM_C * STEM_WPITCH / 8)               // 2368 16-byte pieces
static_assert(STEM_W_CHUNKS % 64 == 0, "the weight DMA covers whole waves");

__device__ unsigned g_stem_zero_row[64];     // 256 zero bytes: DMA source of rows outside the clip

__global__ __launch_bounds__(256) void k_stem_prep(const float* __restrict__ vid, const float* __restrict__ w, bf16_t* __restrict__ vid16,
                                                   bf16_t* __restrict__ wpack, long rows, int W, int vid_blocks) {
    if ((int)blockIdx.x >= vid_blocks) {        // weights: k = (kt*7+kh)*8 + kw, zero where kw = 7 or the row is the 36th
        const int e = (blockIdx.x - vid_blocks) * 256 + threadIdx.x;
        if (e < STEM_C * STEM_WPITCH) {
            const int c = e / STEM_WPITCH, k = e - c * STEM_WPITCH, r = k >> 3, kw = k & 7;
            wpack[e] = f2bf((r < 35 && kw < 7) ? w[c * 245 + r * 7 + kw] : 0.f);
        }
        return;
    }
    const int cpr = (W + 8) >> 3;               // 16-byte pieces per output row
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * cpr) return;
    const long row = e / cpr;
    const int ch = (int)(e - row * cpr);
    const float* src = vid + row * W;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int ix = ch * 8 + k - 3;
        v[k] = (ix >= 0 && ix < W) ? src[ix] : 0.f;
    }
    *reinterpret_cast<u32x4*>(vid16 + row * (W + 8) + ch * 8) = pack8(v);
}

__global__ __launch_bounds__(256) void k_stem_conv_fwd_dma(const StemFwdArgs p, const bf16_t* __restrict__ vid16, const bf16_t* __restrict__ wpack) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sW = reinterpret_cast<bf16_t*>(smem_raw);                 // [64][STEM_WPITCH] (+ pad to whole DMA waves)
    bf16_t* sIn = sW + STEM_W_CHUNKS * 8;                             // [5 * nrows][WP], rows packed with this tile's own row count (+ pad to a whole wave)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HoWo = p.Ho * p.Wo;
    const int WP = p.W + 8, cpr = WP >> 3;
    const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(g_stem_zero_row) + (lane & 7) * 8;

    for (int e0 = 0; e0 + wave * 64 < STEM_W_CHUNKS; e0 += 256) {       // (wave-uniform bound: a DMA instruction moves 64 pieces)
        const int e = e0 + tid;
        const bf16_t* src = wpack + (long)e * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sW + (size_t)(e0 + wave * 64) * 8), 16, 0, 0);
    }

    float st_s[2] = {0.f, 0.f}, st_q[2] = {0.f, 0.f};   // this lane's channel (jt*32 + lane&31), over the rows this lane holds

    for (int tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
        const int f = tile / p.tiles_per_frame, tp = tile - f * p.tiles_per_frame;
        const int b = f / p.T, t = f - b * p.T;
        const int p0 = tp * 128;
        int plast = p0 + 127;
        if (plast > HoWo - 1) plast = HoWo - 1;
        const int y0 = p0 / p.Wo, y1 = plast / p.Wo;
        const int nrows = 2 * (y1 - y0) + 7;
        const int row_base = 2 * y0 - 3;

        __syncthreads();   // previous tile's reads of sIn are complete
        const int chunks = 5 * nrows * cpr;
        for (int e0 = 0; e0 + wave * 64 < chunks; e0 += 256) {
            const int e = e0 + tid;
            const int rr = e / cpr, ch = e - rr * cpr;
            const int kt = rr / nrows, r = rr - kt * nrows;
            const int tt = t + kt - 2, iy = row_base + r;
            const bool ok = e < chunks && tt >= 0 && tt < p.T && iy >= 0 && iy < p.H;
            const bf16_t* src = ok ? vid16 + (((long)b * p.T + tt) * p.H + iy) * WP + ch * 8 : zero_src;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sIn + (size_t)(e0 + wave * 64) * 8), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (also the weights, the first time)
        __syncthreads();

        int pp = p0 + wave * 32 + (lane & 31);
        if (pp > HoWo - 1) pp = HoWo - 1;          // clamp: rows beyond the frame are masked at the store
        const int y = pp / p.Wo, x = pp - y * p.Wo;
        const int abase = (2 * (y - y0)) * WP + 2 * x;
        const int kg = lane >> 5;

        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }

#pragma unroll
        for (int ks = 0; ks < STEM_KROWS / 2; ++ks) {
            const int re = 2 * ks, ro = (2 * ks + 1 < 35) ? 2 * ks + 1 : 34;   // the zero row reads row 34's (finite) pixels
            const int kte = re / 7, khe = re % 7, kto = ro / 7, kho = ro % 7;
            const int kt = kg ? kto : kte, kh = kg ? kho : khe;
            const unsigned* src = reinterpret_cast<const unsigned*>(sIn + (kt * nrows + kh) * WP + abase);
            v4u fa4;
            fa4[0] = src[0]; fa4[1] = src[1]; fa4[2] = src[2]; fa4[3] = src[3];
            const bf16x8 fa = __builtin_bit_cast(bf16x8, fa4);
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const bf16x8 fb = *reinterpret_cast<const bf16x8*>(sW + (jt * 32 + (lane & 31)) * STEM_WPITCH + ks * 16 + kg * 8);
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[jt], 0, 0, 0);
            }
        }

        bf16_t* obase = p.out + (long)f * HoWo * STEM_C;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int c = jt * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int po = p0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (po < HoWo) {
                    const float v = acc[jt][r];
                    obase[(long)po * STEM_C + c] = f2bf(v);
                    st_s[jt] += v;
                    st_q[jt] += v * v;
                }
            }
        }
    }
    if (p.stats != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // a workgroup without a tile still has its weight DMA in flight
        __syncthreads();
        float* sred = reinterpret_cast<float*>(smem_raw);          // [4 waves][2][64]
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const float s = st_s[jt] + __shfl_xor(st_s[jt], 32, 64);
            const float q = st_q[jt] + __shfl_xor(st_q[jt], 32, 64);
            if (lane < 32) {
                sred[(wave * 2 + 0) * STEM_C + jt * 32 + lane] = s;
                sred[(wave * 2 + 1) * STEM_C + jt * 32 + lane] = q;
            }
        }
        __syncthreads();
        if (tid < 2 * STEM_C) {
            const int which = tid >> 6, c = tid & 63;
            const float v = ((sred[(0 * 2 + which) * STEM_C + c] + sred[(1 * 2 + which) * STEM_C + c]) + sred[(2 * 2 + which) * STEM_C + c]) +
                            sred[(3 * 2 + which) * STEM_C + c];
            p.stats[((long)blockIdx.x * 2 + which) * STEM_C + c] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// weight gradient  dW[c][kt][kh][kw] += sum_{b,t,y,x} dY[b,t,y,x,c] * in[b, t+kt-2, 2y+kh-3, 2x+kw-3]
// MFMA view: D[c][k] += sum_m dY^T[c][m] * P[m][k] with m running along x inside one output row.  A fragments
// (8 consecutive positions of one channel) come from the position-major dY tile via ds_read_b64_tr_b16; B fragments
// (8 consecutive x for one tap) are stride-2 pixels, made contiguous by storing the input rows de-interleaved into
// even/odd column planes.
// ------------------------------------------------------------------------------------------------------------
#define SW_RB 4            // output rows per tile
#define SW_DPITCH 96       // dY tile row pitch (bf16): 192 B = 48 banks -> conflict-free transpose reads (see wgrad3x3.hip)

struct StemWgradArgs {
    const float* vid;
    const bf16_t* dy;   // [B*T][Ho][Wo][64]
    float* dw;          // [64][245] fp32, accumulated
    float* part;        // slabs [gridDim.x][64*245]: one per (persistent) workgroup, added into dw by svsr_colsum_rows
    int B, T, H, W, Ho, Wo;
    int WoP;            // Wo rounded up to 16
    int PW;             // plane row width (elements, even): WoP + 8
    int groups_per_frame, total_tiles;
    int use_tr;
};

__device__ unsigned g_stem_zero[4];     // 16 zero bytes: source of predicated-off tile loads

template <bool USE_TR>
__device__ __forceinline__ bf16x8 stem_frag_T(const bf16_t* tile, int ch0, int pos0, int lane) {
    bf16x8 f;
    if (USE_TR) {
        const int gq = lane >> 4, s = lane & 15;
        const bf16_t* base = tile + (pos0 + (gq >> 1) * 8 + (s >> 2)) * SW_DPITCH + ch0 + (gq & 1) * 16 + (s & 3) * 4;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(base));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(base + 4 * SW_DPITCH));
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    } else {
        const bf16_t* base = tile + (pos0 + (lane >> 5) * 8) * SW_DPITCH + ch0 + (lane & 31);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = (short)base[k * SW_DPITCH];
    }
    return f;
}

template <bool USE_TR>
__global__ __launch_bounds__(256) void k_stem_conv_wgrad(const StemWgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sDY = reinterpret_cast<bf16_t*>(smem_raw);                 // [SW_RB*WoP][SW_DPITCH]
    bf16_t* sIn = sDY + SW_RB * p.WoP * SW_DPITCH;                     // [5][2*SW_RB+5][2][PW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nrows = 2 * SW_RB + 5;
    const int kg = lane >> 5;

    // this lane's two taps (B-fragment columns): k = (wave*2 + q)*32 + (lane&31)
    int kbase[2];
    bool kvalid[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int k = (wave * 2 + q) * 32 + (lane & 31);
        kvalid[q] = k < 245;
        const int kk = kvalid[q] ? k : 0;
        const int kt = kk / 49, kh = (kk % 49) / 7, kw = kk % 7;
        const int plane = (kw + 1) & 1, off = (kw + 1) >> 1;      // padded column 2x+kw+1 -> plane, index x + off
        kbase[q] = ((kt * nrows + kh) * 2 + plane) * p.PW + off + kg * 8;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
        const int f = tile / p.groups_per_frame, gi = tile - f * p.groups_per_frame;
        const int b = f / p.T, t = f - b * p.T;
        const int y0 = gi * SW_RB;
        int rb = p.Ho - y0;
        if (rb > SW_RB) rb = SW_RB;
        const int row_base = 2 * y0 - 3;

        __syncthreads();
        // dY tile: [rb][WoP] positions x 8 chunks of 16 B; pad positions are zero
        for (int e = tid; e < SW_RB * p.WoP * 8; e += 256) {
            const int pos = e >> 3, ch = e & 7;
            const int yl = pos / p.WoP, x = pos - yl * p.WoP;
            // (no branch around the load and a compiler vector type: a conditionally assigned u32x4 STRUCT went through scratch memory)
            const bool ok = yl < rb && x < p.Wo;
            const v4u v = *reinterpret_cast<const v4u*>(ok ? p.dy + (((long)f * p.Ho + y0 + yl) * p.Wo + x) * STEM_C + ch * 8
                                                           : reinterpret_cast<const bf16_t*>(g_stem_zero));
            *reinterpret_cast<v4u*>(sDY + pos * SW_DPITCH + ch * 8) = v;
        }
        // input rows, de-interleaved: padded column cp = col + 4 -> plane cp&1, index cp>>1
        for (int rr = tid >> 5; rr < 5 * nrows; rr += 8) {
            const int kt = rr / nrows, r = rr - kt * nrows;
            const int tt = t + kt - 2, iy = row_base + r;
            const bool row_ok = tt >= 0 && tt < p.T && iy >= 0 && iy < p.H;
            const float* src = p.vid + (((long)b * p.T + tt) * p.H + iy) * p.W;
            bf16_t* dst = sIn + (long)rr * 2 * p.PW;
            if ((p.W & 3) == 0) {
                // 16-byte global loads: pixels 4v..4v+3 -> even plane [2v+2, 2v+3] = (x0, x2), odd plane [2v+2, 2v+3] = (x1, x3)
                const int l32 = tid & 31;
                unsigned* even = reinterpret_cast<unsigned*>(dst);
                unsigned* odd = reinterpret_cast<unsigned*>(dst + p.PW);
                const int nd = p.PW >> 1, first = 1, last = (p.W >> 2) + 1;      // dwords per plane; data dwords [first, last)
                for (int d = l32; d < nd; d += 32) {
                    unsigned e = 0u, o = 0u;
                    if (row_ok && d >= first && d < last) {
                        const float4 v = *reinterpret_cast<const float4*>(src + 4 * (d - 1));
                        e = pack2bf(v.x, v.z);
                        o = pack2bf(v.y, v.w);
                    }
                    even[d] = e;
                    odd[d] = o;
                }
            } else {
                for (int cp = tid & 31; cp < 2 * p.PW; cp += 32) {
                    const int ix = cp - 4;
                    float v = 0.f;
                    if (row_ok && ix >= 0 && ix < p.W) v = src[ix];
                    dst[(cp & 1) * p.PW + (cp >> 1)] = f2bf(v);
                }
            }
        }
        __syncthreads();

        for (int yl = 0; yl < rb; ++yl) {
            for (int xs = 0; xs < p.WoP; xs += 16) {
                const int pos0 = yl * p.WoP + xs;
                bf16x8 fa[2], fb[2];
                fa[0] = stem_frag_T<USE_TR>(sDY, 0, pos0, lane);
                fa[1] = stem_frag_T<USE_TR>(sDY, 32, pos0, lane);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int a = kbase[q] + (2 * yl) * 2 * p.PW + xs;      // element address of x = xs + kg*8 for this tap
                    const unsigned* src = reinterpret_cast<const unsigned*>(sIn + (a & ~1));
                    const unsigned sh = (a & 1) * 16;
                    const unsigned d0 = src[0], d1 = src[1], d2 = src[2], d3 = src[3], d4 = src[4];
                    u32x4 fr;          // (a plain struct of four registers: a union with an array went through scratch memory)
                    fr.x = kvalid[q] ? __builtin_amdgcn_alignbit(d1, d0, sh) : 0u;
                    fr.y = kvalid[q] ? __builtin_amdgcn_alignbit(d2, d1, sh) : 0u;
                    fr.z = kvalid[q] ? __builtin_amdgcn_alignbit(d3, d2, sh) : 0u;
                    fr.w = kvalid[q] ? __builtin_amdgcn_alignbit(d4, d3, sh) : 0u;
                    fb[q] = __builtin_bit_cast(bf16x8, fr);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[q], acc[i][q], 0, 0, 0);
            }
        }
    }
    // D[row = channel][col = tap] -> this workgroup's slab (plain stores; the slabs are added in a fixed order afterwards)
    float* dst = p.part + (long)blockIdx.x * (STEM_C * 245);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int k = (wave * 2 + q) * 32 + (lane & 31);
        if (k >= 245) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                dst[c * 245 + k] = acc[i][q][r];
            }
    }
}

template <bool USE_TR>
__global__ __launch_bounds__(256) void k_stem_conv_wgrad_pipe(const StemWgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sDY = reinterpret_cast<bf16_t*>(smem_raw);                 // [SW_RB*WoP][SW_DPITCH]
    bf16_t* sIn = sDY + SW_RB * p.WoP * SW_DPITCH;                     // [5][2*SW_RB+5][2][PW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nrows = 2 * SW_RB + 5;
    const int kg = lane >> 5;

    // this lane's two taps (B-fragment columns): k = (wave*2 + q)*32 + (lane&31)
    int kbase[2];
    bool kvalid[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int k = (wave * 2 + q) * 32 + (lane & 31);
        kvalid[q] = k < 245;
        const int kk = kvalid[q] ? k : 0;
        const int kt = kk / 49, kh = (kk % 49) / 7, kw = kk % 7;
        const int plane = (kw + 1) & 1, off = (kw + 1) >> 1;      // padded column 2x+kw+1 -> plane, index x + off
        kbase[q] = ((kt * nrows + kh) * 2 + plane) * p.PW + off + kg * 8;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Tile operands travel global -> registers -> LDS, and the registers of the NEXT tile are requested before this tile's MFMA
    // block: the first version loaded and stored piece by piece (one global round trip per loop iteration, ~15 per tile), which
    // made a tile cost 13 us for 0.7 us of MFMA work.  Host guarantees: SW_RB*WoP*8 <= NDY*256, W % 4 == 0, PW / 2 <= 32.
    constexpr int NDY = 8, NIN = (5 * (2 * SW_RB + 5) + 7) / 8;
    v4u dyv[NDY];            // (compiler vector types: arrays of the u32x4 / float4 STRUCTS were kept in scratch memory)
    f32x4 inv[NIN];
    const int dy_total = SW_RB * p.WoP * 8;
    const int l32 = tid & 31, rr0 = tid >> 5;
    const int nd = p.PW >> 1, last = (p.W >> 2) + 1;       // dwords per plane; data dwords [1, last)
    // one loop iteration = { stash the registers of `tile` into LDS, request the registers of the next tile, contract `tile` }; the
    // first iteration (tile < 0) only requests
    for (int tile = (int)blockIdx.x - (int)gridDim.x;; tile += gridDim.x) {
        const int nxt = tile + (int)gridDim.x;
        int rb = 0;
        if (tile >= 0) {
            const int gi = tile % p.groups_per_frame;
            rb = p.Ho - gi * SW_RB;
            if (rb > SW_RB) rb = SW_RB;
            __syncthreads();                 // everybody is done reading the previous tile
#pragma unroll
            for (int i = 0; i < NDY; ++i) {
                const int e = tid + 256 * i, pos = e >> 3, ch = e & 7;
                if (e < dy_total) *reinterpret_cast<v4u*>(sDY + pos * SW_DPITCH + ch * 8) = dyv[i];
            }
            // input rows, de-interleaved: pixels 4v..4v+3 -> even plane dword (x0, x2), odd plane dword (x1, x3) at dword v + 1
#pragma unroll
            for (int i = 0; i < NIN; ++i) {
                const int rr = rr0 + 8 * i;
                if (rr < 5 * nrows && l32 < nd) {
                    bf16_t* dst = sIn + (long)rr * 2 * p.PW;
                    reinterpret_cast<unsigned*>(dst)[l32] = pack2bf(inv[i][0], inv[i][2]);
                    reinterpret_cast<unsigned*>(dst + p.PW)[l32] = pack2bf(inv[i][1], inv[i][3]);
                }
            }
            __syncthreads();
        }
        if (nxt < p.total_tiles) {
            const int f = nxt / p.groups_per_frame, gi = nxt - f * p.groups_per_frame;
            const int b = f / p.T, t = f - b * p.T;
            const int y0 = gi * SW_RB;
            int rbn = p.Ho - y0;
            if (rbn > SW_RB) rbn = SW_RB;
            const int row_base = 2 * y0 - 3;
#pragma unroll
            for (int i = 0; i < NDY; ++i) {
                const int e = tid + 256 * i, pos = e >> 3, ch = e & 7;
                const int yl = pos / p.WoP, x = pos - yl * p.WoP;
                const bool ok = e < dy_total && yl < rbn && x < p.Wo;       // (no branch: the register arrays must not end up in scratch)
                dyv[i] = *reinterpret_cast<const v4u*>(ok ? p.dy + (((long)f * p.Ho + y0 + yl) * p.Wo + x) * STEM_C + ch * 8
                                                            : reinterpret_cast<const bf16_t*>(g_stem_zero));
            }
#pragma unroll
            for (int i = 0; i < NIN; ++i) {
                const int rr = rr0 + 8 * i;
                const int kt = rr / nrows, r = rr - kt * nrows;
                const int tt = t + kt - 2, iy = row_base + r;
                const bool ok = rr < 5 * nrows && tt >= 0 && tt < p.T && iy >= 0 && iy < p.H && l32 >= 1 && l32 < last;
                inv[i] = *reinterpret_cast<const f32x4*>(ok ? p.vid + (((long)b * p.T + tt) * p.H + iy) * p.W + 4 * (l32 - 1)
                                                             : reinterpret_cast<const float*>(g_stem_zero));
            }
        }
        if (tile < 0) {
            if (nxt >= p.total_tiles) break;      // (a workgroup without any tile: the host never launches one)
            continue;
        }

        for (int yl = 0; yl < rb; ++yl) {
            for (int xs = 0; xs < p.WoP; xs += 16) {
                const int pos0 = yl * p.WoP + xs;
                bf16x8 fa[2], fb[2];
                fa[0] = stem_frag_T<USE_TR>(sDY, 0, pos0, lane);
                fa[1] = stem_frag_T<USE_TR>(sDY, 32, pos0, lane);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int a = kbase[q] + (2 * yl) * 2 * p.PW + xs;      // element address of x = xs + kg*8 for this tap
                    const unsigned* src = reinterpret_cast<const unsigned*>(sIn + (a & ~1));
                    const unsigned sh = (a & 1) * 16;
                    const unsigned d0 = src[0], d1 = src[1], d2 = src[2], d3 = src[3], d4 = src[4];
                    u32x4 fr;          // (a plain struct of four registers: a union with an array went through scratch memory)
                    fr.x = kvalid[q] ? __builtin_amdgcn_alignbit(d1, d0, sh) : 0u;
                    fr.y = kvalid[q] ? __builtin_amdgcn_alignbit(d2, d1, sh) : 0u;
                    fr.z = kvalid[q] ? __builtin_amdgcn_alignbit(d3, d2, sh) : 0u;
                    fr.w = kvalid[q] ? __builtin_amdgcn_alignbit(d4, d3, sh) : 0u;
                    fb[q] = __builtin_bit_cast(bf16x8, fr);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[q], acc[i][q], 0, 0, 0);
            }
        }
        if (nxt >= p.total_tiles) break;
    }
    // D[row = channel][col = tap] -> this workgroup's slab (plain stores; the slabs are added in a fixed order afterwards)
    float* dst = p.part + (long)blockIdx.x * (STEM_C * 245);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int k = (wave * 2 + q) * 32 + (lane & 31);
        if (k >= 245) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                dst[c * 245 + k] = acc[i][q][r];
            }
    }
}


// ------------------------------------------------------------------------------------------------------------
// The stem's BatchNorm + activation + max-pool backward APPLY pass and the weight gradient as ONE pass (round 5): the gradient of the
// convolution output was written (230 MB at 928 frames) by svsr_stem_bn_act_pool_bwd's apply launch and read back by k_stem_conv_wgrad_pipe
// with nothing else on either stream — the step's last ~265 us.  Here a tile's four gradient rows are made where the contraction wants
// them: the convolution output x takes the place of dY in the prefetch registers, the three pooled rows over the tile (g = dpool * act'
// at the window winners, written by k_stem_bwd_reduce_win, and the arg-max bytes) go through LDS, and every thread turns its eight
// 16-byte pieces of x into dY = coef0 * (sum of the windows this element won - coef1 - xhat * coef2) — norm_act.hip k_stem_bwd_lds<.,true,true>'s
// arithmetic, operation for operation (same window order, same bf16 rounding) — straight into the dY tile.  Tile walk, MFMA order and
// slabs are k_stem_conv_wgrad_pipe's, so dW is bit-identical to the two-launch form (tests/test_gpu_kernels.py).
// ------------------------------------------------------------------------------------------------------------
#define SBW_PR (SW_RB / 2 + 1)      // pooled rows whose windows reach a tile's SW_RB convolution rows (tiles start at even rows)
#define SBW_NG 3                    // 16-byte (g) + 8-byte (arg-max) pieces of the pooled rows per thread: SBW_PR * Wp * 8 <= SBW_NG * 256

struct StemBwdWgradArgs {
    const float* vid;
    const bf16_t* x;              // [B*T][Ho][Wo][64] convolution output (before BatchNorm)
    const bf16_t* gpool;          // [B*T][Hp][Wp][64] dpool * act'(winner)
    const unsigned char* amax;    // [B*T][Hp][Wp][64] window position of the winner, i * 3 + j
    const float* mean;
    const float* rstd;
    const float* coef;            // [3][64]: gamma * rstd, mean(g), mean(g * xhat)
    float* part;                  // slabs [gridDim.x][64*245]
    int B, T, H, W, Ho, Wo, Hp, Wp;
    int WoP, PW, groups_per_frame, total_tiles;
};

// NDY: 16-byte pieces of the x tile per thread, SW_RB * WoP * 8 <= NDY * 256; NIN: float4 pieces of the 65 input rows per thread,
// 65 * (W / 4) <= NIN * 256 (6 and 6 for 88-wide clips: prefetch registers that are never used still count against the 256)
template <int NDY, int NIN>
__global__ __launch_bounds__(256, 2) void k_stem_bwd_wgrad(const StemBwdWgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int nrows = 2 * SW_RB + 5;
    bf16_t* sDY = reinterpret_cast<bf16_t*>(smem_raw);                 // [SW_RB*WoP][SW_DPITCH]
    bf16_t* sIn = sDY + SW_RB * p.WoP * SW_DPITCH;                     // [5][2*SW_RB+5][2][PW]
    bf16_t* sG = sIn + 5 * nrows * 2 * p.PW;                           // [SBW_PR][Wp][64]
    unsigned char* sM = reinterpret_cast<unsigned char*>(sG + SBW_PR * p.Wp * STEM_C);      // [SBW_PR][Wp][64]
    float* sC = reinterpret_cast<float*>(sM + SBW_PR * p.Wp * STEM_C);                       // [5][64]: mean, rstd, coef[0..2]
    const int tid_k = threadIdx.x, lane = tid_k & 63, wave = tid_k >> 6;
    const int kg = lane >> 5;

    int kbase[2];
    bool kvalid[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int k = (wave * 2 + q) * 32 + (lane & 31);
        kvalid[q] = k < 245;
        const int kk = kvalid[q] ? k : 0;
        const int kt = kk / 49, kh = (kk % 49) / 7, kw = kk % 7;
        const int plane = (kw + 1) & 1, off = (kw + 1) >> 1;
        kbase[q] = ((kt * nrows + kh) * 2 + plane) * p.PW + off + kg * 8;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    v4u xv[NDY], gv[SBW_NG];
    v2u mv[SBW_NG];
    f32x4 inv[NIN];
    const int dy_total = SW_RB * p.WoP * 8, g_total = SBW_PR * p.Wp * 8;
    const int nq = p.W >> 2, in_total = 5 * nrows * nq;      // input rows as float4 pieces, flat over (row, piece)
    const float inv_nq = 1.f / (float)nq;
    // the zero borders of the de-interleaved rows (plane dword 0 and dwords past W / 4) are written once; tiles only rewrite the pixels
    for (int d = tid_k; d < 5 * nrows * p.PW; d += 256) reinterpret_cast<unsigned*>(sIn)[d] = 0u;
    // a thread keeps its channel group (256 % 8 == 0); the BatchNorm constants of its eight channels are re-read from LDS for every tile's
    // gradient pieces (held in registers across the contraction they cost 40 of a 256-register budget that two workgroups per CU allow)
    if (tid_k < STEM_C) {
        sC[0 * STEM_C + tid_k] = p.mean[tid_k]; sC[1 * STEM_C + tid_k] = p.rstd[tid_k];
        sC[2 * STEM_C + tid_k] = p.coef[tid_k]; sC[3 * STEM_C + tid_k] = p.coef[STEM_C + tid_k]; sC[4 * STEM_C + tid_k] = p.coef[2 * STEM_C + tid_k];
    }
    const float inv_wop = 1.f / (float)p.WoP;       // tile row of a position: (pos + 0.5) / WoP is never within 0.007 of an integer

    for (int tile = (int)blockIdx.x - (int)gridDim.x;; tile += gridDim.x) {
        const int nxt = tile + (int)gridDim.x;
        int rb = 0;
        // (opaque per tile: the compiler otherwise keeps ~100 registers of tile-invariant piece offsets and predicates across the loop,
        // in scratch memory at the 256 registers two workgroups per CU leave)
        int tid = tid_k;
        asm volatile("" : "+v"(tid));
        const int c0 = (tid & 7) * 8;
        if (tile >= 0) {
            const int gi = tile % p.groups_per_frame, y0 = gi * SW_RB, p_lo = y0 >> 1;
            rb = p.Ho - y0;
            if (rb > SW_RB) rb = SW_RB;
            __syncthreads();                 // everybody is done reading the previous tile
#pragma unroll
            for (int i = 0; i < NDY; ++i) {      // x for now (pad positions zero): every thread turns its own pieces into dY in place below
                const int e = tid + 256 * i;
                if (e < dy_total) *reinterpret_cast<v4u*>(sDY + (e >> 3) * SW_DPITCH + c0) = xv[i];
            }
#pragma unroll
            for (int i = 0; i < SBW_NG; ++i) {
                const int e = tid + 256 * i;
                if (e < g_total) {
                    *reinterpret_cast<v4u*>(sG + (e >> 3) * STEM_C + c0) = gv[i];
                    *reinterpret_cast<v2u*>(sM + (e >> 3) * STEM_C + c0) = mv[i];
                }
            }
#pragma unroll
            for (int i = 0; i < NIN; ++i) {
                const int it = tid + 256 * i, rr = (int)(((float)it + 0.5f) * inv_nq), l = it - rr * nq;
                if (it < in_total) {         // pixels 4l..4l+3 -> even plane dword l + 1 = (x0, x2), odd plane dword l + 1 = (x1, x3)
                    bf16_t* dst = sIn + (long)rr * 2 * p.PW;
                    reinterpret_cast<unsigned*>(dst)[l + 1] = pack2bf(inv[i][0], inv[i][2]);
                    reinterpret_cast<unsigned*>(dst + p.PW)[l + 1] = pack2bf(inv[i][1], inv[i][3]);
                }
            }
            __syncthreads();                 // the pooled rows are in LDS
            float mu[8], rs[8], k0[8], k1[8], k2[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                mu[k] = sC[0 * STEM_C + c0 + k]; rs[k] = sC[1 * STEM_C + c0 + k];
                k0[k] = sC[2 * STEM_C + c0 + k]; k1[k] = sC[3 * STEM_C + c0 + k]; k2[k] = sC[4 * STEM_C + c0 + k];
            }
            // A thread turns 2 x 2 blocks of positions (even row and column first) of its channel group into dY in place: the four pooled
            // pixels whose windows reach the block are read and unpacked once (per element that was up to four reads each), and which window
            // position an element holds in each is a compile-time constant: i * 3 + j with i = h - (2 ph - 1), j = w - (2 pw - 1).  Windows are
            // added in k_stem_bwd_lds's order (ph_lo, pw_lo), (ph_lo, pw_hi), (ph_hi, pw_lo), (ph_hi, pw_hi); one that lies past the frame never
            // matches (code 0xff, clamped address); a non-winner adds +0, which changes nothing (a sum here is never -0).
            const int nbw = (p.Wo + 1) >> 1;
#pragma unroll 1
            for (int e = tid; e < (SW_RB / 2) * nbw * 8; e += 256) {
                const int blk = e >> 3, rp = blk >= nbw ? 1 : 0, cp = blk - rp * nbw;        // (SW_RB / 2 == 2 row pairs)
                if (2 * rp >= rb) continue;
                const bool row1 = 2 * rp + 1 < rb, col1 = 2 * cp + 1 < p.Wo;
                const bool ph1 = p_lo + rp + 1 < p.Hp, pw1 = cp + 1 < p.Wp;
                const int soA = (rp * p.Wp + cp) * STEM_C + c0, soB = soA + (pw1 ? STEM_C : 0);
                const int soC = soA + (ph1 ? p.Wp * STEM_C : 0), soD = soC + (pw1 ? STEM_C : 0);
                const uint2 mA = *reinterpret_cast<const uint2*>(sM + soA), mB = *reinterpret_cast<const uint2*>(sM + soB);
                const uint2 mC = *reinterpret_cast<const uint2*>(sM + soC), mD = *reinterpret_cast<const uint2*>(sM + soD);
                float gA[8], gB[8], gC[8], gD[8];
                unpack8(*reinterpret_cast<const u32x4*>(sG + soA), gA);
                unpack8(*reinterpret_cast<const u32x4*>(sG + soB), gB);
                unpack8(*reinterpret_cast<const u32x4*>(sG + soC), gC);
                unpack8(*reinterpret_cast<const u32x4*>(sG + soD), gD);
                const unsigned offB = pw1 ? 0u : 0xffu, offC = ph1 ? 0u : 0xffu, offD = (ph1 && pw1) ? 0u : 0xffu;      // | 0xff: never a stored code
#define SBW_WIN(acc, m, g, code)                                                                                   \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                             \
        acc[k] += (((m.x >> (8 * k)) & 0xffu) == (code)) ? g[k] : 0.f;                                          \
        acc[k + 4] += (((m.y >> (8 * k)) & 0xffu) == (code)) ? g[k + 4] : 0.f;                                  \
    }
#define SBW_ELEM(dy_, dx_, BODY)                                                                                   \
    {                                                                                                              \
        bf16_t* cell = sDY + ((2 * rp + dy_) * p.WoP + 2 * cp + dx_) * SW_DPITCH + c0;                            \
        float xf[8], a[8], ov[8];                                                                                  \
        unpack8(*reinterpret_cast<const u32x4*>(cell), xf);                                                        \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) a[k] = 0.f;                                               \
        BODY                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                         \
            const float xh = (xf[k] - mu[k]) * rs[k];                                                              \
            ov[k] = k0[k] * (a[k] - k1[k] - xh * k2[k]);                                                           \
        }                                                                                                          \
        *reinterpret_cast<u32x4*>(cell) = pack8(ov);                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    }
                SBW_ELEM(0, 0, SBW_WIN(a, mA, gA, 4u))
                if (col1) SBW_ELEM(0, 1, SBW_WIN(a, mA, gA, 5u) SBW_WIN(a, mB, gB, 3u | offB))
                if (row1) {
                    SBW_ELEM(1, 0, SBW_WIN(a, mA, gA, 7u) SBW_WIN(a, mC, gC, 1u | offC))
                    if (col1) SBW_ELEM(1, 1, SBW_WIN(a, mA, gA, 8u) SBW_WIN(a, mB, gB, 6u | offB) SBW_WIN(a, mC, gC, 2u | offC) SBW_WIN(a, mD, gD, 0u | offD))
                }
#undef SBW_ELEM
#undef SBW_WIN
            }
            __syncthreads();
        }
        {
        if (nxt < p.total_tiles) {
            const int f = nxt / p.groups_per_frame, gi = nxt - f * p.groups_per_frame;
            const int b = f / p.T, t = f - b * p.T;
            const int y0 = gi * SW_RB, p_lo = y0 >> 1;
            int rbn = p.Ho - y0;
            if (rbn > SW_RB) rbn = SW_RB;
            const int row_base = 2 * y0 - 3;
#pragma unroll
            for (int i = 0; i < NDY; ++i) {
                const int e = tid + 256 * i, pos = e >> 3, yl = (int)(((float)pos + 0.5f) * inv_wop), w = pos - yl * p.WoP;
                const bool ok = e < dy_total && yl < rbn && w < p.Wo;       // (no branch: the register arrays must not end up in scratch)
                xv[i] = *reinterpret_cast<const v4u*>(ok ? p.x + (((long)f * p.Ho + y0 + yl) * p.Wo + w) * STEM_C + c0
                                                           : reinterpret_cast<const bf16_t*>(g_stem_zero));
            }
            // the SBW_PR pooled rows from p_lo on are consecutive in memory; rows past the frame read as "no winner here"
            const int g_live = (p.Hp - p_lo) * p.Wp * 8;
#pragma unroll
            for (int i = 0; i < SBW_NG; ++i) {
                const int e = tid + 256 * i;
                const bool ok = e < g_total && e < g_live;
                const long o = (((long)f * p.Hp + p_lo) * p.Wp + (e >> 3)) * STEM_C + c0;
                gv[i] = *reinterpret_cast<const v4u*>(ok ? p.gpool + o : reinterpret_cast<const bf16_t*>(g_stem_zero));
                mv[i] = *reinterpret_cast<const v2u*>(ok ? p.amax + o : reinterpret_cast<const unsigned char*>(g_stem_zero));      // (never read: ph >= Hp is skipped)
            }
#pragma unroll
            for (int i = 0; i < NIN; ++i) {
                const int it = tid + 256 * i, rr = (int)(((float)it + 0.5f) * inv_nq), l = it - rr * nq;
                const int kt = rr / nrows, r = rr - kt * nrows;
                const int tt = t + kt - 2, iy = row_base + r;
                const bool ok = it < in_total && tt >= 0 && tt < p.T && iy >= 0 && iy < p.H;
                inv[i] = *reinterpret_cast<const f32x4*>(ok ? p.vid + (((long)b * p.T + tt) * p.H + iy) * p.W + 4 * l
                                                             : reinterpret_cast<const float*>(g_stem_zero));
            }
        }
        }
        if (tile < 0) {
            if (nxt >= p.total_tiles) break;
            continue;
        }

        for (int yl = 0; yl < rb; ++yl) {
            for (int xs = 0; xs < p.WoP; xs += 16) {
                const int pos0 = yl * p.WoP + xs;
                bf16x8 fa[2], fb[2];
                fa[0] = stem_frag_T<true>(sDY, 0, pos0, lane);
                fa[1] = stem_frag_T<true>(sDY, 32, pos0, lane);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int a = kbase[q] + (2 * yl) * 2 * p.PW + xs;
                    const unsigned* src = reinterpret_cast<const unsigned*>(sIn + (a & ~1));
                    const unsigned sh = (a & 1) * 16;
                    const unsigned d0 = src[0], d1 = src[1], d2 = src[2], d3 = src[3], d4 = src[4];
                    u32x4 fr;
                    fr.x = kvalid[q] ? __builtin_amdgcn_alignbit(d1, d0, sh) : 0u;
                    fr.y = kvalid[q] ? __builtin_amdgcn_alignbit(d2, d1, sh) : 0u;
                    fr.z = kvalid[q] ? __builtin_amdgcn_alignbit(d3, d2, sh) : 0u;
                    fr.w = kvalid[q] ? __builtin_amdgcn_alignbit(d4, d3, sh) : 0u;
                    fb[q] = __builtin_bit_cast(bf16x8, fr);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[q], acc[i][q], 0, 0, 0);
            }
        }
        if (nxt >= p.total_tiles) break;
    }
    float* dst = p.part + (long)blockIdx.x * (STEM_C * 245);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int k = (wave * 2 + q) * 32 + (lane & 31);
        if (k >= 245) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                dst[c * 245 + k] = acc[i][q][r];
            }
    }
}

extern "C" {

static int stem_fwd_grid(int B, int T, int H, int W) {
    const long tiles = (long)(((H / 2) * (W / 2) + 127) / 128) * B * T;
    return (int)(tiles < 768 ? tiles : 768);      // persistent blocks (3 per CU), weights staged once each
}

/* rows of [2][64] BatchNorm partials svsr_stem_conv_fwd writes for this shape (= its persistent workgroups) */
int svsr_stem_conv_fwd_stat_rows(int B, int T, int H, int W) { return (B < 1 || T < 1 || H < 8 || W < 8) ? 0 : stem_fwd_grid(B, T, H, W); }

/* bytes of the workspace the DMA-fed forward path wants (0: this shape takes the direct path and needs none) */
int64_t svsr_stem_conv_fwd_ws_bytes(int B, int T, int H, int W) {
    if ((H & 1) || (W & 7) || H < 8 || W < 8 || B < 1 || T < 1 || !svsr_tune_get(SVSR_TUNE_STEM_FWD_DMA)) return 0;
    return ((int64_t)B * T * H * (W + 8) + (int64_t)STEM_W_CHUNKS * 8) * 2;
}

int svsr_stem_conv_fwd(const float* vid, const float* w, void* out, float* stats, int B, int T, int H, int W, void* ws, int64_t ws_bytes,
                       hipStream_t stream) {
    if ((H & 1) || (W & 1) || H < 8 || W < 8) return SVSR_ERR_ARG;
    StemFwdArgs a;
    a.vid = vid; a.w = w; a.out = (bf16_t*)out; a.stats = stats;
    a.B = B; a.T = T; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2;
    const int HoWo = a.Ho * a.Wo;
    a.tiles_per_frame = (HoWo + 127) / 128;
    a.total_tiles = a.tiles_per_frame * B * T;
    const int span = 127 / a.Wo + 2;                 // output rows a 128-position tile can touch
    a.rows_in_max = 2 * (span - 1) + 7;
    a.WP = W + 6;
    const size_t lds = (size_t)STEM_C * STEM_WPITCH * 2 + (size_t)5 * a.rows_in_max * a.WP * 2;
    if (lds > 160 * 1024) return SVSR_ERR_ARG;
    static size_t lds_set = 0;
    if (lds > lds_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_stem_conv_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_set = lds;
    }
    const int grid = stem_fwd_grid(B, T, H, W);
    const int64_t need = svsr_stem_conv_fwd_ws_bytes(B, T, H, W);
    if (ws != nullptr && need > 0 && ws_bytes >= need) {
        bf16_t* vid16 = (bf16_t*)ws;
        bf16_t* wpack = vid16 + (int64_t)B * T * H * (W + 8);
        const long rows = (long)B * T * H;
        const int vid_blocks = (int)((rows * ((W + 8) / 8) + 255) / 256), w_blocks = (STEM_C * STEM_WPITCH + 255) / 256;
        hipLaunchKernelGGL(k_stem_prep, dim3(vid_blocks + w_blocks), dim3(256), 0, stream, vid, w, vid16, wpack, rows, W, vid_blocks);
        const size_t lds_d = (size_t)STEM_W_CHUNKS * 16 + ((size_t)5 * a.rows_in_max * ((W + 8) / 8) + 63) / 64 * 64 * 16;
        if (lds_d > 160 * 1024) return SVSR_ERR_ARG;
        static size_t lds_dma = 0;
        if (lds_d > lds_dma) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_stem_conv_fwd_dma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_d);
            lds_dma = lds_d;
        }
        hipLaunchKernelGGL(k_stem_conv_fwd_dma, dim3(grid), dim3(256), lds_d, stream, a, (const bf16_t*)vid16, (const bf16_t*)wpack);
        return svsr_check_launch();
    }
    hipLaunchKernelGGL(k_stem_conv_fwd, dim3(grid), dim3(256), lds, stream, a);
    return svsr_check_launch();
}

static int stem_wgrad_grid(int B, int T, int H) {
    const long tiles = (long)((H / 2 + SW_RB - 1) / SW_RB) * B * T;
    return (int)(tiles < 512 ? tiles : 512);
}

/* workspace (floats) svsr_stem_conv_wgrad needs: one [64][245] slab per persistent workgroup */
int svsr_stem_conv_wgrad_plan(int B, int T, int H, int W, int* splits, int64_t* part_floats) {
    if ((H & 1) || (W & 1) || H < 8 || W < 8 || B < 1 || T < 1) return SVSR_ERR_ARG;
    const int g = stem_wgrad_grid(B, T, H);
    if (splits) *splits = g;
    if (part_floats) *part_floats = (int64_t)g * STEM_C * 245;
    return SVSR_OK;
}

int svsr_colsum_rows(const float* ws, int nrows, int64_t ld, float* out0, int64_t n0, float* out1, int64_t n1, int accumulate, float scale,
                     hipStream_t stream);

int svsr_stem_conv_wgrad(const float* vid, const void* dy, float* dw, int B, int T, int H, int W, int use_tr, float* part,
                         int64_t part_floats, hipStream_t stream) {
    if ((H & 1) || (W & 1) || H < 8 || W < 8) return SVSR_ERR_ARG;
    const int grid = stem_wgrad_grid(B, T, H);
    if (part == nullptr || part_floats < (int64_t)grid * STEM_C * 245) return SVSR_ERR_ARG;
    StemWgradArgs a;
    a.vid = vid; a.dy = (const bf16_t*)dy; a.dw = dw; a.part = part;
    a.B = B; a.T = T; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2;
    a.WoP = (a.Wo + 15) / 16 * 16;
    a.PW = a.WoP + 8;
    a.groups_per_frame = (a.Ho + SW_RB - 1) / SW_RB;
    a.total_tiles = a.groups_per_frame * B * T;
    a.use_tr = use_tr;
    const size_t lds = (size_t)SW_RB * a.WoP * SW_DPITCH * 2 + (size_t)5 * (2 * SW_RB + 5) * 2 * a.PW * 2;
    if (lds > 160 * 1024) return SVSR_ERR_ARG;
    static size_t lds_set[2] = {0, 0};
    const void* fn = use_tr ? reinterpret_cast<const void*>(k_stem_conv_wgrad<true>) : reinterpret_cast<const void*>(k_stem_conv_wgrad<false>);
    if (lds > lds_set[use_tr ? 1 : 0]) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_set[use_tr ? 1 : 0] = lds;
    }
    // the register-pipelined kernel covers the shapes whose tile pieces fit its fixed prefetch registers (88 x 88 clips do)
    const bool pipe = use_tr && (W & 3) == 0 && a.PW / 2 <= 32 && SW_RB * a.WoP * 8 <= 8 * 256 && svsr_tune_get(SVSR_TUNE_STEM_WG_PIPE);
    if (pipe) {
        static size_t lds_pipe = 0;
        if (lds > lds_pipe) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_stem_conv_wgrad_pipe<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            lds_pipe = lds;
        }
        hipLaunchKernelGGL(k_stem_conv_wgrad_pipe<true>, dim3(grid), dim3(256), lds, stream, a);
    } else if (use_tr) hipLaunchKernelGGL(k_stem_conv_wgrad<true>, dim3(grid), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(k_stem_conv_wgrad<false>, dim3(grid), dim3(256), lds, stream, a);
    const int rc = svsr_check_launch();
    if (rc != SVSR_OK) return rc;
    return svsr_colsum_rows(part, grid, STEM_C * 245, dw, STEM_C * 245, nullptr, 0, 1, 1.0f, stream);
}


/* shapes the fused stem backward (BatchNorm/activation/pool apply + weight gradient in one pass) takes: 0 = call the two launches */
static bool stem_bwd_wgrad_shape(int B, int T, int H, int W, StemBwdWgradArgs& a, size_t& lds) {
    if ((H & 1) || (W & 3) || H < 8 || W < 8 || B < 1 || T < 1) return false;
    a.B = B; a.T = T; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2;
    a.Hp = (a.Ho - 1) / 2 + 1; a.Wp = (a.Wo - 1) / 2 + 1;
    a.WoP = (a.Wo + 15) / 16 * 16;
    a.PW = a.WoP + 8;
    a.groups_per_frame = (a.Ho + SW_RB - 1) / SW_RB;
    a.total_tiles = a.groups_per_frame * B * T;
    lds = (size_t)SW_RB * a.WoP * SW_DPITCH * 2 + (size_t)5 * (2 * SW_RB + 5) * 2 * a.PW * 2 + (size_t)SBW_PR * a.Wp * STEM_C * 3 + 5 * STEM_C * sizeof(float);
    return a.PW / 2 <= 32 && SW_RB * a.WoP * 8 <= 8 * 256 && 5 * (2 * SW_RB + 5) * (W / 4) <= 8 * 256 && SBW_PR * a.Wp * 8 <= SBW_NG * 256 && lds <= 160 * 1024;
}

int svsr_stem_bwd_wgrad_ok(int B, int T, int H, int W) {
    StemBwdWgradArgs a;
    size_t lds;
    return stem_bwd_wgrad_shape(B, T, H, W, a, lds) ? 1 : 0;
}

int svsr_stem_bwd_wgrad(const float* vid, const void* gpool, const void* amax, const void* x, const float* mean, const float* rstd,
                        const float* coef, float* dw, int B, int T, int H, int W, float* part, int64_t part_floats, hipStream_t stream) {
    StemBwdWgradArgs a;
    size_t lds;
    if (!stem_bwd_wgrad_shape(B, T, H, W, a, lds)) return SVSR_ERR_ARG;
    const int grid = stem_wgrad_grid(B, T, H);
    if (part == nullptr || part_floats < (int64_t)grid * STEM_C * 245) return SVSR_ERR_ARG;
    a.vid = vid; a.x = (const bf16_t*)x; a.gpool = (const bf16_t*)gpool; a.amax = (const unsigned char*)amax;
    a.mean = mean; a.rstd = rstd; a.coef = coef; a.part = part;
    // prefetch-register variants: 88-wide clips need 6 + 6 pieces per thread, the general form holds 8 + 8
    const int v = (SW_RB * a.WoP * 8 <= 6 * 256) ? ((5 * (2 * SW_RB + 5) * (W / 4) <= 6 * 256) ? 0 : 1) : 2;
    const void* fn = v == 0 ? reinterpret_cast<const void*>(k_stem_bwd_wgrad<6, 6>)
                   : v == 1 ? reinterpret_cast<const void*>(k_stem_bwd_wgrad<6, 8>) : reinterpret_cast<const void*>(k_stem_bwd_wgrad<8, 8>);
    static size_t lds_set[3] = {0, 0, 0};
    if (lds > lds_set[v]) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_set[v] = lds;
    }
    if (v == 0) hipLaunchKernelGGL((k_stem_bwd_wgrad<6, 6>), dim3(grid), dim3(256), lds, stream, a);
    else if (v == 1) hipLaunchKernelGGL((k_stem_bwd_wgrad<6, 8>), dim3(grid), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((k_stem_bwd_wgrad<8, 8>), dim3(grid), dim3(256), lds, stream, a);
    const int rc = svsr_check_launch();
    if (rc != SVSR_OK) return rc;
    return svsr_colsum_rows(part, grid, STEM_C * 245, dw, STEM_C * 245, nullptr, 0, 1, 1.0f, stream);
}

}  // extern "C"
