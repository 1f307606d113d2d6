// 3x3 / stride-1 / pad-1 NHWC convolution for 64 -> 64 channels (ResNet18 layer1: forward and data-gradient), as a
// PERSISTENT workgroup per CU with the whole weight tensor resident in LDS (gfx950).
// Replaces conv3x3(64, 64) of layer1's BasicBlocks and its input-gradient (reference LRW/video/src/tcn/models/resnet.py:
// 8-10,36,53,59-72; timm resnet18 twin).
//
// Why a dedicated kernel: at layer1 sizes (449k positions, K = 576) the generic implicit-GEMM kernel re-reads every
// activation row nine times (once per tap) and re-streams the 72 KiB of weights for every 128-row tile: ~0.8 GB of
// L2->LDS traffic for 33 GFLOP.  Here the reduction is laid out over ZERO-PADDED pixel coordinates q of a (H+2)x(W+2)
// grid flattened over the batch, where tap (dy,dx) is the constant row shift dy*(W+2)+dx: one LDS tile of
// 256 + 2(W+3) rows serves all nine taps of a 256-position chunk, is fetched ONCE by LDS-DMA while the previous
// chunk is being contracted, and the weights [9][64][64] are fetched once per workgroup.
// 8 waves (2 per SIMD, so one wave's ds_read_b128 latency hides under the other's MFMAs); a wave owns 32 positions x 64
// channels and issues 9 taps x 4 k-steps x 2 MFMA 32x32x16 per chunk.  The epilogue stores straight from the
// accumulators (lanes = channels, 64-byte segments) so no LDS read sits between the in-flight DMA and the stores.
#include "common.h"

#define C64_CH 256              // padded positions per chunk
#define C64_XR 320              // A-tile rows: 256 + 2*(WP+1) <= 320  ->  W <= 29
#define C64_LDS_W (9 * 64 * 64) // weights, bf16 elements
#define C64_LDS_A (C64_XR * 64) // one A tile, bf16 elements
#define C64_THREADS 512
#define C64_TAB_PAD 64          // svsr_conv3x3_c64_pixtab: entries in front of padded coordinate 0 (a tile starts W + 3 rows before its chunk)

struct Conv64Args {
    const bf16_t* in;       // [Nimg][H][W][64]
    const bf16_t* wt;       // [64][9][64]   (row = output channel, then tap, then input channel)
    bf16_t* out;            // [Nimg][H][W][64]
    const bf16_t* addend;   // optional, laid out like out (may alias it)
    float* stats;           // optional BatchNorm partials [gridDim.x][2][64]: one row per (persistent) workgroup
    int Nimg, H, W, WP, Q, Qtot, total_chunks;
    float inv_q, inv_wp;
    int dy[9], dx[9], tw[9];
    // BatchNorm-backward fusion (see svsr_conv3x3_c64_dgrad_bn): out = g = (y > 0 ? result : 0), stats = sums of {g, g * (x - mean) * rstd}
    const bf16_t* bnb_y;
    const bf16_t* bnb_x;
    const float* bnb_mean;
    const float* bnb_rstd;
    const float* bnb_gamma;     // bnb_y == nullptr: mask recomputed as bn(x) > 0 (forward's own expression), y not read
    const float* bnb_beta;
    int bnb_act;                // 1 ReLU, 2 Swish (bnb_y = the residual input or null; see svsr_igemm_dgrad_bn)
    const int* pixtab;          // svsr_conv3x3_c64_pixtab's table (device copy): source pixel of every padded coordinate, or null
};

__device__ unsigned g_c64_zero_page[64];

#define C64_SWZ(row, chunk) ((row) * 64 + ((((chunk) ^ (((row) >> 1) & 7))) << 3))

__device__ __forceinline__ int c64_pixel(const Conv64Args& p, int q) {
    if (q < 0 || q >= p.Qtot) return -1;
    int n = (int)((float)q * p.inv_q);
    int rem = q - n * p.Q;
    if (rem < 0) { n--; rem += p.Q; } else if (rem >= p.Q) { n++; rem -= p.Q; }
    int yp = (int)((float)rem * p.inv_wp);
    int xp = rem - yp * p.WP;
    if (xp < 0) { yp--; xp += p.WP; } else if (xp >= p.WP) { yp++; xp -= p.WP; }
    if (yp < 1 || yp > p.H || xp < 1 || xp > p.W) return -1;
    return (n * p.H + (yp - 1)) * p.W + (xp - 1);
}

// SWISH: the Swish variant of the BatchNorm-backward epilogue (LRS trunk) is its own instantiation — compiled into the ReLU / plain kernel
// its temporaries push the 255-register MFMA loop over the limit and spill.
template <bool SWISH>
__global__ __launch_bounds__(C64_THREADS) void k_conv3x3_c64(const Conv64Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sW = reinterpret_cast<bf16_t*>(smem_raw);                // [9*64 rows][64]
    bf16_t* sA = sW + C64_LDS_W;                                      // [2][C64_XR][64]
    int* sPix = reinterpret_cast<int*>(sA + 2 * C64_LDS_A);           // [2][256] output pixel index or -1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;    // 8 waves
    const int slot = tid & 7, r0 = tid >> 3;                          // r0 in [0, 64)
    const int csw = slot ^ ((r0 >> 1) & 7);                           // rows r0 + 64*i share (row>>1)&7
    const int wrow = __builtin_amdgcn_readfirstlane(wave) * 8;
    const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(g_c64_zero_page) + slot * 8;
    const int halo = p.WP + 1;

    // ---- weights: 576 rows [tap][co] x 8 pieces, 9 DMA instructions per thread ---------------------------------------
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int row = r0 + 64 * i;              // = t*64 + co with t == i
        const bf16_t* src = p.wt + ((long)(row & 63) * 9 + p.tw[i]) * 64 + csw * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sW + (wrow + 64 * i) * 64), 16, 0, 0);
    }

    // source pixels of the five rows this thread stages: from the table (svsr_conv3x3_c64_pixtab; the entries of a chunk are requested
    // one chunk ahead, pixn) — computing them costs two divisions by reciprocal with fix-ups per row, ~500 instructions per chunk
    // and thread in front of 72 MFMAs per wave — or, without a table, by that arithmetic
    const bool tab = p.pixtab != nullptr;
    int pixn[5];
    auto table_rows = [&](int c) {
        const int* t = p.pixtab + C64_TAB_PAD + c * C64_CH - halo + r0;
#pragma unroll
        for (int i = 0; i < 5; ++i) pixn[i] = t[64 * i];
    };
    auto stage = [&](int c, int buf) {
        const int q0 = c * C64_CH;
        bf16_t* dst = sA + buf * C64_LDS_A + wrow * 64;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int rr = r0 + 64 * i;
            const int pix = tab ? pixn[i] : c64_pixel(p, q0 - halo + rr);
            const bf16_t* src = pix >= 0 ? p.in + (long)pix * 64 + csw * 8 : zero_src;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + i * 64 * 64), 16, 0, 0);
            const int pl = rr - halo;             // chunk-local output position of this row
            if (slot == 0 && pl >= 0 && pl < C64_CH) sPix[buf * C64_CH + pl] = pix;
        }
    };

    float st_s[8], st_q[8];         // channels slot*8 .. +7 of the rows this thread stores, summed over its chunks
#pragma unroll
    for (int k = 0; k < 8; ++k) { st_s[k] = 0.f; st_q[k] = 0.f; }
    const bool bnb = p.bnb_x != nullptr;
    const bool from_x = p.bnb_y == nullptr;
    // per-channel constants of the fused BatchNorm backward live in LDS, not in registers: the MFMA loop below already uses ~240 VGPRs
    // and 32 more, live across the persistent loop, spilled to scratch (90 instead of 74 us per launch)
    float* sC = reinterpret_cast<float*>(sPix + 2 * C64_CH);        // [4][64]: mean, rstd, gamma*rstd, beta - mean*gamma*rstd
    if (bnb && tid < 64) {
        const float m = p.bnb_mean[tid], r = p.bnb_rstd[tid];
        sC[tid] = m; sC[64 + tid] = r;
        const bool affine = from_x || SWISH;
        const float c = affine ? p.bnb_gamma[tid] * r : 0.f;
        sC[128 + tid] = c;
        sC[192 + tid] = affine ? __builtin_fmaf(-m, c, p.bnb_beta[tid]) : 0.f;
    }
    int c = blockIdx.x, buf = 0;
    if (tab && c < p.total_chunks) table_rows(c);
    if (c < p.total_chunks) stage(c, 0);
    if (tab && c + (int)gridDim.x < p.total_chunks) table_rows(c + gridDim.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // weights + first tile
    for (; c < p.total_chunks; c += gridDim.x, buf ^= 1) {
        // every wave's part of this chunk's tile has landed (each waited before arriving here), and everybody is done with
        // the previous chunk's tile / staging, which the next DMA overwrites
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (c + (int)gridDim.x < p.total_chunks) stage(c + gridDim.x, buf ^ 1);
        if (tab && c + 2 * (int)gridDim.x < p.total_chunks) table_rows(c + 2 * gridDim.x);
        const bf16_t* cA = sA + buf * C64_LDS_A;

        f32x16 acc[2];              // D[row = output channel j*32 + ..][col = position]: 4 consecutive channels per lane
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int row = wave * 32 + (lane & 31) + halo + p.dy[t] * p.WP + p.dx[t];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ch = ks * 2 + (lane >> 5);
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(cA + C64_SWZ(row, ch));
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int wr = t * 64 + j * 32 + (lane & 31);
                    const bf16x8 fb = *reinterpret_cast<const bf16x8*>(sW + C64_SWZ(wr, ch));
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, acc[j], 0, 0, 0);
                }
            }
        }

        // ---- epilogue -------------------------------------------------------------------------------------------------
        // The next chunk's DMA was issued before the MFMA block and has landed long ago: this wait is free and keeps the
        // global stores below (counted by vmcnt on gfx950) out of the top-of-loop synchronisation.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // everybody is done reading tile `buf`
        bf16_t* sO = sA + buf * C64_LDS_A;      // staging [256 positions][64 channels], 16-byte pieces XOR-swizzled by row
        {
            const int pos = wave * 32 + (lane & 31);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = j * 32 + 8 * g + 4 * (lane >> 5);
                    uint2 v;
                    v.x = pack2bf(acc[j][4 * g + 0], acc[j][4 * g + 1]);
                    v.y = pack2bf(acc[j][4 * g + 2], acc[j][4 * g + 3]);
                    *reinterpret_cast<uint2*>(sO + pos * 64 + ((((co >> 3) ^ (pos & 7))) << 3) + (co & 7)) = v;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int* pix_tab = sPix + buf * C64_CH;
        int pixv[4];
        u32x4 piece[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 64 * i;
            pixv[i] = pix_tab[row];
            piece[i] = *reinterpret_cast<const u32x4*>(sO + row * 64 + ((slot ^ (row & 7)) << 3));
        }
        u32x4 add[4];
        if (p.addend != nullptr) {
#pragma unroll
            for (int i = 0; i < 4; ++i) add[i] = *reinterpret_cast<const u32x4*>(p.addend + (long)(pixv[i] >= 0 ? pixv[i] : 0) * 64 + slot * 8);
        }
        if (bnb) {
            u32x4 y8[4], x8[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long o = (long)(pixv[i] >= 0 ? pixv[i] : 0) * 64 + slot * 8;
                x8[i] = *reinterpret_cast<const u32x4*>(p.bnb_x + o);
                if (!from_x) y8[i] = *reinterpret_cast<const u32x4*>(p.bnb_y + o);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (pixv[i] < 0) continue;
                float a[8], yv[8], xv[8];
                // (the per-channel constants are re-read from LDS for every row: held across the four rows they cost 32 registers and spilled)
                float mu[8], rs[8], sc[8], sh[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 m4 = *reinterpret_cast<const f32x4*>(sC + slot * 8 + 4 * h), r4 = *reinterpret_cast<const f32x4*>(sC + 64 + slot * 8 + 4 * h);
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(sC + 128 + slot * 8 + 4 * h), h4 = *reinterpret_cast<const f32x4*>(sC + 192 + slot * 8 + 4 * h);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { mu[4 * h + k] = m4[k]; rs[4 * h + k] = r4[k]; sc[4 * h + k] = c4[k]; sh[4 * h + k] = h4[k]; }
                }
                unpack8(piece[i], a);
                unpack8(x8[i], xv);
                if (from_x) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) yv[k] = __builtin_fmaf(xv[k], sc[k], sh[k]);
                } else {
                    unpack8(y8[i], yv);
                }
                if (p.addend != nullptr) {
                    float b[8];
                    unpack8(add[i], b);
#pragma unroll
                    for (int k = 0; k < 8; ++k) a[k] = bf2f(f2bf(a[k] + b[k]));
                }
                if (SWISH) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float z = from_x ? yv[k] : __builtin_fmaf(xv[k], sc[k], sh[k]) + yv[k];     // (from_x: yv already is bn(x))
                        a[k] = bf2f(f2bf(a[k] * swish_grad(z)));
                        st_s[k] += a[k];
                        st_q[k] += a[k] * (xv[k] - mu[k]) * rs[k];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        a[k] = yv[k] > 0.f ? a[k] : 0.f;
                        st_s[k] += a[k];
                        st_q[k] += a[k] * (xv[k] - mu[k]) * rs[k];
                    }
                }
                *reinterpret_cast<u32x4*>(p.out + (long)pixv[i] * 64 + slot * 8) = pack8(a);
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (pixv[i] < 0) continue;
            float a[8];
            unpack8(piece[i], a);
#pragma unroll
            for (int k = 0; k < 8; ++k) { st_s[k] += a[k]; st_q[k] += a[k] * a[k]; }
            u32x4 v = piece[i];
            if (p.addend != nullptr) {
                float b[8];
                unpack8(add[i], b);
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] += b[k];
                v = pack8(a);
            }
            *reinterpret_cast<u32x4*>(p.out + (long)pixv[i] * 64 + slot * 8) = v;
        }
    }
    if (p.stats != nullptr) {
        // threads tid = slot + 8*m share a channel group: reduce through LDS in a fixed order, then one plain store per channel
        // and statistic into this workgroup's row of the partials (svsr_bn_finalize adds the rows)
        __syncthreads();
        float* sred = reinterpret_cast<float*>(sA);        // [512][16]
#pragma unroll
        for (int k = 0; k < 8; ++k) { sred[tid * 16 + k] = st_s[k]; sred[tid * 16 + 8 + k] = st_q[k]; }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, ch = tid & 63, sl8 = ch >> 3, k = ch & 7;
            float s = 0.f;
            for (int m = 0; m < 64; ++m) s += sred[(sl8 + 8 * m) * 16 + which * 8 + k];
            p.stats[((long)blockIdx.x * 2 + which) * 64 + ch] = s;
        }
    }
}

static int c64_grid(long total_chunks) {
    const int cus = svsr_reduction_cus();      // (a fixed number, not the device's: one BatchNorm partial row per persistent workgroup)
    return (int)(total_chunks < cus ? total_chunks : cus);
}

/* rows of [2][64] BatchNorm partials svsr_conv3x3_c64 writes for this shape (= its persistent workgroups) */
extern "C" int svsr_conv3x3_c64_stat_rows(int Nimg, int H, int W) {
    if (Nimg < 1 || H < 1 || W < 1) return 0;
    return c64_grid(((long)Nimg * (H + 2) * (W + 2) + C64_CH - 1) / C64_CH);
}

static int c64_run(const void* in, const void* wt, void* out, const void* addend, float* stats, int Nimg, int H, int W,
                   const int* dy, const int* dx, const int* tw, const void* bnb_y, const void* bnb_x, const float* bnb_mean,
                   const float* bnb_rstd, const float* bnb_gamma, const float* bnb_beta, int bnb_act, const int* pixtab, hipStream_t stream) {
    if (W + 2 > (C64_XR - C64_CH) / 2 - 1 || H < 1 || W < 1 || Nimg < 1) return SVSR_ERR_ARG;
    Conv64Args a;
    a.pixtab = pixtab;
    a.in = (const bf16_t*)in; a.wt = (const bf16_t*)wt; a.out = (bf16_t*)out; a.addend = (const bf16_t*)addend; a.stats = stats;
    a.bnb_y = (const bf16_t*)bnb_y; a.bnb_x = (const bf16_t*)bnb_x; a.bnb_mean = bnb_mean; a.bnb_rstd = bnb_rstd;
    a.bnb_gamma = bnb_gamma; a.bnb_beta = bnb_beta; a.bnb_act = bnb_act;
    a.Nimg = Nimg; a.H = H; a.W = W; a.WP = W + 2; a.Q = (H + 2) * (W + 2);
    const long qtot = (long)Nimg * a.Q;
    if (qtot >= (1L << 24) || (long)Nimg * H * W >= (1L << 25)) return SVSR_ERR_ARG;
    a.Qtot = (int)qtot;
    a.total_chunks = (a.Qtot + C64_CH - 1) / C64_CH;
    a.inv_q = 1.0f / (float)a.Q; a.inv_wp = 1.0f / (float)a.WP;
    for (int i = 0; i < 9; ++i) { a.dy[i] = dy[i]; a.dx[i] = dx[i]; a.tw[i] = tw[i]; }
    const size_t lds = (size_t)(C64_LDS_W + 2 * C64_LDS_A) * sizeof(bf16_t) + 2 * C64_CH * sizeof(int) + 4 * 64 * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_c64<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_c64<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int grid = c64_grid(a.total_chunks);
    if (bnb_x != nullptr && bnb_act == 2) hipLaunchKernelGGL(k_conv3x3_c64<true>, dim3(grid), dim3(C64_THREADS), lds, stream, a);
    else hipLaunchKernelGGL(k_conv3x3_c64<false>, dim3(grid), dim3(C64_THREADS), lds, stream, a);
    return svsr_check_launch();
}

/* svsr_conv3x3_c64_pixtab: the pixel table for a shape — entry C64_TAB_PAD + q = pixel index of the padded
 * coordinate q ((H+2) x (W+2) grid per image, flattened over the batch) or -1 (padding ring, outside the batch); covers every tile row
 * of every chunk.  Returns the number of entries; with out == nullptr only that (host memory; the caller keeps a device copy). */
extern "C" int64_t svsr_conv3x3_c64_pixtab(int Nimg, int H, int W, int* out, int64_t cap) {
    if (Nimg < 1 || H < 1 || W < 1 || W + 2 > (C64_XR - C64_CH) / 2 - 1) return -SVSR_ERR_ARG;
    const long Q = (long)(H + 2) * (W + 2), qtot = (long)Nimg * Q;
    const long chunks = (qtot + C64_CH - 1) / C64_CH, n = C64_TAB_PAD + chunks * C64_CH + C64_XR;
    if (qtot >= (1L << 24)) return -SVSR_ERR_ARG;
    if (out == nullptr) return n;
    if (cap < n) return -SVSR_ERR_ARG;
    for (long e = 0; e < n; ++e) {
        const long q = e - C64_TAB_PAD;
        int pix = -1;
        if (q >= 0 && q < qtot) {
            const long img = q / Q, rem = q - img * Q, yp = rem / (W + 2), xp = rem - yp * (W + 2);
            if (yp >= 1 && yp <= H && xp >= 1 && xp <= W) pix = (int)((img * H + (yp - 1)) * W + (xp - 1));
        }
        out[e] = pix;
    }
    return n;
}

extern "C" int svsr_conv3x3_c64(const void* in, const void* wt, void* out, const void* addend, float* stats, int Nimg, int H, int W,
                                const int* dy, const int* dx, const int* tw, const int* pixtab, hipStream_t stream) {
    return c64_run(in, wt, out, addend, stats, Nimg, H, W, dy, dx, tw, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, pixtab, stream);
}

/* svsr_conv3x3_c64_dgrad_bn: the data gradient of a 64 -> 64 convolution whose result is the gradient of a BatchNorm + ReLU output
 * y = relu(bn(x) [+ residual]) (reference tcn/models/resnet.py:59-72 backward).  Stores g = (y > 0 ? result [+ addend] : 0) and writes
 * per workgroup the column sums of g and g * (x - mean) * rstd into stats[svsr_conv3x3_c64_stat_rows][2][64]: the first pass of the
 * BatchNorm backward, taken while the tile is in registers (svsr_bn_bwd_from_stats finishes it).  addend may alias out.
 * y == nullptr (output without residual branch): mask = bn(x) > 0 recomputed with gamma / beta, y is not read. */
extern "C" int svsr_conv3x3_c64_dgrad_bn(const void* in, const void* wt, void* out, const void* addend, float* stats, int Nimg, int H, int W,
                                         const int* dy, const int* dx, const int* tw, const void* y, const void* x, const float* mean,
                                         const float* rstd, const float* gamma, const float* beta, int act, const int* pixtab, hipStream_t stream) {
    if (x == nullptr || mean == nullptr || rstd == nullptr || stats == nullptr || (act != 1 && act != 2)) return SVSR_ERR_ARG;
    if ((y == nullptr || act == 2) && (gamma == nullptr || beta == nullptr)) return SVSR_ERR_ARG;
    return c64_run(in, wt, out, addend, stats, Nimg, H, W, dy, dx, tw, y, x, mean, rstd, gamma, beta, act, pixtab, stream);
}
