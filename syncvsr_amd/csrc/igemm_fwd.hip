// Implicit-GEMM forward contraction on MFMA (gfx950): NHWC bf16 convolution forward / data-gradient and dense linear
// layers (a linear layer is the 1-tap, 1x1-pixel case).
//
// Replaces the ATen/MIOpen kernels the reference reaches through torch.nn:
//   nn.Conv2d 3x3 / 1x1 of the ResNet18 trunk   (reference LRW/video/src/tcn/models/resnet.py:8-16,36,53; timm twin)
//   nn.Linear of the BERT encoder and the heads (reference LRW/video/src/lightning.py:82,92,107,161,168)
// and their input-gradients.
//
// Block = 256 threads = 4 waves (2x2), tile BM positions x BN output channels, K step 64 channels of one tap.
// Main kernel (k_igemm_fwd_glds): A (gathered activation rows) and B (weight rows) go global -> LDS by direct DMA
// (global_load_lds_dwordx4, 1 KiB = 8 rows x 128 B per wave instruction) into a 3-deep ring, two tiles in flight behind a
// counted s_waitcnt vmcnt(N) + one s_barrier per K step; the XOR swizzle that makes the ds_read_b128 fragment reads
// conflict-free is applied on the per-lane SOURCE address (the LDS image of a DMA is lane-linear), rows outside the grid
// read a zero page.
// Epilogue: accumulators -> LDS fp32 -> +bias +addend, exact GELU, 16-byte stores; optional BatchNorm partial sums, one row
// of [2][Co] per M tile (plain stores, no atomics: svsr_bn_finalize adds the rows in a fixed order, so a step is reproducible).

#include <algorithm>
#include <map>
#include <vector>

#include "common.h"
#include "igemm_fwd.h"

// ---------------------------------------------------------------------------------------------------------------------
// Launch plan (host-built, device-resident int32 words): the rows of the contraction are grouped into CLASSES of output
// positions that share the same set of in-grid taps, so the kernel never multiplies by the zero padding of a convolution
// (3x3 / pad 1 on 3x3, 6x6, 11x11 maps: 40 %, 21 %, 12 % of an im2col contraction are zeros) and needs no per-row masks:
//   words[0] = number of classes, words[1] = word offset of the position table
//   class c at words[2 + 24 c]: { ntaps, P, pos_off, tile_begin, delta[9], tw[9], pad[2] }
//       rows of the class: m = n * P + j  (image n, j-th position of the class), M_c = Nimg * P, tiled by BM from tile_begin
//       tap t reads source pixel src + delta[t] with weight tap tw[t]
//   position table: pairs (src, dst) = source-centre / target pixel index inside one image, per class at pos_off
// A stride-2 data gradient is ONE launch: its four output-parity classes are just more classes (with their own taps).
// Classes are ordered by decreasing tap count, so the short ones fill the tail of the launch.
// ---------------------------------------------------------------------------------------------------------------------
#define PLAN_CLS_WORDS 24
#define PLAN_HDR_WORDS 2

__device__ unsigned g_zero_page[64];     // 256 zero bytes: DMA source for rows outside the grid


// ---------------------------------------------------------------------------------------------------------------------
// shared epilogue
// ---------------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int TM, int TN>
__device__ __forceinline__ void igemm_epilogue(const IgemmFwdArgs& p, f32x16 (&acc)[TM][TN], unsigned char* smem_raw, const long* sRow,
                                               int wm0, int wn0, int n0, int tid, bool active, int m_tile) {
    // tid: 0..255 inside the group of four waves that owns the tile; `active` is false for the waves of a second K group, which only
    // take part in the barriers
    constexpr int WN = BN / 2;
    const int lane = tid & 63, wave = tid >> 6;
    // per-channel statistics from the fp32 accumulators (rows outside M carry exact zeros)
    float st_s[TN], st_q[TN];
    const bool bnb = p.bnb_x != nullptr;
    if (p.stats != nullptr && !bnb) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; s += v; q += v * v; }
            st_s[j] = s + __shfl_xor(s, 32, 64);
            st_q[j] = q + __shfl_xor(q, 32, 64);
        }
    }
    // accumulators -> LDS fp32 [BM][BN] (the tile buffers are dead: the caller has passed a barrier after its last read)
    float* sOut = reinterpret_cast<float*>(smem_raw);
    if (active) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    sOut[row * BN + wn0 + j * 32 + (lane & 31)] = acc[i][j][r];
                }
    }
    __syncthreads();
    const bool vec_pitch = (p.out_pitch & 7) == 0;
    constexpr int CV = BN / 8;
    static_assert(256 % CV == 0, "a thread keeps its column group across the store loop");
    // a thread's 8 output channels are the same in every iteration: their bias is fetched once, ahead of the loop (inside it
    // the load's L2 round trip was paid per iteration — half of a small linear layer's epilogue)
    float bias8[8];
    {
        const int nb = n0 + (tid % CV) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) bias8[k] = (p.bias != nullptr && nb + k < p.Co) ? p.bias[nb + k] : 0.f;
    }
    // A thread keeps its 8-column group and walks rows rbase, rbase + RSTEP, ...  When the whole group is inside Co (the usual case) the
    // row offsets, the staged accumulators and the addend pieces of ALL its rows are requested before the first one is used: the
    // loop form paid one LDS round trip — and, with an addend, one global round trip — per row, which was most of the epilogue.
    constexpr int ITERS = BM * CV / 256, RSTEP = 256 / CV;
    static_assert(BM * CV % 256 == 0, "whole rows per pass");
    const int c8 = tid % CV, rbase = tid / CV, n = n0 + c8 * 8;
    float bs1[8], bs2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { bs1[k] = 0.f; bs2[k] = 0.f; }
    if (bnb) {
        // data-gradient launch feeding a BatchNorm+ReLU backward (host: vec_pitch, Co % 8 == 0, no bias / activation / dropout, bf16 out):
        // the global operands (y, x, addend pieces) of ALL rows are requested before the first is used — one round trip per tile —
        // while the staged accumulators are read from LDS row by row (the full-tile batch of the path below plus these would not fit)
        float mu[8], rs[8];
        if (active) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { mu[k] = p.bnb_mean[n + k]; rs[k] = p.bnb_rstd[n + k]; }
            const bool from_x = p.bnb_y == nullptr, swish_act = p.bnb_act == 2;
            float sc[8], sh[8];
            if (from_x || swish_act) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { sc[k] = p.bnb_gamma[n + k] * rs[k]; sh[k] = __builtin_fmaf(-mu[k], sc[k], p.bnb_beta[n + k]); }
            }
            long offs[ITERS];
            u32x4 add8[ITERS], y8[ITERS], x8[ITERS];
#pragma unroll
            for (int i = 0; i < ITERS; ++i) offs[i] = sRow[rbase + RSTEP * i];
#pragma unroll
            for (int i = 0; i < ITERS; ++i) {
                const long o = (offs[i] >= 0 ? offs[i] : 0) + n;
                x8[i] = *reinterpret_cast<const u32x4*>(p.bnb_x + o);
                if (!from_x) y8[i] = *reinterpret_cast<const u32x4*>(p.bnb_y + o);
                if (p.addend != nullptr) add8[i] = *reinterpret_cast<const u32x4*>(p.addend + o);
            }
#pragma unroll
            for (int i = 0; i < ITERS; ++i) {
                if (offs[i] < 0) continue;
                const int r = rbase + RSTEP * i;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(sOut + r * BN + c8 * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(sOut + r * BN + c8 * 8 + 4);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                float yv[8], xv[8];
                unpack8(x8[i], xv);
                if (from_x) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) yv[k] = __builtin_fmaf(xv[k], sc[k], sh[k]);
                } else {
                    unpack8(y8[i], yv);
                }
                if (p.addend != nullptr) {
                    float a8[8];
                    unpack8(add8[i], a8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += a8[k];
                }
                // sums over the values the apply pass reads back (rounded to bf16): mean(g) is then the mean of what it is subtracted from
                if (swish_act) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float z = __builtin_fmaf(xv[k], sc[k], sh[k]) + (from_x ? 0.f : yv[k]);       // yv: the residual input here
                        v[k] = bf2f(f2bf(bf2f(f2bf(v[k])) * swish_grad(z)));
                        bs1[k] += v[k];
                        bs2[k] += v[k] * (xv[k] - mu[k]) * rs[k];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        v[k] = yv[k] > 0.f ? bf2f(f2bf(v[k])) : 0.f;
                        if (p.alpha != 1.f) v[k] = bf2f(f2bf(v[k] * p.alpha));       // svsr_igemm_dgrad_relu: the dropout scale of the masked units (uniform branch)
                        bs1[k] += v[k];
                        bs2[k] += v[k] * (xv[k] - mu[k]) * rs[k];
                    }
                }
                *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.out) + offs[i] + n) = pack8(v);
            }
        }
    } else if (active && vec_pitch && n + 8 <= p.Co && p.epi_batched) {
        long offs[ITERS];
        f32x4 lo[ITERS], hi[ITERS];
        u32x4 add8[ITERS];
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int r = rbase + RSTEP * i;
            offs[i] = sRow[r];
            lo[i] = *reinterpret_cast<const f32x4*>(sOut + r * BN + c8 * 8);
            hi[i] = *reinterpret_cast<const f32x4*>(sOut + r * BN + c8 * 8 + 4);
        }
        if (p.addend != nullptr) {
#pragma unroll
            for (int i = 0; i < ITERS; ++i)
                add8[i] = *reinterpret_cast<const u32x4*>(p.addend + (offs[i] >= 0 ? offs[i] : 0) + n);
        }
        const unsigned key = p.drop.seed != nullptr ? drop_key(p.drop) : 0u;
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const long off = offs[i];
            if (off < 0) continue;
            float v[8] = {lo[i][0], lo[i][1], lo[i][2], lo[i][3], hi[i][0], hi[i][1], hi[i][2], hi[i][3]};
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += bias8[k];
            if (p.act == 1) {
                if (p.out_pre != nullptr) *reinterpret_cast<u32x4*>(p.out_pre + off + n) = pack8(v);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = gelu_erf(v[k]);
            } else if (p.act == 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
            }
            if (p.drop.seed != nullptr) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = drop_keep(key, p.drop.thresh, (unsigned)(off + n + k)) ? v[k] * p.drop.scale : 0.f;
            }
            if (p.alpha != 1.f) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] *= p.alpha;
            }
            if (p.addend != nullptr) {
                float a8[8];
                unpack8(add8[i], a8);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += a8[k];
            }
            if (p.out_f32) {
                float* o = reinterpret_cast<float*>(p.out) + off + n;
                *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
            } else {
                *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.out) + off + n) = pack8(v);
            }
        }
    } else
    for (int task = active ? tid : BM * CV; task < BM * CV; task += 256) {
        const int r = task / CV, c8 = task - r * CV;
        const long off = sRow[r];
        const int n = n0 + c8 * 8;
        if (off < 0 || n >= p.Co) continue;
        float v[8];
        const f32x4 lo = *reinterpret_cast<const f32x4*>(sOut + r * BN + c8 * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(sOut + r * BN + c8 * 8 + 4);
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        if (vec_pitch && n + 8 <= p.Co) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += bias8[k];
            if (p.act == 1) {
                if (p.out_pre != nullptr) *reinterpret_cast<u32x4*>(p.out_pre + off + n) = pack8(v);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = gelu_erf(v[k]);
            } else if (p.act == 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
            }
            if (p.drop.seed != nullptr) {
                const unsigned key = drop_key(p.drop);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = drop_keep(key, p.drop.thresh, (unsigned)(off + n + k)) ? v[k] * p.drop.scale : 0.f;
            }
            if (p.alpha != 1.f) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] *= p.alpha;
            }
            if (p.addend != nullptr) {
                float a8[8];
                unpack8(*reinterpret_cast<const u32x4*>(p.addend + off + n), a8);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += a8[k];
            }
            if (p.out_f32) {
                float* o = reinterpret_cast<float*>(p.out) + off + n;
                *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
            } else {
                *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.out) + off + n) = pack8(v);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (n + k >= p.Co) break;
                float x = v[k] + bias8[k];
                if (p.act == 1) {
                    if (p.out_pre != nullptr) p.out_pre[off + n + k] = f2bf(x);
                    x = gelu_erf(x);
                } else if (p.act == 2) {
                    x = fmaxf(x, 0.f);
                }
                if (p.drop.seed != nullptr) x = drop_keep(drop_key(p.drop), p.drop.thresh, (unsigned)(off + n + k)) ? x * p.drop.scale : 0.f;
                x *= p.alpha;
                if (p.addend != nullptr) x += bf2f(p.addend[off + n + k]);
                if (p.out_f32) reinterpret_cast<float*>(p.out)[off + n + k] = x;
                else reinterpret_cast<bf16_t*>(p.out)[off + n + k] = f2bf(x);
            }
        }
    }
    if (bnb) {
        // threads rbase * CV + c8 share the column group c8: their sums are added through LDS in the order of rbase
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_raw);   // [256][16]
        if (active) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { red[tid * 16 + k] = bs1[k]; red[tid * 16 + 8 + k] = bs2[k]; }
        }
        __syncthreads();
        for (int c = active ? tid : 2 * BN; c < 2 * BN; c += 256) {
            const int which = c / BN, cc = c - which * BN;
            float s = 0.f;
            for (int rb = 0; rb < RSTEP; ++rb) s += red[(rb * CV + (cc >> 3)) * 16 + which * 8 + (cc & 7)];
            if (n0 + cc < p.Co) p.stats[((long)m_tile * 2 + which) * p.Co + n0 + cc] = s;
        }
    } else if (p.stats != nullptr) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_raw);   // [4 waves][WN][2]
#pragma unroll
        for (int j = 0; j < TN; ++j)
            if (active && lane < 32) {
                red[(wave * WN + j * 32 + lane) * 2 + 0] = st_s[j];
                red[(wave * WN + j * 32 + lane) * 2 + 1] = st_q[j];
            }
        __syncthreads();
        for (int c = active ? tid : BN; c < BN; c += 256) {
            const int wcol = c / WN, cc = c - wcol * WN;      // waves (0,wcol) and (1,wcol) own this column
            const float s = red[((0 * 2 + wcol) * WN + cc) * 2 + 0] + red[((1 * 2 + wcol) * WN + cc) * 2 + 0];
            const float q = red[((0 * 2 + wcol) * WN + cc) * 2 + 1] + red[((1 * 2 + wcol) * WN + cc) * 2 + 1];
            if (n0 + c < p.Co) {
                p.stats[((long)m_tile * 2 + 0) * p.Co + n0 + c] = s;
                p.stats[((long)m_tile * 2 + 1) * p.Co + n0 + c] = q;
            }
        }
    }
}

template <int BM, int BN, int TM, int TN>
__device__ __forceinline__ void igemm_mma_tile(const bf16_t* cA, const bf16_t* cB, f32x16 (&acc)[TM][TN], int wm0, int wn0, int lane) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int ch = ks * 2 + (lane >> 5);
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm0 + i * 32 + (lane & 31);
            fa[i] = *reinterpret_cast<const bf16x8*>(cA + LDS_SWZ(row, ch));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = wn0 + j * 32 + (lane & 31);
            fb[j] = *reinterpret_cast<const bf16x8*>(cB + LDS_SWZ(row, ch));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA pipeline (default)
// ---------------------------------------------------------------------------------------------------------------------
// s_waitcnt vmcnt(LPT * later) + s_barrier with a compile-time immediate for every possible `later` in [0, MAXL]
template <int LPT, int MAXL>
__device__ __forceinline__ void wait_tiles_barrier(int later) {
    if constexpr (MAXL > 0) {
        if (later == MAXL) { SVSR_WAIT_VM_BARRIER(LPT * MAXL); return; }
        wait_tiles_barrier<LPT, MAXL - 1>(later);
    } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// m -> (image, index inside the class), exact for m < 2^24 rows
__device__ __forceinline__ void split_row(int m, int P, float inv_p, int& n, int& j) {
    n = (int)((float)m * inv_p);
    j = m - n * P;
    if (j < 0) { n--; j += P; } else if (j >= P) { n++; j -= P; }
}

// KG = 2: in-workgroup split of the contraction.  Two groups of four waves work on the SAME output tile, each with its own LDS ring
// over one half of the K steps; the second group's accumulators are added to the first's through LDS (always in that order) before
// the epilogue.  For launches with far fewer tiles than CUs and a long K (the encoder's 960-row linears with 512 outputs: 120
// tiles, 24-40 K steps) this halves the serial K loop of the few workgroups there are.
template <int BM, int BN, int NS, int KG = 1>
__global__ __launch_bounds__(256 * KG) void k_igemm_fwd_glds(const IgemmFwdArgs p) {
    constexpr int BK = 64;
    static_assert(NS >= 2 && (BM / 32 + BN / 32) * (NS - 2) <= 63, "vmcnt immediate is 6 bits");
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, S_ELEMS = A_ELEMS + B_ELEMS;
    constexpr int AR = BM / 32, BR = BN / 32, LPT = AR + BR;     // DMA instructions per thread per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int grp = KG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    bf16_t* sStage = reinterpret_cast<bf16_t*>(smem_raw) + grp * NS * S_ELEMS;      // this group's ring [NS][A | B]
    long* sRow = reinterpret_cast<long*>(reinterpret_cast<bf16_t*>(smem_raw) + KG * NS * S_ELEMS);
    int* sTap = reinterpret_cast<int*>(sRow + BM);                        // delta[9], tw[9]

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    const int bx = blockIdx.x, by = blockIdx.y;
    const int n0 = by * BN;
    const int slot = tid & 7, r0 = tid >> 3;                 // lane writes LDS chunk `slot` of row r0 + 32*i ...
    const int csw = slot ^ ((r0 >> 1) & 7);                  // ... which must hold global chunk csw (swizzle on the source)
    const int wrow = __builtin_amdgcn_readfirstlane(wave) * 8;   // first row of this wave's 8-row group

    // this block's class (wave-uniform scalar loads; tile_begin is ascending)
    // words[0] = classes | identity << 16: identity = the one class maps row m to source row m and target row m (a plain dense layer):
    // no position table to fetch — one dependent global round trip less in front of the first DMA of these latency-bound launches
    const int hdr0 = p.plan[0];
    const int ncls = hdr0 & 0xffff;
    const bool identity = (hdr0 >> 16) != 0;
    int cls = 0;
    for (int c = 1; c < ncls; ++c)
        if (bx >= p.plan[PLAN_HDR_WORDS + c * PLAN_CLS_WORDS + 3]) cls = c;
    const int* cw = p.plan + PLAN_HDR_WORDS + cls * PLAN_CLS_WORDS;
    const int ntaps = cw[0], P = cw[1];
    const int* pos = p.plan + p.plan[1] + 2 * cw[2];
    const int m0 = (bx - cw[3]) * BM;
    const int Mc = p.Nimg * P;
    const float inv_p = 1.0f / (float)P;
    if (tid < 18) sTap[tid] = cw[4 + tid];                 // (both K groups write the same values)

    // Everything that depends on the row is hoisted out of the K loop: a pointer to the row's centre pixel (rows beyond the
    // class read the zero page for every tap).  A K step then costs one 64-bit add per DMA.
    const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(g_zero_page) + slot * 8;
    const bf16_t* a_ptr[AR];
    unsigned a_ok = 0;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + r0 + 32 * i;
        const bool ok = m < Mc;
        if (identity) {
            a_ptr[i] = p.in + (long)(ok ? m : 0) * p.in_pix * p.in_pitch + csw * 8;
        } else {
            int n, j;
            split_row(ok ? m : 0, P, inv_p, n, j);
            a_ptr[i] = p.in + ((long)n * p.in_pix + pos[2 * j]) * p.in_pitch + csw * 8;
        }
        a_ok |= (ok ? 1u : 0u) << i;
    }
    const bf16_t* b_ptr[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + r0 + 32 * i;
        b_ptr[i] = n < p.Co ? p.wt + (long)n * p.wt_taps * p.Ci + csw * 8 : nullptr;
    }
    for (int r = tid; r < BM; r += 256) {          // target pixel offsets for the epilogue
        const int m = m0 + r;
        long off = -1;
        if (m < Mc) {
            if (identity) {
                off = (long)m * p.out_pix * p.out_pitch;
            } else {
                int n, j;
                split_row(m, P, inv_p, n, j);
                off = ((long)n * p.out_pix + pos[2 * j + 1]) * p.out_pitch;
            }
        }
        sRow[r] = off;
    }
    __syncthreads();

    const int KT_all = ntaps * (p.Ci / BK);
    // K steps of this group: [k_begin, k_begin + KT); the loop below runs KT_loop iterations in every group so that the groups
    // meet at the same barriers (a group that is one step short idles through its last iteration)
    const int KT_loop = (KT_all + KG - 1) / KG;
    const int k_begin = grp * KT_loop;
    const int KT = KT_all - k_begin < KT_loop ? (KT_all - k_begin > 0 ? KT_all - k_begin : 0) : KT_loop;
    const int spt = p.Ci / BK;                                           // K steps per tap
    int t_next = k_begin / spt, c_next = (k_begin - (k_begin / spt) * spt) * BK;
    auto stage = [&](int buf) {
        const int tw = sTap[9 + t_next];
        const int c0 = c_next;
        const long a_off = (long)sTap[t_next] * p.in_pitch + c0;     // wave-uniform
        bf16_t* dstA = sStage + buf * S_ELEMS + wrow * 64;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const bf16_t* src = ((a_ok >> i) & 1u) ? a_ptr[i] + a_off : zero_src;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dstA + i * 32 * 64), 16, 0, 0);
        }
        bf16_t* dstB = sStage + buf * S_ELEMS + A_ELEMS + wrow * 64;
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const bf16_t* src = b_ptr[i] != nullptr ? b_ptr[i] + (long)tw * p.Ci + c0 : zero_src;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dstB + i * 32 * 64), 16, 0, 0);
        }
        c_next += BK;
        if (c_next >= p.Ci) { c_next = 0; ++t_next; }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: NS-1 tiles in flight (the LDS-DMA round trip is ~1 us: the loop is bound by latency / tiles in flight)
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < KT) stage(s);
    int buf = 0;
    for (int it = 0; it < KT_loop; ++it) {
        // tile `it` has landed once at most min(NS-2, tiles left) later tiles' DMAs are still outstanding (vmcnt retires in
        // order); the barrier makes every wave's part visible and proves everybody is done reading the buffer the next
        // stage() overwrites.
        int later = KT - 1 - it < NS - 2 ? KT - 1 - it : NS - 2;
        if (later < 0) later = 0;
        wait_tiles_barrier<LPT, NS - 2>(later);
        if (it + NS - 1 < KT) stage(buf >= 1 ? buf - 1 : NS - 1);   // == (it + NS - 1) % NS
        if (KG == 1 || it < KT) {
            const bf16_t* cA = sStage + buf * S_ELEMS;
            igemm_mma_tile<BM, BN, TM, TN>(cA, cA + A_ELEMS, acc, wm0, wn0, lane);
        }
        buf = buf + 1 == NS ? 0 : buf + 1;
    }
    __syncthreads();
    if constexpr (KG == 2) {
        // group 1 -> LDS (its own, now dead, ring) -> group 0: acc0 + acc1, one fixed order
        float* sX = reinterpret_cast<float*>(reinterpret_cast<bf16_t*>(smem_raw) + NS * S_ELEMS);      // [TM*TN*16][256]
        static_assert((size_t)TM * TN * 16 * 256 * sizeof(float) <= (size_t)NS * S_ELEMS * sizeof(bf16_t), "exchange buffer fits the ring");
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sX[((i * TN + j) * 16 + r) * 256 + tid] = acc[i][j][r];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += sX[((i * TN + j) * 16 + r) * 256 + tid];
        }
        __syncthreads();
    }
    igemm_epilogue<BM, BN, TM, TN>(p, acc, smem_raw, sRow, wm0, wn0, n0, tid, grp == 0, bx);
}

// ---------------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int NS, int KG = 1>
static int launch_glds(const IgemmFwdArgs& a, int gx, int gy, hipStream_t stream) {
    const size_t lds = (size_t)KG * NS * (BM + BN) * 64 * sizeof(bf16_t) + (size_t)BM * sizeof(long) + 128 + (size_t)svsr_tune_get(SVSR_TUNE_IGEMM_LDS_PAD);
    static size_t attr_set = 0;
    if (attr_set < lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_igemm_fwd_glds<BM, BN, NS, KG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = lds;
    }
    IgemmFwdArgs b = a;
    hipLaunchKernelGGL((k_igemm_fwd_glds<BM, BN, NS, KG>), dim3(gx, gy), dim3(256 * KG), lds, stream, b);
    return svsr_check_launch();
}

// Tile / ring-depth choice.  One K step of a block costs max(MFMA time, LDS-DMA issue + round trip / tiles in flight):
// with at least ~2 blocks per CU a shallow ring and several co-resident blocks is best.  Stage counts: 128x128 -> 2,
// 128x64 -> 3 (2 for the few-tile convolutions), 64x64 -> 4 (3 if many blocks).
struct IgemmFwdPlan { int bm, bn, ns, gy; };

static IgemmFwdPlan igemm_fwd_plan(long M, int Co, int max_taps) {
    const int cus = svsr_stream_cus(nullptr);
    IgemmFwdPlan pl;
    const int forced = svsr_tune_get(SVSR_TUNE_IGEMM_TILE);            // 0 auto, 64 / 128 forced
    const int thr = svsr_tune_get(SVSR_TUNE_IGEMM_M128);
    int bm;
    if (forced == 64 || forced == 128) bm = forced;
    else if (Co <= 64) bm = M >= 16384 ? 128 : 64;
    else {
        // 128x128 tiles once they still give ~a block per CU (LRS linears at 2,400 rows: qkv/ffn1/heads yes, 768-wide outputs no)
        const long blocks128 = ((M + 127) / 128) * ((Co + 127) / 128);
        bm = (M >= thr || blocks128 >= 224) ? 128 : 64;
    }
    // few-tile convolutions (LRW layer4: 66 x 4 tiles of 128x128 = about one workgroup per CU): 128x64 tiles with a 2-deep ring
    // fit three workgroups per CU and measure 82 -> 77 us; with more tiles the 128x128 shape wins (layer2 61 vs 66 us)
    // single-tap launches (linears) too small for 128x128 tiles but with many rows (LRS: 2,400 x 768-wide outputs): 128x64 tiles, 3-deep ring
    const int lin_rows = svsr_tune_get(SVSR_TUNE_IGEMM_LIN_BN64);
    if (bm == 64 && forced == 0 && Co > 64 && max_taps == 1 && lin_rows > 0 && M >= lin_rows) { pl.bm = 128; pl.bn = 64; pl.ns = 3; }
    else if (bm == 128 && Co > 64 && max_taps > 1 && ((M + 127) / 128) * ((Co + 127) / 128) < svsr_tune_get(SVSR_TUNE_IGEMM_BN64_BELOW)) { pl.bm = 128; pl.bn = 64; pl.ns = 2; }
    // 64 output channels, at most four taps (the stride-2 data gradients into layer1: K loops of 2-8 steps, 3,510 tiles whose time is the dependent
    // round trips of prologue and epilogue): a 2-deep ring fits three workgroups per CU instead of two — 61.3 -> 51.6 us at 928 frames
    else if (bm == 128 && Co <= 64) { pl.bm = 128; pl.bn = 64; pl.ns = max_taps <= 4 ? 2 : 3; }
    else if (bm == 128) { pl.bm = 128; pl.bn = 128; pl.ns = 2; }
    else { pl.bm = 64; pl.bn = 64; pl.ns = 0; }
    pl.gy = (Co + pl.bn - 1) / pl.bn;
    if (pl.ns == 0) pl.ns = ((M + pl.bm - 1) / pl.bm) * pl.gy <= (long)cus * 5 / 2 ? 4 : 3;
    if (pl.bm == 64 && pl.bn == 64 && (svsr_tune_get(SVSR_TUNE_IGEMM_NS64) == 6 || svsr_tune_get(SVSR_TUNE_IGEMM_NS64) == 8)) pl.ns = svsr_tune_get(SVSR_TUNE_IGEMM_NS64);
    return pl;
}

// ---- host-side plan builder ------------------------------------------------------------------------------------------
struct PlanClass { int ntaps; int delta[9], tw[9]; std::vector<int> pos; };     // pos: (src, dst) pairs

// meta: {bm, bn, ns, tiles (= grid.x = BatchNorm partial rows), grid.y, classes, max taps of a class, total rows / 2^0 (low 31 bits)}
// p8 format (igemm_fwd.h): per 256-row M tile a descriptor and a row table of global pixel indices, tiles in class order
static int plan_emit_p8(const std::vector<PlanClass>& cls, int Nimg, int Co, int in_pix, int out_pix, int* words, int cap_words, int* meta, long M, int max_taps, int bn) {
    long tiles_m = 0;
    for (const PlanClass& c : cls) tiles_m += ((long)Nimg * (long)(c.pos.size() / 2) + P8_BM - 1) / P8_BM;
    const int gy = Co / bn;
    const long nwords = P8_HDR_WORDS + tiles_m * (P8_DESC_WORDS + 2 * P8_BM);
    if (nwords > 0x7fffffffL || (long)Nimg * in_pix >= (1L << 31) || (long)Nimg * out_pix >= (1L << 31)) return -SVSR_ERR_ARG;
    if (words != nullptr) {
        if (cap_words < nwords) return -SVSR_ERR_ARG;
        const int off_desc = P8_HDR_WORDS, off_rows = P8_HDR_WORDS + (int)tiles_m * P8_DESC_WORDS;
        words[0] = P8_MAGIC; words[1] = (int)tiles_m; words[2] = gy; words[3] = off_desc; words[4] = off_rows;
        words[5] = (int)((tiles_m + 7) / 8 * 8 * gy); words[6] = 0; words[7] = 0;
        long t = 0;
        for (const PlanClass& c : cls) {
            const long P = (long)(c.pos.size() / 2), Mc = (long)Nimg * P;
            for (long m0 = 0; m0 < Mc; m0 += P8_BM, ++t) {
                int* d = words + off_desc + t * P8_DESC_WORDS;
                for (int k = 0; k < P8_DESC_WORDS; ++k) d[k] = 0;
                d[0] = c.ntaps; d[1] = (int)(Mc - m0 < P8_BM ? Mc - m0 : P8_BM);
                for (int k = 0; k < c.ntaps; ++k) { d[2 + k] = c.delta[k]; d[11 + k] = c.tw[k]; }
                int* rt = words + off_rows + t * 2 * P8_BM;
                for (int rr = 0; rr < P8_BM; ++rr) {
                    const long m = m0 + rr;
                    if (m < Mc) {
                        const long n = m / P, j = m - n * P;
                        rt[2 * rr] = (int)(n * in_pix + c.pos[2 * j]);
                        rt[2 * rr + 1] = (int)(n * out_pix + c.pos[2 * j + 1]);
                    } else { rt[2 * rr] = rt[0]; rt[2 * rr + 1] = -1; }       // a readable source (the tile's first row); no target
                }
            }
        }
    }
    if (meta != nullptr) {
        meta[0] = P8_BM; meta[1] = bn; meta[2] = 3; meta[3] = (int)tiles_m; meta[4] = gy; meta[5] = (int)cls.size(); meta[6] = max_taps;
        meta[7] = (int)(M & 0x7fffffff);
    }
    return (int)nwords;
}

// p8_pix: {source pixels per image, target pixels per image} when the caller's shape qualifies for the persistent kernel, else null
static int plan_emit(std::vector<PlanClass>& cls, int Nimg, int Co, int* words, int cap_words, int* meta, const int* p8_pix = nullptr) {
    std::stable_sort(cls.begin(), cls.end(), [](const PlanClass& a, const PlanClass& b) { return a.ntaps > b.ntaps; });
    long M = 0;
    int max_taps = 0;
    for (const PlanClass& c : cls) { M += (long)Nimg * (long)(c.pos.size() / 2); if (c.ntaps > max_taps) max_taps = c.ntaps; }
    if (M >= (1L << 24) * 64 || cls.empty()) return -SVSR_ERR_ARG;
    if (p8_pix != nullptr && svsr_tune_get(SVSR_TUNE_P8) && Co % 64 == 0) {
        long tiles_m = 0;
        int min_taps = 9;
        for (const PlanClass& c : cls) { tiles_m += ((long)Nimg * (long)(c.pos.size() / 2) + P8_BM - 1) / P8_BM; if (c.ntaps < min_taps) min_taps = c.ntaps; }
        // (the kernel's pipeline needs >= 4 K tiles per tile: every class of a padded 3x3 convolution has >= 4 taps)
        const int min_items = svsr_tune_get(SVSR_TUNE_P8_MIN_ITEMS);
        if (min_taps >= 4 && Co % P8_BN == 0 && tiles_m * (Co / P8_BN) >= min_items)
            return plan_emit_p8(cls, Nimg, Co, p8_pix[0], p8_pix[1], words, cap_words, meta, M, max_taps, P8_BN);
        // too few 256 x 128 items to give every CU one (layer4: 33 row tiles x 4): 256 x 64 tiles (tune key p8_bn64)
        if (min_taps >= 4 && Co >= 128 && tiles_m * (Co / 64) >= min_items && svsr_tune_get(SVSR_TUNE_P8_BN64))
            return plan_emit_p8(cls, Nimg, Co, p8_pix[0], p8_pix[1], words, cap_words, meta, M, max_taps, 64);
    }
    const IgemmFwdPlan pl = igemm_fwd_plan(M, Co, max_taps);
    const int ncls = (int)cls.size();
    int nwords = PLAN_HDR_WORDS + ncls * PLAN_CLS_WORDS;
    const int pos_word0 = nwords;
    for (const PlanClass& c : cls) nwords += (int)c.pos.size();
    long tiles = 0;
    if (words != nullptr) {
        if (cap_words < nwords) return -SVSR_ERR_ARG;
        words[0] = ncls; words[1] = pos_word0;
        int pos_off = 0;
        for (int i = 0; i < ncls; ++i) {
            const PlanClass& c = cls[i];
            int* w = words + PLAN_HDR_WORDS + i * PLAN_CLS_WORDS;
            const int P = (int)(c.pos.size() / 2);
            if ((long)Nimg * P >= (1L << 24)) return -SVSR_ERR_ARG;          // split_row's float reciprocal is exact below 2^24 rows
            w[0] = c.ntaps; w[1] = P; w[2] = pos_off; w[3] = (int)tiles;
            for (int t = 0; t < 9; ++t) { w[4 + t] = t < c.ntaps ? c.delta[t] : 0; w[13 + t] = t < c.ntaps ? c.tw[t] : 0; }
            w[22] = 0; w[23] = 0;
            std::copy(c.pos.begin(), c.pos.end(), words + pos_word0 + 2 * pos_off);
            pos_off += P;
            tiles += ((long)Nimg * P + pl.bm - 1) / pl.bm;
        }
    } else {
        for (const PlanClass& c : cls) tiles += ((long)Nimg * (long)(c.pos.size() / 2) + pl.bm - 1) / pl.bm;
    }
    if (tiles > 0x7fffffffL) return -SVSR_ERR_ARG;
    if (meta != nullptr) {
        meta[0] = pl.bm; meta[1] = pl.bn; meta[2] = pl.ns; meta[3] = (int)tiles; meta[4] = pl.gy; meta[5] = ncls; meta[6] = max_taps;
        meta[7] = (int)(M & 0x7fffffff);
    }
    return nwords;
}

/* svsr_conv_plan (host): launch plan of a k x k / stride / pad convolution over Nimg images whose FORWARD input is H x W.
 *   mode 0: forward        in = x  [H][W],   out = y  [Ho][Wo]      (weights [Co][k*k][Ci])
 *   mode 1: data gradient  in = dy [Ho][Wo], out = dx [H][W]        (transposed weights [Ci][k*k][Co]); every pixel of dx is
 *           written, also those no tap reaches (k < stride): they get 0 (+ bias / addend)
 *   mode 2: as mode 1 for an IN-PLACE accumulation (addend aliases out): pixels no tap reaches are left alone
 * Co_out = output channels of the launch (tile choice).  words == null: only counts.  Returns the number of int32 words, or a
 * negative error.  meta[8] = {bm, bn, ns, tiles, grid_y, classes, max taps, rows}. */
extern "C" int svsr_conv_plan(int mode, int Nimg, int H, int W, int Co_out, int k, int stride, int pad, int* words, int cap_words, int* meta) {
    if (Nimg < 1 || H < 1 || W < 1 || k < 1 || k > 3 || stride < 1 || pad < 0 || Co_out < 1 || mode < 0 || mode > 2) return -SVSR_ERR_ARG;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (Ho < 1 || Wo < 1) return -SVSR_ERR_ARG;
    std::vector<PlanClass> cls;
    std::map<std::vector<int>, int> index;      // key: the tap list (delta, tw)* of a position -> class
    auto add = [&](const std::vector<int>& key, int src, int dst) {
        auto it = index.find(key);
        if (it == index.end()) {
            PlanClass c;
            c.ntaps = (int)(key.size() / 2);
            for (int t = 0; t < c.ntaps; ++t) { c.delta[t] = key[2 * t]; c.tw[t] = key[2 * t + 1]; }
            cls.push_back(c);
            it = index.emplace(key, (int)cls.size() - 1).first;
        }
        cls[it->second].pos.push_back(src);
        cls[it->second].pos.push_back(dst);
    };
    std::vector<int> key;
    if (mode == 0) {
        for (int a = 0; a < Ho; ++a)
            for (int b = 0; b < Wo; ++b) {
                key.clear();
                for (int kh = 0; kh < k; ++kh)
                    for (int kw = 0; kw < k; ++kw) {
                        const int iy = a * stride + kh - pad, ix = b * stride + kw - pad;
                        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                        key.push_back((kh - pad) * W + (kw - pad));
                        key.push_back(kh * k + kw);
                    }
                // centre pixel (a*stride, b*stride) may itself lie outside the grid only through the taps; deltas are relative to it
                add(key, (a * stride) * W + b * stride, a * Wo + b);
            }
    } else {
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                key.clear();
                const int a = y / stride, b = x / stride, py = y % stride, px = x % stride;
                for (int kh = 0; kh < k; ++kh) {
                    if ((py + pad - kh) % stride) continue;
                    for (int kw = 0; kw < k; ++kw) {
                        if ((px + pad - kw) % stride) continue;
                        // floor division: (py + pad - kh) may be negative but is a multiple of stride
                        const int dq = (py + pad - kh) / stride, dp = (px + pad - kw) / stride;
                        const int oy = a + dq, ox = b + dp;
                        if (oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
                        key.push_back(dq * Wo + dp);
                        key.push_back(kh * k + kw);
                    }
                }
                // the centre (a, b) of dy may lie outside its grid (odd H with stride 2): clamp it and fold the shift into nothing —
                // positions whose centre is out of range get their own key via the deltas being relative to the clamped centre
                int ca = a < Ho ? a : Ho - 1, cb = b < Wo ? b : Wo - 1;
                if (ca != a || cb != b) {
                    const int shift = (a - ca) * Wo + (b - cb);
                    for (size_t t = 0; t < key.size(); t += 2) key[t] += shift;
                }
                if (mode == 2 && key.empty()) continue;
                add(key, ca * Wo + cb, y * W + x);
            }
    }
    // stride-1 3x3 / pad 1 (forward and data gradient alike: same map size on both sides): candidates for the persistent kernel
    // (tune key p8 = 3: stride 1 only)
    const int p8_pix[2] = {mode == 0 ? H * W : Ho * Wo, mode == 0 ? Ho * Wo : H * W};
    // (and the FORWARD of the stride-2 3x3 convolutions that open layer2-4: the same tap classes over a quarter of the positions; their data
    // gradients have parity classes of one or two taps, too short for the kernel's pipeline)
    const bool p8_ok = k == 3 && pad == 1 && H >= 2 && W >= 2 && (stride == 1 || (stride == 2 && mode == 0 && svsr_tune_get(SVSR_TUNE_P8) >= 1 && svsr_tune_get(SVSR_TUNE_P8) != 3));
    return plan_emit(cls, Nimg, Co_out, words, cap_words, meta, p8_ok ? p8_pix : nullptr);
}

/* svsr_rows_plan (host): plan of a dense layer over rows grouped in Nimg sequences: row (n, j), j < P, reads source row
 * n * in_pix + src0 + j and writes target row n * out_pix + dst0 + j (one tap).  A plain linear layer is P = 1, src0 = dst0 = 0
 * with in_pix = out_pix = 1; picking / scattering a slice of every sequence uses P = slice length. */
extern "C" int svsr_rows_plan(int Nimg, int P, int src0, int dst0, int Co_out, int* words, int cap_words, int* meta) {
    if (Nimg < 1 || P < 1 || Co_out < 1) return -SVSR_ERR_ARG;
    std::vector<PlanClass> cls(1);
    cls[0].ntaps = 1; cls[0].delta[0] = 0; cls[0].tw[0] = 0;
    cls[0].pos.reserve(2 * (size_t)P);
    for (int j = 0; j < P; ++j) { cls[0].pos.push_back(src0 + j); cls[0].pos.push_back(dst0 + j); }
    const int n = plan_emit(cls, Nimg, Co_out, words, cap_words, meta);
    if (n > 0 && words != nullptr && P == 1 && src0 == 0 && dst0 == 0) words[0] |= 1 << 16;      // identity rows: see k_igemm_fwd_glds
    return n;
}

/* K groups of the instantiation svsr_igemm_fwd / svsr_igemm_dgrad_bn will launch for this plan: 2 = k_igemm_fwd_glds<64,64,4,2> (the
 * contraction split over two wave groups of one workgroup: few 64x64 tiles, >= 12 K steps), else 1.  Host-side query (bench.py labels
 * its per-kernel table with it, so the table's names are the profiler's). */
extern "C" int svsr_igemm_fwd_kgroups(const int* meta, int Ci, int Co, int bn_epilogue) {
    if (meta == nullptr || Ci < 64) return 1;
    const int bm = meta[0], bn = meta[1], gx = meta[3], gy = (Co + bn - 1) / bn;
    const int ks = svsr_tune_get(SVSR_TUNE_IGEMM_KSPLIT);
    if (!bn_epilogue && bm == 64 && bn == 64 && ks && (long)gx * gy <= ks && (long)meta[6] * (Ci / 64) >= 12) return 2;
    // dense layers on 128 x 64 tiles that give about ONE four-wave workgroup per CU (2,560 rows x 768 columns = 240 tiles) and a long
    // contraction: a single wave per SIMD cannot hide its own LDS-DMA round trips (K = 3,072: 29.2 us, 413 TFLOP/s against hipBLASLt's
    // 20.2); two wave groups halve every workgroup's K loop and put two waves on every SIMD
    const int ks128 = svsr_tune_get(SVSR_TUNE_IGEMM_KSPLIT128);
    if (!bn_epilogue && bm == 128 && bn == 64 && meta[2] == 3 && meta[6] == 1 && ks128 > 0 && Ci / 64 >= ks128 && (long)gx * gy <= 288) return 2;
    return 1;
}

/* svsr_igemm_fwd: runs a plan.  plan_dev = device copy of the words, meta = the host meta[8] svsr_*_plan returned with them. */
static int igemm_fwd_run(const void* in, const void* wt, void* out, void* out_pre, const float* bias, const void* addend,
                         float* stats, const int* plan_dev, const int* meta, int Nimg, int in_pix, int Ci, int in_pitch, int Co,
                         int out_pix, int out_pitch, int wt_taps, int act, int out_f32, float alpha, const unsigned* drop_seed,
                         unsigned drop_site, float drop_p, const void* bnb_y, const void* bnb_x, const float* bnb_mean,
                         const float* bnb_rstd, const float* bnb_gamma, const float* bnb_beta, int bnb_act, hipStream_t stream) {
    if (plan_dev == nullptr || meta == nullptr || Ci % 64 != 0 || Ci <= 0 || Co <= 0 || in_pitch % 8 != 0 || Nimg <= 0 || wt_taps < 1)
        return SVSR_ERR_ARG;
    if (act != 0 && (addend != nullptr || alpha != 1.f)) return SVSR_ERR_ARG;     // the activation is applied before alpha / addend
    IgemmFwdArgs a;
    a.in = (const bf16_t*)in; a.wt = (const bf16_t*)wt; a.out = out; a.out_pre = (bf16_t*)out_pre;
    a.bias = bias; a.addend = (const bf16_t*)addend; a.stats = stats; a.plan = plan_dev;
    a.Nimg = Nimg; a.in_pix = in_pix; a.Ci = Ci; a.in_pitch = in_pitch; a.Co = Co; a.out_pix = out_pix; a.out_pitch = out_pitch;
    a.wt_taps = wt_taps; a.act = act; a.out_f32 = out_f32; a.alpha = alpha;
    a.drop = svsr_make_drop(drop_seed, drop_site, drop_p);
    a.epi_batched = svsr_tune_get(SVSR_TUNE_EPI_BATCHED);
    a.bnb_y = (const bf16_t*)bnb_y; a.bnb_x = (const bf16_t*)bnb_x; a.bnb_mean = bnb_mean; a.bnb_rstd = bnb_rstd;
    a.bnb_gamma = bnb_gamma; a.bnb_beta = bnb_beta; a.bnb_act = bnb_act;
    const int bm = meta[0], bn = meta[1], ns = meta[2], gx = meta[3], gy = (Co + bn - 1) / bn;
    if (gx < 1) return SVSR_ERR_ARG;
    if (bm == P8_BM) return (bnb_x != nullptr && alpha != 1.f) ? SVSR_ERR_ARG : igemm_p8_launch(a, meta, stream);      // (the persistent kernel's BatchNorm-backward epilogue has no scale)
    // few tiles, long contraction: split K inside the workgroup (see k_igemm_fwd_glds, KG = 2)
    if (svsr_igemm_fwd_kgroups(meta, Ci, Co, bnb_x != nullptr) == 2) {
        if (bm == 128) return launch_glds<128, 64, 3, 2>(a, gx, gy, stream);
        return launch_glds<64, 64, 4, 2>(a, gx, gy, stream);
    }
#define SVSR_IGEMM_CASE(BM_, BN_, NS_) if (bm == BM_ && bn == BN_ && ns == NS_) return launch_glds<BM_, BN_, NS_>(a, gx, gy, stream)
    SVSR_IGEMM_CASE(128, 128, 2);
    SVSR_IGEMM_CASE(128, 64, 2);
    SVSR_IGEMM_CASE(128, 64, 3);
    SVSR_IGEMM_CASE(64, 64, 4);
    SVSR_IGEMM_CASE(64, 64, 3);
    SVSR_IGEMM_CASE(64, 64, 6);
    SVSR_IGEMM_CASE(64, 64, 8);
#undef SVSR_IGEMM_CASE
    return SVSR_ERR_ARG;
}

extern "C" int svsr_igemm_fwd(const void* in, const void* wt, void* out, void* out_pre, const float* bias, const void* addend,
                              float* stats, const int* plan_dev, const int* meta, int Nimg, int in_pix, int Ci, int in_pitch, int Co,
                              int out_pix, int out_pitch, int wt_taps, int act, int out_f32, float alpha, const unsigned* drop_seed,
                              unsigned drop_site, float drop_p, hipStream_t stream) {
    return igemm_fwd_run(in, wt, out, out_pre, bias, addend, stats, plan_dev, meta, Nimg, in_pix, Ci, in_pitch, Co, out_pix, out_pitch,
                         wt_taps, act, out_f32, alpha, drop_seed, drop_site, drop_p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

/* svsr_igemm_dgrad_relu: the data gradient of a linear layer / convolution whose INPUT was y = dropout(relu(z)) — the feed-forward blocks of the
 * Conformer and of the decoder (positionwise_feed_forward.py:28-30: w_2(dropout(relu(w_1 x)))).  The launch stores
 * dz = (y > 0 ? gscale * dL/dy : 0) instead of dL/dy (y: the saved activation, zero where ReLU or the dropout mask cut; gscale = 1 / (1 - p)) and
 * writes, per row tile, the column sums of dz into stats[meta[3]][2][Co] (first half of every row: the partial rows of the bias gradient of
 * w_1; the second half is scratch) — what svsr_bias_act_bwd did in a launch of its own behind the data gradient.  It is the BatchNorm-backward
 * epilogue of svsr_igemm_dgrad_bn run with mean 0, rstd 1, gamma 1, beta 0: zeros / ones are device vectors of Co floats holding those values. */
extern "C" int svsr_igemm_dgrad_relu(const void* in, const void* wt, void* out, float* stats, const int* plan_dev, const int* meta, int Nimg,
                                     int in_pix, int Ci, int in_pitch, int Co, int out_pix, int out_pitch, int wt_taps, const void* y,
                                     const float* zeros, const float* ones, float gscale, hipStream_t stream) {
    if (y == nullptr || zeros == nullptr || ones == nullptr || stats == nullptr || Co % 8 != 0 || out_pitch % 8 != 0 || !(gscale > 0.f)) return SVSR_ERR_ARG;
    return igemm_fwd_run(in, wt, out, nullptr, nullptr, nullptr, stats, plan_dev, meta, Nimg, in_pix, Ci, in_pitch, Co, out_pix, out_pitch,
                         wt_taps, 0, 0, gscale, nullptr, 0, 0.f, nullptr, y, zeros, ones, ones, zeros, 1, stream);
}

/* svsr_igemm_dgrad_bn: a data-gradient plan (every target pixel visited exactly once: svsr_conv_plan mode 1) whose result dL/dy is the
 * gradient of a BatchNorm + ReLU output y = relu(bn(x) [+ residual]) with the geometry of `out` (reference tcn/models/resnet.py:59-72
 * backward).  The launch stores g = (y > 0 ? dL/dy [+ addend] : 0) instead of dL/dy and writes, per row tile, the column sums of g and of
 * g * (x - mean) * rstd into stats[meta[3]][2][Co] — the first pass of the BatchNorm backward, taken while the tile is in registers;
 * svsr_bn_bwd_from_stats finishes it.  addend may alias out.  y == nullptr (only for an output WITHOUT residual branch): the mask is
 * recomputed from x as bn(x) > 0 with gamma / beta, in the forward pass's own arithmetic, and y is not read.
 * act = 2 (Swish, the LRS trunk and Conformer convolution module): g = dL/dy * swish'(bn(x) + r) with `y` = the residual INPUT r of the
 * output (null: none); gamma / beta are required. */
extern "C" int svsr_igemm_dgrad_bn(const void* in, const void* wt, void* out, const void* addend, float* stats, const int* plan_dev,
                                   const int* meta, int Nimg, int in_pix, int Ci, int in_pitch, int Co, int out_pix, int out_pitch,
                                   int wt_taps, const void* y, const void* x, const float* mean, const float* rstd, const float* gamma,
                                   const float* beta, int act, hipStream_t stream) {
    if (x == nullptr || mean == nullptr || rstd == nullptr || stats == nullptr || Co % 8 != 0 || out_pitch % 8 != 0) return SVSR_ERR_ARG;
    if (act != 1 && act != 2) return SVSR_ERR_ARG;
    if ((y == nullptr || act == 2) && (gamma == nullptr || beta == nullptr)) return SVSR_ERR_ARG;
    return igemm_fwd_run(in, wt, out, nullptr, nullptr, addend, stats, plan_dev, meta, Nimg, in_pix, Ci, in_pitch, Co, out_pix, out_pitch,
                         wt_taps, 0, 0, 1.0f, nullptr, 0, 0.f, y, x, mean, rstd, gamma, beta, act, stream);
}
