// Weight-gradient contraction on MFMA (gfx950):
//     dW[co][tw[t]][ci] += sum_m dY[target(m)][co] * X[source(m, t)][ci]            (fp32)
// for every nn.Conv2d / nn.Linear of the path (autograd of reference LRW/video/src/tcn/models/resnet.py:8-16,59-72 and
// lightning.py:82,92,107).
//
// Rows come from a host-built PLAN (svsr_wgrad_plan): for every tap, the list of (source pixel, target pixel) pairs of the
// output positions whose source lies inside the grid — the reduction never runs over a convolution's zero padding (3x3 / pad 1
// on 3x3, 6x6, 11x11 maps: 40 %, 21 %, 12 % of the products of a padded formulation are zeros) and needs no masks.
//
// Grid: x = K split (ranges of 64-row chunks), y = (co tile, ci tile, tap).  Block = 4 waves (2x2), tile BC x BC channels of one
// tap.  Both operand tiles [64 rows][BC channels] go global -> LDS by direct DMA (global_load_lds_dwordx4) into an NS-deep ring
// behind a counted s_waitcnt vmcnt(N) + one s_barrier per chunk: the first version staged them through registers and was bound
// by the ds_write path (32 KiB of ds_write_b128 per chunk at ~79 B/clk/CU against 512 MFMA cycles).  The reduction index (rows)
// is the slow index of both operands, so the MFMA fragments (8 consecutive rows of one channel per lane) are produced by the
// gfx950 transpose read ds_read_b64_tr_b16.  A 32-lane group of that read touches 4 consecutive rows x 64 bytes; a DMA image is
// lane-linear (no row padding possible), so the 64-byte units of a row are XOR-swizzled with the row index — on the per-lane
// SOURCE address of the DMA and on the read address — which puts the four rows on the four bank quarters (conflict-free).
//
// Split-K without atomics: the workgroups of split s store their tiles into slab s of a caller-owned workspace with plain
// stores and svsr_colsum_rows (runtime.hip) adds the slabs into dW in a fixed order, so the gradient is reproducible; with a
// single split the tile is added to dW directly (one writer per element).  The bias gradient of an nn.Linear (column sums of dY)
// comes out of the same staged dY tiles through one extra MFMA per step against a fragment of ones.
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.h"
#include "../../include/syncvsr_hip.h"

#define WPLAN_HDR 2           // words[0] = taps, words[1] = word offset of the position table
#define WPLAN_TAP_WORDS 4     // per tap: { P, pos_off (pairs), tw, 0 }
#define WG_MAXP 1024          // positions per image and tap the LDS copy of the table can hold

struct WgradArgs {
    const bf16_t* x;       // source pixels (pitch in_pitch)
    const bf16_t* dy;      // target pixels (pitch out_pitch)
    float* dw;             // [Co][wt_taps][Ci] fp32, accumulated
    float* db;             // optional [Co] fp32: += column sums of dY (the bias gradient of an nn.Linear)
    float* part;           // splits > 1: slabs [splits][slab], slab = Co*wt_taps*Ci (+ Co when db != null) floats
    const int* plan;       // device plan words
    long slab;
    int splits, chunks_per_split;      // 64-row chunks per split (of the longest tap; shorter taps end early)
    int Nimg, in_pix, Ci, in_pitch, Co, out_pix, out_pitch, wt_taps;
};

__device__ unsigned g_wg_zero_page[64];     // 256 zero bytes: DMA source for rows beyond the end of a tap's row list

// row-major [64][BC] bf16 tile whose 64-byte units (32 channels) are XOR-swizzled by the row: element offset of (row, ch)
template <int BC>
__device__ __forceinline__ int wg_swz(int row, int ch) {
    constexpr int U = BC / 32;                       // 64-byte units per row: 4 (BC 128) or 2 (BC 64)
    const int f = U == 4 ? (row & 3) : ((row >> 1) & 1);
    return row * BC + ((((ch >> 5) ^ f) << 5) | (ch & 31));
}

// fragment for MFMA 32x32x16: lane l supplies matrix row (l&31) = channel ch0 + (l&31), k = (l>>5)*8 .. +7 = rows pos0 + ...
// 16-lane group gq: channels ch0 + (gq&1)*16 .., rows pos0 + (gq>>1)*8 ..; lane s of the group addresses row (s>>2) of a
// [4 rows][16 ch] block at channel sub-block (s&3)*4 and receives channel column s.  Rows r and r + 4 share their swizzle.
template <int BC>
__device__ __forceinline__ bf16x8 wg_frag_T(const bf16_t* tile, int ch0, int pos0, int lane) {
    const int gq = lane >> 4, s = lane & 15;
    const int row = pos0 + (gq >> 1) * 8 + (s >> 2), ch = ch0 + (gq & 1) * 16 + (s & 3) * 4;
    const bf16_t* base = tile + wg_swz<BC>(row, ch);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(base));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(base + 4 * BC));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

template <int LPT, int MAXL>
__device__ __forceinline__ void wg_wait_tiles_barrier(int later) {
    if constexpr (MAXL > 0) {
        if (later == MAXL) { SVSR_WAIT_VM_BARRIER(LPT * MAXL); return; }
        wg_wait_tiles_barrier<LPT, MAXL - 1>(later);
    } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// frag_dst != null (unit-list launches): the workgroup's tile goes out in the MFMA FRAGMENT layout — float4 group
// g = (((wave * TT + i) * TT + j) * 4 + rq) * 64 + lane holds accumulator registers rq*4 .. rq*4+3 of fragment (i, j) — as 16-byte
// stores of 1 KiB per wave-instruction (16 per wave and 128 x 128 tile instead of 64 dword stores of two 128-byte rows each); followed by
// the BC bias sums when BIAS.  k_wgrad_unit_reduce adds a task's partial tiles in slot order and scatters the sum into dW once.
// unit_begin / unit_count >= 0 replace the (split_idx, chunks_per_split) range.
template <int BC, int NS, bool BIAS>
__device__ __forceinline__ void wg_body(const WgradArgs& p, unsigned char* smem_raw, const int split_idx, const int task_idx,
                                        float* frag_dst = nullptr, const int unit_begin = -1, const int unit_count = 0) {
    constexpr int T_ELEMS = 64 * BC;                 // one operand tile
    constexpr int S_ELEMS = 2 * T_ELEMS;             // stage = dY tile | X tile
    constexpr int RPI = 1024 / (BC * 2);             // rows per DMA instruction (1 KiB): 4 (BC 128) or 8 (BC 64)
    constexpr int AR = 64 / RPI / 4;                 // DMA instructions per thread, operand and chunk: 4 (BC 128) or 2 (BC 64)
    constexpr int LPT = 2 * AR;
    constexpr int CPR = BC / 8;                      // 16-byte pieces per row
    constexpr int U = BC / 32;                       // 64-byte units per row
    constexpr int WT = BC / 2, TT = WT / 32;         // wave tile edge, 32x32 MFMA tiles per edge
    static_assert(NS >= 2 && LPT * (NS - 2) <= 63, "vmcnt immediate is 6 bits");
    bf16_t* sStage = reinterpret_cast<bf16_t*>(smem_raw);                 // [NS][dY | X]
    int* sPos = reinterpret_cast<int*>(sStage + NS * S_ELEMS);            // [P][2] (x pixel, dY pixel) of this tap

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co_tiles = (p.Co + BC - 1) / BC, ci_tiles = (p.Ci + BC - 1) / BC;
    int rest = task_idx;
    const int cot = rest % co_tiles; rest /= co_tiles;
    const int cit = rest % ci_tiles; rest /= ci_tiles;
    const int t = rest;
    const int co0 = cot * BC, ci0 = cit * BC;
    const int wco = (wave >> 1) * WT, wci = (wave & 1) * WT;
    const int* tapw = p.plan + WPLAN_HDR + t * WPLAN_TAP_WORDS;
    const int P = tapw[0], tw = tapw[2];
    const int* pos = p.plan + p.plan[1] + 2 * tapw[1];
    for (int i = tid; i < 2 * P; i += 256) sPos[i] = pos[i];
    const int Mt = p.Nimg * P;                       // rows of this tap (0 for a tap no position reaches)
    const float inv_p = P > 0 ? 1.0f / (float)P : 0.f;
    // Row enumeration (plan flag, words[0] bit 16).  Row-major: chunk c = rows c*64.. of the list (image n, position j) with j fastest:
    // every lane divides its row index and looks its position up (LDS) for every DMA piece — ~300 instructions and four dependent LDS
    // round trips per chunk in front of 16 MFMAs.  IMAGE-BLOCK major (plans with >= 64 images, every dense layer): chunk c = position
    // j = c / cpp of the 64 images nb*64.. (cpp = ceil(Nimg / 64) blocks per position): position and block are wave-uniform, a lane's
    // row is image nb*64 + rt — its address is a scalar base (block, position, tile channel) plus a per-lane constant, so a piece is
    // `global_load_lds_dwordx4 v_off, s[base]` with no vector arithmetic at all; only a position's last block (images past Nimg read
    // the zero page) and tiles that overhang Co / Ci take the path with per-lane selects.
    const bool imgmajor = ((p.plan[0] >> 16) & 1) != 0;
    const int cpp = (p.Nimg + 63) >> 6;
    const int total_chunks = imgmajor ? P * cpp : (Mt + 63) / 64;
    const int c_begin = unit_begin >= 0 ? unit_begin : split_idx * p.chunks_per_split;
    int c_end = unit_begin >= 0 ? unit_begin + unit_count : c_begin + p.chunks_per_split;
    if (c_end > total_chunks) c_end = total_chunks;
    const int KT = c_end > c_begin ? c_end - c_begin : 0;

    // lane -> (row inside a DMA instruction's row group, 16-byte piece); the piece a lane FETCHES is the swizzle image of the piece
    // it writes (the LDS image of a DMA is lane-linear)
    const int lrow = lane / CPR, lpiece = lane % CPR;
    const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(g_wg_zero_page);
    const int wrow0 = __builtin_amdgcn_readfirstlane(wave) * RPI;        // instruction i of wave w covers rows (i*4 + w)*RPI ..
    __syncthreads();

    // image-block mode: per-lane constants of every piece, and the stream position (position sj, image block snb) of the next chunk to stage
    unsigned offY[AR], offX[AR];
    bool chY[AR], chX[AR];
    int rtv[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int rt = i * 4 * RPI + wrow0 + lrow;
        const int f = U == 4 ? (rt & 3) : ((rt >> 1) & 1);
        const int gpiece = (((lpiece >> 2) ^ f) << 2) | (lpiece & 3);
        rtv[i] = rt;
        chY[i] = co0 + gpiece * 8 < p.Co;
        chX[i] = ci0 + gpiece * 8 < p.Ci;
        offY[i] = ((unsigned)rt * (unsigned)p.out_pix * (unsigned)p.out_pitch + (unsigned)(gpiece * 8)) * 2u;
        offX[i] = ((unsigned)rt * (unsigned)p.in_pix * (unsigned)p.in_pitch + (unsigned)(gpiece * 8)) * 2u;
    }
    const bool full_tile = co0 + BC <= p.Co && ci0 + BC <= p.Ci;
    int sj = 0, snb = 0, xp_cur = 0, yp_cur = 0, xp_nxt = 0, yp_nxt = 0;
    if (imgmajor && KT > 0) {
        sj = c_begin / cpp; snb = c_begin - sj * cpp;
        xp_cur = pos[2 * sj]; yp_cur = pos[2 * sj + 1];
        if (sj + 1 < P) { xp_nxt = pos[2 * sj + 2]; yp_nxt = pos[2 * sj + 3]; }
    }
    auto stage_img = [&](int buf) {
        bf16_t* dstY = sStage + buf * S_ELEMS;
        bf16_t* dstX = dstY + T_ELEMS;
        const char* ybase = reinterpret_cast<const char*>(p.dy) + (((long)(snb * 64) * p.out_pix + yp_cur) * p.out_pitch + co0) * 2;
        const char* xbase = reinterpret_cast<const char*>(p.x) + (((long)(snb * 64) * p.in_pix + xp_cur) * p.in_pitch + ci0) * 2;
        const bool tail = snb * 64 + 64 > p.Nimg;
        if (full_tile && !tail) {
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ybase + offY[i]),
                                                 (__attribute__((address_space(3))) void*)(dstY + (i * 4 * RPI + wrow0) * BC), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xbase + offX[i]),
                                                 (__attribute__((address_space(3))) void*)(dstX + (i * 4 * RPI + wrow0) * BC), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const bool ok = snb * 64 + rtv[i] < p.Nimg;
                const void* sy = (ok && chY[i]) ? (const void*)(ybase + offY[i]) : (const void*)zero_src;
                const void* sx = (ok && chX[i]) ? (const void*)(xbase + offX[i]) : (const void*)zero_src;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sy,
                                                 (__attribute__((address_space(3))) void*)(dstY + (i * 4 * RPI + wrow0) * BC), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sx,
                                                 (__attribute__((address_space(3))) void*)(dstX + (i * 4 * RPI + wrow0) * BC), 16, 0, 0);
            }
        }
        // advance the stream: next block of this position, or block 0 of the next position (whose offsets were requested a position ago)
        if (++snb == cpp) {
            snb = 0; ++sj;
            xp_cur = xp_nxt; yp_cur = yp_nxt;
            if (sj + 1 < P) { xp_nxt = pos[2 * sj + 2]; yp_nxt = pos[2 * sj + 3]; }
        }
    };
    auto stage_row = [&](int c, int buf) {
        bf16_t* dstY = sStage + buf * S_ELEMS;
        bf16_t* dstX = dstY + T_ELEMS;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int rt = i * 4 * RPI + wrow0 + lrow;                  // row inside the tile
            const int m = c * 64 + rt;
            const bool ok = m < Mt;
            const int mm = ok ? m : 0;
            int n = (int)((float)mm * inv_p);
            int j = mm - n * P;
            if (j < 0) { n--; j += P; } else if (j >= P) { n++; j -= P; }
            const int xp = sPos[2 * j], yp = sPos[2 * j + 1];
            const int f = U == 4 ? (rt & 3) : ((rt >> 1) & 1);
            const int gpiece = (((lpiece >> 2) ^ f) << 2) | (lpiece & 3);
            const int cy = co0 + gpiece * 8, cx = ci0 + gpiece * 8;
            const bf16_t* sy = (ok && cy < p.Co) ? p.dy + ((long)n * p.out_pix + yp) * p.out_pitch + cy : zero_src;
            const bf16_t* sx = (ok && cx < p.Ci) ? p.x + ((long)n * p.in_pix + xp) * p.in_pitch + cx : zero_src;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sy,
                                             (__attribute__((address_space(3))) void*)(dstY + (i * 4 * RPI + wrow0) * BC), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sx,
                                             (__attribute__((address_space(3))) void*)(dstX + (i * 4 * RPI + wrow0) * BC), 16, 0, 0);
        }
    };
    auto stage = [&](int c, int buf) {           // (chunks are staged in increasing order, one call each: stage_img keeps the position itself)
        if (imgmajor) stage_img(buf); else stage_row(c, buf);
    };

    f32x16 acc[TT][TT], accb[BIAS ? TT : 1];
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        if (BIAS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < TT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    bf16x8 ones;
#pragma unroll
    for (int k = 0; k < 8; ++k) ones[k] = (short)0x3f80;                        // bf16 1.0

#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < KT) stage(c_begin + s, s);
    int buf = 0;
    for (int it = 0; it < KT; ++it) {
        // chunk `it` has landed once at most min(NS-2, chunks left) later chunks' DMAs are still outstanding; the barrier makes
        // every wave's part visible and proves everybody is done reading the buffer the next stage() overwrites
        const int later = KT - 1 - it < NS - 2 ? KT - 1 - it : NS - 2;
        wg_wait_tiles_barrier<LPT, NS - 2>(later);
        if (it + NS - 1 < KT) stage(c_begin + it + NS - 1, buf >= 1 ? buf - 1 : NS - 1);
        const bf16_t* sY = sStage + buf * S_ELEMS;
        const bf16_t* sX = sY + T_ELEMS;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 fa[TT], fb[TT];
#pragma unroll
            for (int i = 0; i < TT; ++i) fa[i] = wg_frag_T<BC>(sY, wco + i * 32, ks * 16, lane);
#pragma unroll
            for (int j = 0; j < TT; ++j) fb[j] = wg_frag_T<BC>(sX, wci + j * 32, ks * 16, lane);
#pragma unroll
            for (int i = 0; i < TT; ++i)
#pragma unroll
                for (int j = 0; j < TT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            if (BIAS) {          // (every wave of a bias block: a lane-dependent condition here costs an AGPR round trip per step)
#pragma unroll
                for (int i = 0; i < TT; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], ones, accb[i], 0, 0, 0);
            }
        }
        buf = buf + 1 == NS ? 0 : buf + 1;
    }
    if (frag_dst != nullptr) {
        f32x4* d4 = reinterpret_cast<f32x4*>(frag_dst);
#pragma unroll
        for (int i = 0; i < TT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    f32x4 v;
                    v[0] = acc[i][j][rq * 4 + 0]; v[1] = acc[i][j][rq * 4 + 1]; v[2] = acc[i][j][rq * 4 + 2]; v[3] = acc[i][j][rq * 4 + 3];
                    d4[(((wave * TT + i) * TT + j) * 4 + rq) * 64 + lane] = v;
                }
        if (BIAS && wci == 0 && (lane & 31) == 0) {
            float* dbd = frag_dst + BC * BC;
#pragma unroll
            for (int i = 0; i < TT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) dbd[wco + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = accb[i][r];
        }
        return;
    }
    // D[row = co][col = ci]: one writer per element — slab `blockIdx.x` (plain store) or, without a split, dW itself
    const bool direct = p.splits <= 1;
    float* dst = direct ? p.dw : p.part + (long)split_idx * p.slab;
#pragma unroll
    for (int j = 0; j < TT; ++j) {
        const int ci = ci0 + wci + j * 32 + (lane & 31);
        if (ci >= p.Ci) continue;
#pragma unroll
        for (int i = 0; i < TT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < p.Co) {
                    float* d = dst + ((long)co * p.wt_taps + tw) * p.Ci + ci;
                    *d = direct ? *d + acc[i][j][r] : acc[i][j][r];
                }
            }
    }
    if (BIAS && wci == 0 && (lane & 31) == 0) {      // every column of accb holds the same sums: lanes 0 and 32 own 16 rows each
        float* dbd = direct ? p.db : p.part + (long)split_idx * p.slab + (long)p.Co * p.wt_taps * p.Ci;
#pragma unroll
        for (int i = 0; i < TT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < p.Co) dbd[co] = direct ? dbd[co] + accb[i][r] : accb[i][r];
            }
    }
}

template <int BC, int NS>
__global__ __launch_bounds__(256) void k_igemm_wgrad(const WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // the workgroups of ci tile 0 / tap 0 also produce the bias gradient: a scalar (blockIdx) branch between two instantiations
    const int co_tiles = (p.Co + BC - 1) / BC, ci_tiles = (p.Ci + BC - 1) / BC;
    const int rest = (int)blockIdx.y / co_tiles;
    if (p.db != nullptr && rest % ci_tiles == 0 && rest / ci_tiles == 0) wg_body<BC, NS, true>(p, smem_raw, blockIdx.x, blockIdx.y);
    else wg_body<BC, NS, false>(p, smem_raw, blockIdx.x, blockIdx.y);
}

// GROUPED launch: one grid over the tiles of several independent weight-gradient problems (the encoder's and the heads' linear layers:
// 26 contractions of 15 K chunks each, 7-17 us apiece as launches of their own — latency, not work).  table[i] = the problem's arguments
// and the first tile it owns; every workgroup finds its problem with a scan over the (wave-uniform) table and runs the ordinary body.
// No K split inside a group (splits = 1: each tile adds its result to dW directly, one writer per element), so nothing needs a slab.
struct WgradGroupEntry { WgradArgs a; int tile_begin; int pad_; };

// The table reaches the device as kernel arguments of a writer kernel (16 entries a launch), not as a host-to-device copy: a launch is
// captured into a HIP graph (and re-issued by the native step list) with its arguments by value, a copy node would keep a pointer
// into host memory that is gone by the time the graph replays.
constexpr int WG_TABLE_CHUNK = 16;
struct WgradGroupChunk { WgradGroupEntry e[WG_TABLE_CHUNK]; };
static_assert(sizeof(WgradGroupEntry) % 8 == 0 && sizeof(WgradGroupChunk) <= 3584, "a chunk travels in the kernel-argument segment");

__global__ __launch_bounds__(256) void k_wgrad_group_table(const WgradGroupChunk c, int words, long long* __restrict__ dst) {
    const long long* src = reinterpret_cast<const long long*>(&c);
    for (int i = threadIdx.x; i < words; i += 256) dst[i] = src[i];
}

template <int BC, int NS>
__global__ __launch_bounds__(256) void k_igemm_wgrad_group(const WgradGroupEntry* __restrict__ table, int n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int idx = 0;
    for (int i = 1; i < n; ++i)
        if ((int)blockIdx.x >= table[i].tile_begin) idx = i;
    const WgradArgs p = table[idx].a;
    const int task = (int)blockIdx.x - table[idx].tile_begin;
    const int co_tiles = (p.Co + BC - 1) / BC, ci_tiles = (p.Ci + BC - 1) / BC;
    const int rest = task / co_tiles;
    if (p.db != nullptr && rest % ci_tiles == 0 && rest / ci_tiles == 0) wg_body<BC, NS, true>(p, smem_raw, 0, task);
    else wg_body<BC, NS, false>(p, smem_raw, 0, task);
}

// UNIT-LIST launch (plan format 2): the grid is a host-built list of work units {task, first chunk, chunks, slab slot} of nearly equal
// length.  The (split, task) grid above gives every tap the split count of the LONGEST tap: on a 3 x 3 map (layer4) the taps have 9 / 6 / 4
// positions, so a third of the workgroups ran 45 chunks while the rest ran 15 or none — the launch lasted twice its mean.  Here a task of L
// chunks is cut into round(L / target) units, tiles leave in fragment layout (above) and one reduce launch sums each task's slots.
// Unit words: {task, chunk_begin, chunk_count, slot (-1: the task's only unit adds to dW directly)}.
#define WUNIT_WORDS 4
struct WgradUnitArgs { WgradArgs a; const int* units; long tile_stride; };

template <int BC, int NS>
__device__ __forceinline__ void wg_unit(const WgradArgs& p, const int* __restrict__ u, long tile_stride, unsigned char* smem_raw) {
    const int task = u[0], cb = u[1], cn = u[2], slot = u[3];
    const int co_tiles = (p.Co + BC - 1) / BC, ci_tiles = (p.Ci + BC - 1) / BC;
    const int rest = task / co_tiles;
    float* dst = slot >= 0 ? p.part + (long)slot * tile_stride : nullptr;
    if (p.db != nullptr && rest % ci_tiles == 0 && rest / ci_tiles == 0) wg_body<BC, NS, true>(p, smem_raw, 0, task, dst, cb, cn);
    else wg_body<BC, NS, false>(p, smem_raw, 0, task, dst, cb, cn);
}

template <int BC, int NS>
__global__ __launch_bounds__(256) void k_igemm_wgrad_units(const WgradUnitArgs q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    wg_unit<BC, NS>(q.a, q.units + (long)blockIdx.x * WUNIT_WORDS, q.tile_stride, smem_raw);
}

// dW[co][tw][ci] += sum over a task's slots (in slot order) of its partial tiles; db likewise.  Task words: {slot_begin, slots}.
// grid (BC*BC/4/64 (+1 for the bias row), tasks); a thread owns one float4 group of the fragment layout and one of four slot lanes.
struct WgradReduceArgs {
    const float* part; float* dw; float* db; const int* plan; const int* tasktab;
    long tile_stride; int Co, Ci, wt_taps, co_tiles, ci_tiles;
};

// Block = 64 float4 groups x 4 slot lanes: lane sl adds slots sl, sl + 4, ... in increasing order (two loads in flight), the four lane
// sums are added in lane order — a fixed association, and a task with a hundred slots (the 1 x 1 convolutions: two tasks, K split 200
// ways) is not one thread's chain of a hundred dependent round trips.
template <int BC>
__global__ __launch_bounds__(256) void k_wgrad_unit_reduce(const WgradReduceArgs q) {
    constexpr int TT = BC / 64, WT = BC / 2, GB = BC * BC / 4 / 64;       // GB: blocks of 64 groups per tile
    __shared__ f32x4 sred[4][64];
    const int task = blockIdx.y, blk = blockIdx.x;
    const int s0 = q.tasktab[2 * task], ns = q.tasktab[2 * task + 1];
    if (ns <= 0) return;                                // (a task whose single unit wrote dW itself)
    int rest = task;
    const int cot = rest % q.co_tiles; rest /= q.co_tiles;
    const int cit = rest % q.ci_tiles; rest /= q.ci_tiles;
    const int tw = q.plan[WPLAN_HDR + rest * WPLAN_TAP_WORDS + 2];
    const float* src = q.part + (long)s0 * q.tile_stride;
    if (blk == GB) {                                    // the bias sums of this task's column tile (tasks of ci tile 0 / tap 0 only)
        if (q.db == nullptr || cit != 0 || rest != 0) return;
        for (int c = threadIdx.x; c < BC; c += 256) {
            float a = 0.f;
            for (int s = 0; s < ns; ++s) a += src[(long)s * q.tile_stride + BC * BC + c];
            if (cot * BC + c < q.Co) q.db[cot * BC + c] += a;
        }
        return;
    }
    const int gl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int g = blk * 64 + gl;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    int s = sl;
    for (; s + 28 < ns; s += 32) {          // eight loads in flight, added in the lane's slot order
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = reinterpret_cast<const f32x4*>(src + (long)(s + 4 * k) * q.tile_stride)[g];
#pragma unroll
        for (int k = 0; k < 8; ++k) a = a + v[k];
    }
    for (; s + 4 < ns; s += 8) {
        const f32x4 v0 = reinterpret_cast<const f32x4*>(src + (long)s * q.tile_stride)[g];
        const f32x4 v1 = reinterpret_cast<const f32x4*>(src + (long)(s + 4) * q.tile_stride)[g];
        a = (a + v0) + v1;
    }
    for (; s < ns; s += 4) a = a + reinterpret_cast<const f32x4*>(src + (long)s * q.tile_stride)[g];
    sred[sl][gl] = a;
    __syncthreads();
    if (sl != 0) return;
    a = ((sred[0][gl] + sred[1][gl]) + sred[2][gl]) + sred[3][gl];
    const int lane = g & 63, rq = (g >> 6) & 3, fj = (g >> 8) % TT, fi = (g >> 8) / TT % TT, wave = (g >> 8) / (TT * TT);
    const int ci = cit * BC + (wave & 1) * WT + fj * 32 + (lane & 31);
    const int co = cot * BC + (wave >> 1) * WT + fi * 32 + 8 * rq + 4 * (lane >> 5);
    if (ci >= q.Ci) return;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (co + k < q.Co) {
            float* d = q.dw + ((long)(co + k) * q.wt_taps + tw) * q.Ci + ci;
            *d += a[k];
        }
}

// The same for plans whose tasks have FEW slots (the sentence-level model's dense layers: 3-4 units per 128 x 128 tile, 144 tiles): a thread
// owns one float4 group and adds the slots in order with all loads in flight; 256 groups per block.  (The four-slot-lane form above spends
// a 256-thread block on 64 groups and one load per thread: 9,360 blocks of ~4 KB for a 3,072 x 768 gradient, 17 us for 33 MB.)
template <int BC>
__global__ __launch_bounds__(256) void k_wgrad_unit_reduce_flat(const WgradReduceArgs q) {
    constexpr int TT = BC / 64, WT = BC / 2, GB = BC * BC / 4 / 256;      // GB: blocks of 256 groups per tile
    const int task = blockIdx.y, blk = blockIdx.x;
    const int s0 = q.tasktab[2 * task], ns = q.tasktab[2 * task + 1];
    if (ns <= 0) return;
    int rest = task;
    const int cot = rest % q.co_tiles; rest /= q.co_tiles;
    const int cit = rest % q.ci_tiles; rest /= q.ci_tiles;
    const int tw = q.plan[WPLAN_HDR + rest * WPLAN_TAP_WORDS + 2];
    const float* src = q.part + (long)s0 * q.tile_stride;
    if (blk == GB) {
        if (q.db == nullptr || cit != 0 || rest != 0) return;
        for (int c = threadIdx.x; c < BC; c += 256) {
            float a = 0.f;
            for (int s = 0; s < ns; ++s) a += src[(long)s * q.tile_stride + BC * BC + c];
            if (cot * BC + c < q.Co) q.db[cot * BC + c] += a;
        }
        return;
    }
    const int g = blk * 256 + threadIdx.x;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 3 < ns; s += 4) {
        const f32x4 v0 = reinterpret_cast<const f32x4*>(src + (long)s * q.tile_stride)[g];
        const f32x4 v1 = reinterpret_cast<const f32x4*>(src + (long)(s + 1) * q.tile_stride)[g];
        const f32x4 v2 = reinterpret_cast<const f32x4*>(src + (long)(s + 2) * q.tile_stride)[g];
        const f32x4 v3 = reinterpret_cast<const f32x4*>(src + (long)(s + 3) * q.tile_stride)[g];
        a = (((a + v0) + v1) + v2) + v3;
    }
    for (; s < ns; ++s) a = a + reinterpret_cast<const f32x4*>(src + (long)s * q.tile_stride)[g];
    const int lane = g & 63, rq = (g >> 6) & 3, fj = (g >> 8) % TT, fi = (g >> 8) / TT % TT, wave = (g >> 8) / (TT * TT);
    const int ci = cit * BC + (wave & 1) * WT + fj * 32 + (lane & 31);
    const int co = cot * BC + (wave >> 1) * WT + fi * 32 + 8 * rq + 4 * (lane >> 5);
    if (ci >= q.Ci) return;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (co + k < q.Co) {
            float* d = q.dw + ((long)(co + k) * q.wt_taps + tw) * q.Ci + ci;
            *d += a[k];
        }
}

// ---- host side -------------------------------------------------------------------------------------------------------
struct WgradLaunch { int bc, ns, splits, chunks_per_split, tasks; long slab; };

// image-block row enumeration (see wg_body): from 64 images on (a partial block of fewer would be mostly padding)
static bool wgrad_imgmajor(int Nimg) { return Nimg >= 64 && svsr_tune_get(SVSR_TUNE_WG_IMGMAJOR) != 0; }

static WgradLaunch wgrad_launch(long total_chunks_l, int Co, int Ci, int ntaps, int wt_taps, bool bias) {
    WgradLaunch pl;
    const int tasks128 = ((Co + 127) / 128) * ((Ci + 127) / 128) * ntaps;
    const int total_chunks = (int)total_chunks_l;
    // short contractions (LRW encoder linears: 960 rows = 15 chunks): a workgroup's whole K loop is a few microseconds, so the cost
    // is the fp32 slab written per split and the reduction launch behind it.  64-wide tiles give enough tasks to fill the chip
    // WITHOUT splitting K: no slabs, no second launch (qkv 20.6 -> 16.1 us, ffn1 23.1 -> 16.3, audio head 25.7 -> 17.3).
    const int short_k = svsr_tune_get(SVSR_TUNE_WG_SHORT_K);
    const bool is_short = total_chunks <= short_k;
    pl.bc = (Co >= 128 && Ci >= 128 && tasks128 >= 36 && !is_short) ? 128 : 64;
    pl.ns = pl.bc == 128 ? 2 : 3;                      // (a 6-deep ring for the short contractions measured no gain: 16.6 vs 16.1 us)
    const int BC = pl.bc;
    pl.tasks = ((Co + BC - 1) / BC) * ((Ci + BC - 1) / BC) * ntaps;
    // one round of resident workgroups: 2 per CU for the 128-wide tile (64 KiB of LDS ring, ~200 VGPRs), 3 per CU for the 64-wide
    // one — a grid a little above that runs a second, almost empty round (layer4: 576 workgroups on 512 slots took 1.4x longer)
    const int cus = svsr_reduction_cus();      // (a fixed number, not the device's: the split decides the order of the additions)
    const int target_env = svsr_tune_get(SVSR_TUNE_WG_BLOCKS);
    const int target_blocks = target_env > 0 ? target_env : cus * (BC == 128 ? 2 : 3);
    int splits = target_blocks / pl.tasks;                    // every split costs a slab of Co*taps*Ci floats written and re-read
    if (is_short && pl.tasks >= cus / 2) splits = 1;
    if (splits > total_chunks) splits = total_chunks;
    if (splits < 1) splits = 1;
    pl.chunks_per_split = (total_chunks + splits - 1) / splits;
    // keep the epilogue amortised: >= 12 K-chunks per block when the problem has them (LRS linears: 2,400 rows =
    // 38 chunks -> 3 splits measured best), >= 4 for the short LRW sequences
    const int min_chunks = total_chunks >= 36 ? 12 : 4;
    if (pl.chunks_per_split < min_chunks && total_chunks >= min_chunks) pl.chunks_per_split = min_chunks;
    pl.splits = (total_chunks + pl.chunks_per_split - 1) / pl.chunks_per_split;
    if (pl.splits < 1) pl.splits = 1;
    pl.slab = (long)Co * wt_taps * Ci + (bias ? Co : 0);
    return pl;
}

// Unit list of a plan (format 2): tasks in index order (tap, ci tile, co tile), a task of L chunks cut into n = ceil(L / U) units of
// floor / ceil(L / n) chunks, U the smallest unit length that keeps the number of units within `rounds` full rounds of resident
// workgroups (a grid a little above a round runs an almost empty extra round: layer4's 576 workgroups on 512 slots took 1.4x longer).
struct WgradUnits { std::vector<int> units, tasktab; int slots = 0; };

static WgradUnits wgrad_units(const std::vector<long>& tap_chunks, int co_tiles, int ci_tiles, int bc) {
    const int cus = svsr_reduction_cus();      // (a fixed number, not the device's: the split decides the order of the additions)
    const int target_env = svsr_tune_get(SVSR_TUNE_WG_BLOCKS);
    const long resident = target_env > 0 ? target_env : (long)cus * (bc == 128 ? 2 : 3);
    const int umax = svsr_tune_get(SVSR_TUNE_WG_UNIT_MAX) > 0 ? svsr_tune_get(SVSR_TUNE_WG_UNIT_MAX) : 48;
    const int umin = svsr_tune_get(SVSR_TUNE_WG_UNIT_MIN) > 0 ? svsr_tune_get(SVSR_TUNE_WG_UNIT_MIN) : 8;
    const int tiles = co_tiles * ci_tiles;
    long total = 0, longest = 0;
    for (long L : tap_chunks) { total += L * tiles; if (L > longest) longest = L; }
    long rounds = (total + resident * umax - 1) / (resident * umax);
    if (rounds < 1) rounds = 1;
    long U = (total + resident * rounds - 1) / (resident * rounds);
    if (U < umin) U = umin;
    for (;; ++U) {
        long n = 0;
        for (long L : tap_chunks) n += ((L + U - 1) / U) * tiles;
        if (n <= resident * rounds || U >= longest) break;
    }
    WgradUnits w;
    const int ntaps = (int)tap_chunks.size();
    w.tasktab.assign((size_t)2 * ntaps * tiles, 0);
    for (int t = 0; t < ntaps; ++t) {
        const long L = tap_chunks[t];
        const int n = (int)((L + U - 1) / U);
        for (int tile = 0; tile < tiles; ++tile) {
            const int task = t * tiles + tile;         // == (t * ci_tiles + cit) * co_tiles + cot
            if (n > 1) { w.tasktab[2 * task] = w.slots; w.tasktab[2 * task + 1] = n; }
            long c = 0;
            for (int k = 0; k < n; ++k) {
                const long len = L / n + (k < L % n ? 1 : 0);
                w.units.push_back(task); w.units.push_back((int)c); w.units.push_back((int)len); w.units.push_back(n > 1 ? w.slots++ : -1);
                c += len;
            }
        }
    }
    // long units first: the dispatcher hands workgroups out in index order, the short ones fill the tail of the round
    const int nu = (int)(w.units.size() / WUNIT_WORDS);
    std::vector<int> order(nu);
    for (int i = 0; i < nu; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w.units[a * WUNIT_WORDS + 2] > w.units[b * WUNIT_WORDS + 2]; });
    if (svsr_tune_get(SVSR_TUNE_WG_XCD) && ntaps > 1) {
        // Multi-tap plans (the 3 x 3 / stride-2 convolutions): the units of all taps and channel tiles that cover the same stretch of the
        // contraction read the same rows of x and dy.  Workgroup b runs on XCD b % 8 (observed, never relied on for correctness), each XCD has
        // its own L2, and in plain index order every XCD ends up streaming ALL rows: 289 MB of fabric traffic per layer3 launch for <= 58 MB of
        // operands, next to which the main stream's HBM-bound BatchNorm passes ran three times slower.  So the units are dealt to eight queues
        // by the eighth of their task's contraction they sit in, long units first inside a queue, and the queues are interleaved: XCD x then
        // touches about one eighth of the rows.  The order of a task's slots — the order its partial tiles are added in — is untouched.
        // Measured (round 5): FETCH_SIZE per launch 261 -> 155 MB (layer3 / layer4 3 x 3), 160 -> 97 MB (stride-2 / 1 x 1 plans); standalone weight
        // gradient + reduce of layer3's 3 x 3 69.7 -> 48.4 us (565 -> 814 TFLOP/s), layer3.0.conv1 53.4 -> 39.0, layer2.0.conv1 70.3 -> 44.7, layer4
        // unchanged; LRW step 5.00-5.02 -> 4.92-4.93 ms same box, LRS -0.1 ms.  Walking a queue's stretch in order instead of long-first, and
        // dealing single-tap plans (1 x 1 convolutions, dense layers) the same way, measured no further change.
        std::vector<std::vector<int>> q(8);
        for (int i = 0; i < nu; ++i) {
            const int u = order[i];
            const int task = w.units[u * WUNIT_WORDS], cb = w.units[u * WUNIT_WORDS + 1], cn = w.units[u * WUNIT_WORDS + 2];
            const long L = tap_chunks[task / tiles];
            int x = (int)((8 * (2L * cb + cn)) / (2 * (L > 0 ? L : 1)));
            q[x < 0 ? 0 : (x > 7 ? 7 : x)].push_back(u);
        }
        std::vector<int> dealt;
        std::vector<size_t> at(8, 0);
        for (bool any = true; any;) {
            any = false;
            for (int x = 0; x < 8; ++x)
                if (at[x] < q[x].size()) { dealt.push_back(q[x][at[x]++]); any = true; }
        }
        order.swap(dealt);
    }
    std::vector<int> sorted((size_t)nu * WUNIT_WORDS);
    for (int i = 0; i < nu; ++i) std::copy(w.units.begin() + order[i] * WUNIT_WORDS, w.units.begin() + (order[i] + 1) * WUNIT_WORDS, sorted.begin() + i * WUNIT_WORDS);
    w.units.swap(sorted);
    return w;
}

// meta: {tile edge, ring depth, K splits, chunks per split, tasks, taps, max positions of a tap, 0}
// format 2 (unit list; meta[7] > 0): {tile edge, ring depth, slab slots, units, tasks, taps, max positions, word offset of the unit table};
// the task table {first slot, slots} x tasks follows the units.
static int wplan_emit(const std::vector<std::vector<int>>& taps_pos, const std::vector<int>& tws, int Nimg, int Co, int Ci, int wt_taps,
                      int has_bias, int* words, int cap_words, int* meta, int64_t* part_floats) {
    const int ntaps = (int)taps_pos.size();
    int nwords = WPLAN_HDR + ntaps * WPLAN_TAP_WORDS;
    const int pos_word0 = nwords;
    long maxP = 0;
    for (const auto& v : taps_pos) { nwords += (int)v.size(); if ((long)v.size() / 2 > maxP) maxP = (long)v.size() / 2; }
    if (maxP < 1 || maxP > WG_MAXP || (long)Nimg * maxP >= (1L << 24)) return -SVSR_ERR_ARG;
    const bool im = wgrad_imgmajor(Nimg);
    const long chunks = im ? maxP * (long)((Nimg + 63) / 64) : ((long)Nimg * maxP + 63) / 64;
    const WgradLaunch pl = wgrad_launch(chunks, Co, Ci, ntaps, wt_taps, has_bias != 0);
    // unit lists for the long contractions (the short ones — LRW's 960-row linears — have no K split to balance and ride the grouped launch)
    const bool use_units = svsr_tune_get(SVSR_TUNE_WG_UNITS) != 0 && chunks > svsr_tune_get(SVSR_TUNE_WG_SHORT_K);
    WgradUnits wu;
    const int units_word0 = nwords;
    if (use_units) {
        std::vector<long> tap_chunks;
        for (const auto& v : taps_pos) { const long P = (long)v.size() / 2; tap_chunks.push_back(im ? P * (long)((Nimg + 63) / 64) : ((long)Nimg * P + 63) / 64); }
        wu = wgrad_units(tap_chunks, (Co + pl.bc - 1) / pl.bc, (Ci + pl.bc - 1) / pl.bc, pl.bc);
        nwords += (int)wu.units.size() + (int)wu.tasktab.size();
    }
    if (words != nullptr) {
        if (cap_words < nwords) return -SVSR_ERR_ARG;
        if (use_units) {
            std::copy(wu.units.begin(), wu.units.end(), words + units_word0);
            std::copy(wu.tasktab.begin(), wu.tasktab.end(), words + units_word0 + wu.units.size());
        }
        words[0] = ntaps | ((im ? 1 : 0) << 16); words[1] = pos_word0;
        int pos_off = 0;
        for (int t = 0; t < ntaps; ++t) {
            int* w = words + WPLAN_HDR + t * WPLAN_TAP_WORDS;
            const int P = (int)(taps_pos[t].size() / 2);
            w[0] = P; w[1] = pos_off; w[2] = tws[t]; w[3] = 0;
            std::copy(taps_pos[t].begin(), taps_pos[t].end(), words + pos_word0 + 2 * pos_off);
            pos_off += P;
        }
    }
    if (meta != nullptr) {
        meta[0] = pl.bc; meta[1] = pl.ns; meta[2] = pl.splits; meta[3] = pl.chunks_per_split; meta[4] = pl.tasks; meta[5] = ntaps;
        meta[6] = (int)maxP; meta[7] = 0;
        if (use_units) { meta[2] = wu.slots; meta[3] = (int)(wu.units.size() / WUNIT_WORDS); meta[7] = units_word0; }
    }
    if (part_floats != nullptr) {
        *part_floats = pl.splits > 1 ? (int64_t)pl.splits * pl.slab : 0;
        if (use_units) *part_floats = (int64_t)wu.slots * ((int64_t)pl.bc * pl.bc + (has_bias ? pl.bc : 0));
    }
    return nwords;
}

template <int BC, int NS>
static int launch_wgrad(const WgradArgs& a, int tasks, int maxP, hipStream_t stream) {
    const size_t lds_max = (size_t)NS * 2 * 64 * BC * sizeof(bf16_t) + (size_t)WG_MAXP * 2 * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_igemm_wgrad<BC, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        attr_set = true;
    }
    const size_t lds = (size_t)NS * 2 * 64 * BC * sizeof(bf16_t) + (((size_t)maxP * 2 * sizeof(int) + 127) & ~(size_t)127);
    hipLaunchKernelGGL((k_igemm_wgrad<BC, NS>), dim3(a.splits, tasks), dim3(256), lds, stream, a);
    return svsr_check_launch();
}

template <int BC, int NS>
static int launch_wgrad_units(const WgradArgs& a, const int* plan_dev, const int* meta, hipStream_t stream) {
    const int slots = meta[2], n_units = meta[3], tasks = meta[4], maxP = meta[6];
    WgradUnitArgs q;
    q.a = a; q.a.splits = 1;
    q.units = plan_dev + meta[7];
    q.tile_stride = (long)BC * BC + (a.db != nullptr ? BC : 0);
    if (n_units > 0) {
        const size_t lds_max = (size_t)NS * 2 * 64 * BC * sizeof(bf16_t) + (size_t)WG_MAXP * 2 * sizeof(int);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_igemm_wgrad_units<BC, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
            attr_set = true;
        }
        const size_t lds = (size_t)NS * 2 * 64 * BC * sizeof(bf16_t) + (((size_t)maxP * 2 * sizeof(int) + 127) & ~(size_t)127);
        hipLaunchKernelGGL((k_igemm_wgrad_units<BC, NS>), dim3(n_units), dim3(256), lds, stream, q);
        const int rc = svsr_check_launch();
        if (rc != SVSR_OK) return rc;
    }
    if (slots <= 0) return SVSR_OK;
    WgradReduceArgs r;
    r.part = a.part; r.dw = a.dw; r.db = a.db; r.plan = plan_dev; r.tasktab = plan_dev + meta[7] + n_units * WUNIT_WORDS;
    r.tile_stride = q.tile_stride; r.Co = a.Co; r.Ci = a.Ci; r.wt_taps = a.wt_taps;
    r.co_tiles = (a.Co + BC - 1) / BC; r.ci_tiles = (a.Ci + BC - 1) / BC;
    // (slots / tasks: the mean number of partial tiles per task; with few of them a thread adds them all, see k_wgrad_unit_reduce_flat)
    if (slots <= 8 * tasks) hipLaunchKernelGGL((k_wgrad_unit_reduce_flat<BC>), dim3(BC * BC / 1024 + (a.db != nullptr ? 1 : 0), tasks), dim3(256), 0, stream, r);
    else hipLaunchKernelGGL((k_wgrad_unit_reduce<BC>), dim3(BC * BC / 256 + (a.db != nullptr ? 1 : 0), tasks), dim3(256), 0, stream, r);
    return svsr_check_launch();
}

extern "C" {

/* svsr_wgrad_plan (host): plan of the weight gradient of a k x k / stride / pad convolution over Nimg images [H][W] -> [Ho][Wo]
 * (x = forward input pixels, dy = output-gradient pixels; dw [Co][k*k][Ci]).  words == null: only counts.  Returns the number
 * of int32 words or a negative error; meta[8] = {tile edge, ring depth, K splits, chunks per split, tasks, taps, max positions};
 * *part_floats = workspace floats svsr_igemm_wgrad needs (0 when a single split writes dW directly). */
int svsr_wgrad_plan(int Nimg, int H, int W, int Ci, int Co, int k, int stride, int pad, int* words, int cap_words, int* meta,
                    int64_t* part_floats) {
    if (Nimg < 1 || H < 1 || W < 1 || k < 1 || k > 3 || stride < 1 || pad < 0 || Ci < 1 || Co < 1) return -SVSR_ERR_ARG;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (Ho < 1 || Wo < 1) return -SVSR_ERR_ARG;
    std::vector<std::vector<int>> tp;
    std::vector<int> tws;
    for (int kh = 0; kh < k; ++kh)
        for (int kw = 0; kw < k; ++kw) {
            std::vector<int> v;      // stays empty for a tap no position reaches (its slab tiles are written as zeros)
            for (int a = 0; a < Ho; ++a)
                for (int b = 0; b < Wo; ++b) {
                    const int iy = a * stride + kh - pad, ix = b * stride + kw - pad;
                    if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                    v.push_back(iy * W + ix);
                    v.push_back(a * Wo + b);
                }
            tp.push_back(v);
            tws.push_back(kh * k + kw);
        }
    return wplan_emit(tp, tws, Nimg, Co, Ci, k * k, 0, words, cap_words, meta, part_floats);
}

/* svsr_wgrad_rows_plan (host): dense layer over Nimg sequences: row (n, j < P) pairs source row n*in_pix + src0 + j with
 * target row n*out_pix + dst0 + j (plain linear: P = 1, in_pix = out_pix = 1). */
int svsr_wgrad_rows_plan(int Nimg, int P, int src0, int dst0, int Ci, int Co, int has_bias, int* words, int cap_words, int* meta,
                         int64_t* part_floats) {
    if (Nimg < 1 || P < 1 || Ci < 1 || Co < 1) return -SVSR_ERR_ARG;
    std::vector<std::vector<int>> tp(1);
    for (int j = 0; j < P; ++j) { tp[0].push_back(src0 + j); tp[0].push_back(dst0 + j); }
    return wplan_emit(tp, std::vector<int>{0}, Nimg, Co, Ci, 1, has_bias, words, cap_words, meta, part_floats);
}

int svsr_colsum_rows(const float* ws, int nrows, int64_t ld, float* out0, int64_t n0, float* out1, int64_t n1, int accumulate, float scale,
                     hipStream_t stream);

/* svsr_igemm_wgrad: runs a plan (plan_dev = device copy of the words, meta = the host meta of svsr_wgrad_*plan).
 * dw fp32 [Co][wt_taps][Ci] += ...; dbias (optional, fp32 [Co]) += column sums of dy.  part: workspace of *part_floats floats. */
int svsr_igemm_wgrad(const void* x, const void* dyp, float* dw, float* dbias, const int* plan_dev, const int* meta, int Nimg, int in_pix,
                     int Ci, int in_pitch, int Co, int out_pix, int out_pitch, int wt_taps, float* part, int64_t part_floats,
                     hipStream_t stream) {
    if (plan_dev == nullptr || meta == nullptr || Ci < 1 || Co < 1 || in_pitch % 8 != 0 || out_pitch % 8 != 0 || Ci % 8 != 0 || Nimg < 1)
        return SVSR_ERR_ARG;
    WgradArgs a;
    a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dyp; a.dw = dw; a.db = dbias; a.part = part; a.plan = plan_dev;
    a.splits = meta[2]; a.chunks_per_split = meta[3];
    a.slab = (long)Co * wt_taps * Ci + (dbias != nullptr ? Co : 0);
    a.Nimg = Nimg; a.in_pix = in_pix; a.Ci = Ci; a.in_pitch = in_pitch; a.Co = Co; a.out_pix = out_pix; a.out_pitch = out_pitch; a.wt_taps = wt_taps;
    const int bc = meta[0], ns = meta[1], tasks = meta[4], maxP = meta[6];
    if (meta[7] > 0) {          // unit-list plan
        const int64_t need = (int64_t)meta[2] * ((int64_t)bc * bc + (dbias != nullptr ? bc : 0));
        if (meta[2] > 0 && (part == nullptr || part_floats < need)) return SVSR_ERR_ARG;
        if (bc == 128 && ns == 2) return launch_wgrad_units<128, 2>(a, plan_dev, meta, stream);
        if (bc == 64 && ns == 3) return launch_wgrad_units<64, 3>(a, plan_dev, meta, stream);
        return SVSR_ERR_ARG;
    }
    if (a.splits > 1 && (part == nullptr || part_floats < (int64_t)a.splits * a.slab)) return SVSR_ERR_ARG;
    int rc;
    if (bc == 128 && ns == 2) rc = launch_wgrad<128, 2>(a, tasks, maxP, stream);
    else if (bc == 64 && ns == 3) rc = launch_wgrad<64, 3>(a, tasks, maxP, stream);
    else return SVSR_ERR_ARG;
    if (rc != SVSR_OK || a.splits <= 1) return rc;
    const int64_t n = (int64_t)Co * wt_taps * Ci;
    return svsr_colsum_rows(part, a.splits, a.slab, dw, n, dbias, dbias != nullptr ? Co : 0, 1, 1.0f, stream);
}

/* svsr_igemm_wgrad_group: n independent svsr_igemm_wgrad problems (each described like a call of its own: plan_dev + meta of
 * svsr_wgrad_*plan) in ONE launch.  Every problem must have a plan without K split on 64-wide tiles (meta = {64, 3, 1, ...}: the short
 * contractions this exists for) — anything else returns SVSR_ERR_ARG and the caller launches it alone.  table_dev: caller-owned device
 * buffer of at least svsr_igemm_wgrad_group_bytes(n) bytes (the problem table is written there on `stream`, by a kernel that carries it as arguments, ahead of the contraction: graph- and replay-safe). */
int64_t svsr_igemm_wgrad_group_bytes(int n) { return n < 1 ? 0 : (int64_t)n * (int64_t)sizeof(WgradGroupEntry); }

int svsr_igemm_wgrad_group(const svsr_wgrad_problem* problems, int n, void* table_dev, int64_t table_bytes, hipStream_t stream) {
    if (problems == nullptr || n < 1 || n > 256 || table_dev == nullptr || table_bytes < svsr_igemm_wgrad_group_bytes(n)) return SVSR_ERR_ARG;
    std::vector<WgradGroupEntry> tab((size_t)((n + WG_TABLE_CHUNK - 1) / WG_TABLE_CHUNK) * WG_TABLE_CHUNK);
    int tiles = 0, maxP = 1;
    for (int i = 0; i < n; ++i) {
        const svsr_wgrad_problem& q = problems[i];
        if (q.plan_dev == nullptr || q.meta == nullptr || q.Ci < 1 || q.Co < 1 || q.in_pitch % 8 != 0 || q.out_pitch % 8 != 0 || q.Ci % 8 != 0 || q.Nimg < 1 ||
            q.x == nullptr || q.dy == nullptr || q.dw == nullptr)
            return SVSR_ERR_ARG;
        if (q.meta[0] != 64 || q.meta[1] != 3 || q.meta[2] != 1) return SVSR_ERR_ARG;
        WgradArgs& a = tab[i].a;
        a.x = (const bf16_t*)q.x; a.dy = (const bf16_t*)q.dy; a.dw = q.dw; a.db = q.dbias; a.part = nullptr; a.plan = q.plan_dev;
        a.splits = 1; a.chunks_per_split = q.meta[3];
        a.slab = (long)q.Co * q.wt_taps * q.Ci + (q.dbias != nullptr ? q.Co : 0);
        a.Nimg = q.Nimg; a.in_pix = q.in_pix; a.Ci = q.Ci; a.in_pitch = q.in_pitch; a.Co = q.Co; a.out_pix = q.out_pix; a.out_pitch = q.out_pitch;
        a.wt_taps = q.wt_taps;
        tab[i].tile_begin = tiles; tab[i].pad_ = 0;
        tiles += q.meta[4];
        if (q.meta[6] > maxP) maxP = q.meta[6];
    }
    for (int i0 = 0; i0 < n; i0 += WG_TABLE_CHUNK) {
        const int cnt = n - i0 < WG_TABLE_CHUNK ? n - i0 : WG_TABLE_CHUNK;
        WgradGroupChunk chunk;
        std::memcpy(&chunk, &tab[(size_t)i0], sizeof(chunk));
        hipLaunchKernelGGL(k_wgrad_group_table, dim3(1), dim3(256), 0, stream, chunk, (int)(cnt * sizeof(WgradGroupEntry) / 8),
                           reinterpret_cast<long long*>(static_cast<WgradGroupEntry*>(table_dev) + i0));
    }
    constexpr int BC = 64, NS = 3;
    const size_t lds_max = (size_t)NS * 2 * 64 * BC * sizeof(bf16_t) + (size_t)WG_MAXP * 2 * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_igemm_wgrad_group<BC, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        attr_set = true;
    }
    const size_t lds = (size_t)NS * 2 * 64 * BC * sizeof(bf16_t) + (((size_t)maxP * 2 * sizeof(int) + 127) & ~(size_t)127);
    hipLaunchKernelGGL((k_igemm_wgrad_group<BC, NS>), dim3(tiles), dim3(256), lds, stream, (const WgradGroupEntry*)table_dev, n);
    return svsr_check_launch();
}

}  // extern "C"
