// Weight-gradient contraction on MFMA (gfx950):
//     dW[co][tw[t]][ci] += sum_m dY[target(m)][co] * X[source(m, t)][ci]            (fp32)
// Split-K without atomics: the workgroups of split s store their tiles into slab s of a caller-owned workspace with plain
// stores and svsr_colsum_rows (runtime.hip) adds the slabs into dW in a fixed order, so the gradient is reproducible; with a
// single split the tile is added to dW directly (one writer per element).
// for every nn.Conv2d / nn.Linear of the path (autograd of reference LRW/video/src/tcn/models/resnet.py:8-16,59-72 and
// lightning.py:82,92,107).  The reduction index m (positions) is the slow index of both operands, so
// the MFMA fragments (8 consecutive positions for one channel per lane) are produced from position-major LDS tiles by
// the gfx950 transpose read ds_read_b64_tr_b16.  A 32-lane group of that read touches 4 consecutive rows x 64 bytes, so the
// row pitch must be 16 * odd banks (mod 64) for the four 16-bank windows to be disjoint: rows are padded by 32 elements
// (with 16, PMC showed SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.33).
//
// Grid: x = split of the position range (64-position chunks), y = (co tile, ci tile, tap).  Block = 4 waves (2x2), tile
// BC x BC channels of one tap; the next chunk's global loads are issued before the MFMA block of the current one.
#include "igemm_common.h"

struct IgemmWgradArgs {
    IgemmGeom g;
    const bf16_t* x;       // source pixels (pitch in_pitch)
    const bf16_t* dy;      // target pixels (pitch out_pitch)
    float* dw;             // [Co][wt_taps][Ci] fp32, accumulated
    float* db;             // optional [Co] fp32: += column sums of dY (the bias gradient of an nn.Linear), taken from the dY tiles
                           // the ci-tile-0 / tap-0 workgroups stage anyway
    float* part;           // splits > 1: slabs [splits][slab], slab = Co*wt_taps*Ci (+ Co when db != null) floats
    long slab;
    int splits;
    int chunks_per_block;  // 64-position chunks each block reduces
};

__device__ __forceinline__ bf16x4 lds_tr_read(const bf16_t* addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(addr));
}

// fragment for MFMA 32x32x16: lane l supplies matrix row (l&31) = channel, k = (l>>5)*8 .. +7 = positions.
template <bool USE_TR, int PITCH>
__device__ __forceinline__ bf16x8 load_frag_T(const bf16_t* tile, int ch0, int pos0, int lane) {
    bf16x8 f;
    if (USE_TR) {
        // 16-lane group gq: channels ch0 + (gq&1)*16 .., positions pos0 + (gq>>1)*8 ..; lane s of the group addresses
        // row (s>>2) of a [4 pos][16 ch] block at channel sub-block (s&3)*4 and receives channel column s.
        const int gq = lane >> 4, s = lane & 15;
        const bf16_t* base = tile + (pos0 + (gq >> 1) * 8 + (s >> 2)) * PITCH + ch0 + (gq & 1) * 16 + (s & 3) * 4;
        const bf16x4 lo = lds_tr_read(base);
        const bf16x4 hi = lds_tr_read(base + 4 * PITCH);
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    } else {
        const bf16_t* base = tile + (pos0 + (lane >> 5) * 8) * PITCH + ch0 + (lane & 31);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = (short)base[k * PITCH];
    }
    return f;
}

template <bool USE_TR, int BC>
__global__ __launch_bounds__(256) void k_igemm_wgrad(const IgemmWgradArgs p) {
    constexpr int PITCH = BC + 32;          // bf16 elements per LDS row: 48 (BC 64) / 80 (BC 128) banks = 16 * odd, see below
    constexpr int CPR = BC / 8;             // 16-byte chunks per row
    constexpr int NL = CPR / 8;             // loads per (operand, row) per thread: thread covers chunks c, c+8, ...
    constexpr int WT = BC / 2, TT = WT / 32;  // wave tile edge, 32x32 MFMA tiles per edge
    __shared__ __attribute__((aligned(16))) bf16_t sm[2 * 64 * PITCH];
    bf16_t* sY = sm;
    bf16_t* sX = sm + 64 * PITCH;
    const IgemmGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co_tiles = (g.Co + BC - 1) / BC, ci_tiles = (g.Ci + BC - 1) / BC;
    int rest = blockIdx.y;
    const int cot = rest % co_tiles; rest /= co_tiles;
    const int cit = rest % ci_tiles; rest /= ci_tiles;
    const int t = rest;
    const int co0 = cot * BC, ci0 = cit * BC;
    const int wco = (wave >> 1) * WT, wci = (wave & 1) * WT;
    int dyt = 0, dxt = 0, tw = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)          // compile-time indices only (see igemm_common.h)
        if (i == t) { dyt = g.dy[i]; dxt = g.dx[i]; tw = g.tw[i]; }
    const int chunk = tid & 7, r0 = tid >> 3;

    f32x16 acc[TT][TT];
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int c_begin = blockIdx.x * p.chunks_per_block;
    int c_end = c_begin + p.chunks_per_block;
    const int total_chunks = (g.M + 63) / 64;
    if (c_end > total_chunks) c_end = total_chunks;

    u32x4 vy[2][NL], vx[2][NL];
    unsigned ld_ok = 0;        // bits: (row i, load l) of dY at i*NL+l, of X at 8 + i*NL+l
    auto load_chunk = [&](int c) {
        ld_ok = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = c * 64 + r0 + 32 * i;
            const bool ok_m = m < g.M;
            int n, a, b;
            decode_pos(g, ok_m ? m : 0, n, a, b);
            const long opix = ((long)n * g.Ho + (a * g.OS + g.oy0)) * g.Wo + (b * g.OS + g.ox0);
            const int iy = a * g.S + dyt, ix = b * g.S + dxt;
            const bool okp = ok_m && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
            const long ipix = okp ? ((long)n * g.Hi + iy) * g.Wi + ix : 0;
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const int cy = co0 + (chunk + 8 * l) * 8, cx = ci0 + (chunk + 8 * l) * 8;
                const bool oky = ok_m && cy < g.Co, okx = okp && cx < g.Ci;
                vy[i][l] = *reinterpret_cast<const u32x4*>(p.dy + (oky ? opix * g.out_pitch + cy : 0));
                vx[i][l] = *reinterpret_cast<const u32x4*>(p.x + (okx ? ipix * g.in_pitch + cx : 0));
                ld_ok |= (oky ? 1u : 0u) << (i * NL + l);
                ld_ok |= (okx ? 1u : 0u) << (8 + i * NL + l);
            }
        }
    };
    const bool do_bias = p.db != nullptr && cit == 0 && t == 0;
    float bsum[NL][8];
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int k = 0; k < 8; ++k) bsum[l][k] = 0.f;
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const bool oky = (ld_ok >> (i * NL + l)) & 1u, okx = (ld_ok >> (8 + i * NL + l)) & 1u;
                u32x4 y = vy[i][l], x = vx[i][l];
                y.x = oky ? y.x : 0u; y.y = oky ? y.y : 0u; y.z = oky ? y.z : 0u; y.w = oky ? y.w : 0u;
                if (do_bias) {
                    float f[8];
                    unpack8(y, f);
#pragma unroll
                    for (int k = 0; k < 8; ++k) bsum[l][k] += f[k];
                }
                x.x = okx ? x.x : 0u; x.y = okx ? x.y : 0u; x.z = okx ? x.z : 0u; x.w = okx ? x.w : 0u;
                *reinterpret_cast<u32x4*>(sY + (r0 + 32 * i) * PITCH + (chunk + 8 * l) * 8) = y;
                *reinterpret_cast<u32x4*>(sX + (r0 + 32 * i) * PITCH + (chunk + 8 * l) * 8) = x;
            }
    };

    if (c_begin < c_end) load_chunk(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
        __syncthreads();                           // previous chunk's fragment reads are done
        store_chunk();
        __syncthreads();
        if (c + 1 < c_end) load_chunk(c + 1);      // in flight while this chunk is contracted
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 fa[TT], fb[TT];
#pragma unroll
            for (int i = 0; i < TT; ++i) fa[i] = load_frag_T<USE_TR, PITCH>(sY, wco + i * 32, ks * 16, lane);
#pragma unroll
            for (int j = 0; j < TT; ++j) fb[j] = load_frag_T<USE_TR, PITCH>(sX, wci + j * 32, ks * 16, lane);
#pragma unroll
            for (int i = 0; i < TT; ++i)
#pragma unroll
                for (int j = 0; j < TT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    if (do_bias) {          // threads with the same `chunk` (tid & 7) hold the same channels: reduce over r0 through LDS
        __syncthreads();
        float* sred = reinterpret_cast<float*>(sm);          // [256][8*NL] floats <= the tile buffers
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int k = 0; k < 8; ++k) sred[tid * (8 * NL) + l * 8 + k] = bsum[l][k];
        __syncthreads();
        if (tid < BC) {
            const int grp = tid >> 3, k = tid & 7;            // channel tid of the tile = 16-byte group grp, element k
            const int chk = grp & 7, l = grp >> 3;
            float s = 0.f;
            for (int r = 0; r < 32; ++r) s += sred[(r * 8 + chk) * (8 * NL) + l * 8 + k];
            if (co0 + tid < g.Co) {
                if (p.splits > 1) p.part[(long)blockIdx.x * p.slab + (long)g.Co * g.wt_taps * g.Ci + co0 + tid] = s;
                else p.db[co0 + tid] += s;
            }
        }
        __syncthreads();
    }
    // D[row = co][col = ci]: one writer per element — slab `blockIdx.x` (plain store) or, without a split, dW itself
    float* dst = p.splits > 1 ? p.part + (long)blockIdx.x * p.slab : p.dw;
    const bool direct = p.splits <= 1;
#pragma unroll
    for (int j = 0; j < TT; ++j) {
        const int ci = ci0 + wci + j * 32 + (lane & 31);
        if (ci >= g.Ci) continue;
#pragma unroll
        for (int i = 0; i < TT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < g.Co) {
                    float* d = dst + ((long)co * g.wt_taps + tw) * g.Ci + ci;
                    *d = direct ? *d + acc[i][j][r] : acc[i][j][r];
                }
            }
    }
}

struct WgradPlan { int bc, splits, chunks_per_block, tasks; long slab; };

static WgradPlan wgrad_plan(int M, int Co, int Ci, int ntaps, int wt_taps, bool bias) {
    WgradPlan pl;
    const int tasks128 = ((Co + 127) / 128) * ((Ci + 127) / 128) * ntaps;
    pl.bc = (Co >= 128 && Ci >= 128 && tasks128 >= 36) ? 128 : 64;
    const int BC = pl.bc;
    pl.tasks = ((Co + BC - 1) / BC) * ((Ci + BC - 1) / BC) * ntaps;
    const int total_chunks = (M + 63) / 64;
    const int target_env = svsr_tune_get(SVSR_TUNE_WG_BLOCKS);
    const int target_blocks = target_env > 0 ? target_env : (BC == 128 ? 512 : 1024);    // measured optimum per tile size
    int splits = (target_blocks + pl.tasks - 1) / pl.tasks;   // every split costs a slab of Co*taps*Ci floats written and re-read
    if (splits > total_chunks) splits = total_chunks;
    if (splits < 1) splits = 1;
    pl.chunks_per_block = (total_chunks + splits - 1) / splits;
    // keep the epilogue amortised: >= 12 K-chunks per block when the problem has them (LRS linears: 2,400 rows =
    // 38 chunks -> 3 splits measured best), >= 4 for the short LRW sequences
    const int min_chunks = total_chunks >= 36 ? 12 : 4;
    if (pl.chunks_per_block < min_chunks && total_chunks >= min_chunks) pl.chunks_per_block = min_chunks;
    pl.splits = (total_chunks + pl.chunks_per_block - 1) / pl.chunks_per_block;
    pl.slab = (long)Co * wt_taps * Ci + (bias ? Co : 0);
    return pl;
}

template <bool USE_TR, int BC>
static int launch_wgrad(IgemmWgradArgs& a, const WgradPlan& pl, hipStream_t stream) {
    hipLaunchKernelGGL((k_igemm_wgrad<USE_TR, BC>), dim3(pl.splits, pl.tasks), dim3(256), 0, stream, a);
    return svsr_check_launch();
}

/* svsr_igemm_wgrad_plan: tile edge, number of K splits and the workspace (floats) svsr_igemm_wgrad needs for this shape
 * (0 floats when a single split writes dW directly). */
extern "C" int svsr_igemm_wgrad_plan(int M, int Co, int Ci, int ntaps, int wt_taps, int has_bias, int* bc, int* splits, int64_t* part_floats) {
    if (M <= 0 || Co <= 0 || Ci <= 0 || ntaps < 1 || wt_taps < ntaps) return SVSR_ERR_ARG;
    const WgradPlan pl = wgrad_plan(M, Co, Ci, ntaps, wt_taps, has_bias != 0);
    if (bc) *bc = pl.bc;
    if (splits) *splits = pl.splits;
    if (part_floats) *part_floats = pl.splits > 1 ? (int64_t)pl.splits * pl.slab : 0;
    return SVSR_OK;
}

extern "C" int svsr_colsum_rows(const float* ws, int nrows, int64_t ld, float* out0, int64_t n0, float* out1, int64_t n1, int accumulate,
                                float scale, hipStream_t stream);

extern "C" int svsr_igemm_wgrad(const void* x, const void* dyp, float* dw, int Nimg, int Hi, int Wi, int Ci, int in_pitch, int Co,
                                int Ho, int Wo, int out_pitch, int Ha, int Wa, int S, int OS, int oy0, int ox0, int ntaps, int wt_taps,
                                const int* dy, const int* dx, const int* tw, int use_tr, float* dbias, float* part, int64_t part_floats,
                                hipStream_t stream) {
    IgemmWgradArgs a;
    int rc = fill_geom(a.g, Nimg, Hi, Wi, Ci, in_pitch, Co, Ho, Wo, out_pitch, Ha, Wa, S, OS, oy0, ox0, ntaps, wt_taps, dy, dx, tw);
    if (rc != SVSR_OK) return rc;
    if (out_pitch % 8 != 0) return SVSR_ERR_ARG;
    const WgradPlan pl = wgrad_plan(a.g.M, Co, Ci, ntaps, wt_taps, dbias != nullptr);
    if (pl.splits > 1 && (part == nullptr || part_floats < (int64_t)pl.splits * pl.slab)) return SVSR_ERR_ARG;
    a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dyp; a.dw = dw; a.db = dbias;
    a.part = part; a.slab = pl.slab; a.splits = pl.splits; a.chunks_per_block = pl.chunks_per_block;
    if (use_tr) rc = pl.bc == 128 ? launch_wgrad<true, 128>(a, pl, stream) : launch_wgrad<true, 64>(a, pl, stream);
    else rc = pl.bc == 128 ? launch_wgrad<false, 128>(a, pl, stream) : launch_wgrad<false, 64>(a, pl, stream);
    if (rc != SVSR_OK || pl.splits <= 1) return rc;
    const int64_t n = (int64_t)Co * wt_taps * Ci;
    return svsr_colsum_rows(part, pl.splits, pl.slab, dw, n, dbias, dbias != nullptr ? Co : 0, 1, 1.0f, stream);
}
