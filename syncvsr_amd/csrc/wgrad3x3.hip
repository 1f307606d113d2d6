// Weight gradient of a 3x3 / stride-1 / pad-1 NHWC convolution with the nine taps sharing one pass over the
// activations (gfx950):     dW[co][kh][kw][ci] += sum_{n,y,x} dY[n,y,x,co] * X[n,y+kh-1,x+kw-1,ci]
// (autograd of the stride-1 conv3x3 of the ResNet18 trunk, reference LRW/video/src/tcn/models/resnet.py:8-10,59-72;
// SURVEY.md §8 a16).
//
// The generic kernel (igemm_wgrad.hip) runs one tap per workgroup, so every activation and every output gradient is
// re-read nine times; at layer1 sizes that is ~1 GB of L2->LDS traffic per convolution and bounds it at ~7 TB/s.  Here the
// reduction runs over ZERO-PADDED pixel coordinates q = (n, y', x') of a (H+2) x (W+2) grid flattened over the batch:
// in that space tap (kh,kw) is the constant row shift kh*(W+2)+kw, pad pixels carry dY = 0, and a contiguous chunk of 128
// positions needs one dY tile [128][64 co] and ONE X tile [128 + 2(W+3)][64 ci] for all nine taps.  Per 16-position
// step a wave reads one dY^T fragment and nine shifted X^T fragments (ds_read_b64_tr_b16) and issues nine MFMAs into
// nine 32x32 accumulators.  The next chunk's rows are fetched into registers while the current one is contracted.
// Split-K without atomics: split s stores its nine tiles into slab s of a caller-owned workspace and svsr_colsum_rows
// (runtime.hip) adds the slabs into dW in a fixed order (reproducible gradients).

#include "common.h"

#define W3_CH 128         // padded positions per chunk
#define W3_PITCH 96        // bf16 elements per LDS row: 192 B = 48 banks, so the 4 rows a 32-lane ds_read_b64_tr_b16 group touches
                           // start at banks 0/48/32/16 and their 16-bank windows are disjoint (pitch 80 = 40 banks made row 3 wrap
                           // onto row 0: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.45)
#define W3_MAXXR 192       // max X-tile rows: 128 + 2*(WP+1), i.e. W <= 29
#define W3D_MAXXR 256      // dense form (below): 128 REAL positions span up to this many padded rows (+ halo); checked per shape on the host

struct Wgrad3Args {
    const bf16_t* x;       // [Nimg][H][W][Ci]
    const bf16_t* dy;      // [Nimg][H][W][Co]
    float* dw;             // [Co][9][Ci] fp32, accumulated
    float* part;           // splits > 1: slabs [splits][Co*9*Ci]
    int splits;
    int Nimg, H, W, Ci, Co;
    int WP, Q, Qtot, XR;   // padded row length, padded pixels per image, total, X-tile rows
    float inv_q, inv_wp;
    int chunks_per_block, total_chunks;
    int P, Ptot;           // dense form: real pixels per image, in all
};

// Slab format: a workgroup's nine 64 x 64 tiles leave in MFMA FRAGMENT layout — float4 group ((tap * 4 + wave) * 4 + rq) * 64 + lane holds
// accumulator registers rq*4 .. rq*4+3 — as 36 16-byte stores per lane (1 KiB per wave-instruction) instead of 144 dword stores of two
// 128-byte rows each; k_wgrad3_reduce adds a task's slabs in split order (64 float4 groups x 4 slot lanes per block) and scatters the sum
// into dW once.  Up to four convolutions of the same geometry can share a launch (blockIdx.z): the launch's workgroups are divided among
// them, so each writes 1/n of the slabs of a launch of its own (grid.z = n; the shipped entry point launches n = 1).
// (Measured and dropped in round 4: ONE 8-wave workgroup per CU whose two 4-wave groups take alternate CHUNKS one barrier apart, accumulators
// merged through LDS — half the slabs, but 94 vs 78 us per launch: a group's wait for its next chunk's rows, 2-3 us from HBM, stalls the shared
// barrier for both groups.  The 8-wave form that is the default since round 5 splits the TAPS instead: both groups work on the same chunk.)
struct Wgrad3Multi { const bf16_t* x[4]; const bf16_t* dy[4]; float* dw[4]; };
#define W3_TILE_FLOATS (9 * 64 * 64)

__device__ __forceinline__ bf16x8 w3_frag_T(const bf16_t* tile, int ch0, int pos0, int lane) {
    // MFMA 32x32x16 fragment: lane l <- channel ch0 + (l&31), positions pos0 + (l>>5)*8 .. +7, via two transpose reads
    const int gq = lane >> 4, s = lane & 15;
    const bf16_t* base = tile + (pos0 + (gq >> 1) * 8 + (s >> 2)) * W3_PITCH + ch0 + (gq & 1) * 16 + (s & 3) * 4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(base));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(base + 4 * W3_PITCH));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// the same fragment with the two row groups of a lane given as rows (gathered positions: the dense form)
__device__ __forceinline__ bf16x8 w3_frag_T2(const bf16_t* tile, int ch0, int rowa, int rowb, int lane) {
    const int col = ch0 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(tile + rowa * W3_PITCH + col));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(tile + rowb * W3_PITCH + col));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// padded index q -> pixel index n*H*W + (y'-1)*W + (x'-1), or -1 for pad pixels / outside the batch
__device__ __forceinline__ long w3_pixel(const Wgrad3Args& p, int q) {
    if (q < 0 || q >= p.Qtot) return -1;
    int n = (int)((float)q * p.inv_q);
    int rem = q - n * p.Q;
    if (rem < 0) { n--; rem += p.Q; } else if (rem >= p.Q) { n++; rem -= p.Q; }
    int yp = (int)((float)rem * p.inv_wp);
    int xp = rem - yp * p.WP;
    if (xp < 0) { yp--; xp += p.WP; } else if (xp >= p.WP) { yp++; xp -= p.WP; }
    if (yp < 1 || yp > p.H || xp < 1 || xp > p.W) return -1;
    return ((long)n * p.H + (yp - 1)) * p.W + (xp - 1);
}

// NW = 4: four waves = the four 32 x 32 quadrants of the 64 x 64 tile, nine taps each (144 accumulator registers; two workgroups per CU).
// NW = 8 (round 5): eight waves, one workgroup per CU — waves 0-3 take taps 0-4 of their quadrant, waves 4-7 taps 5-8: all eight waves work on
// the SAME chunk (no merge of accumulators: different taps are different outputs), a workgroup covers twice the positions, so a launch writes
// HALF the slabs, and a wave needs ~130 registers instead of 247 (two 4-wave workgroups held 496 of a SIMD's 512 registers).
// DENSE (round 6): the contraction runs over the REAL pixels only.  In the padded walk above every pad pixel is a zero row of dY that is
// multiplied all the same: 13 x 13 / 11 x 11 = 1.40x the MFMAs and fragment reads at layer2, 24 x 24 / 22 x 22 = 1.19x at layer1 (the profile
// showed it: matrix pipe 32 % busy for 19 % of the peak delivered).  Here a chunk is 128 consecutive real pixels: the dY tile is dense, the X tile
// is still the padded stretch those pixels span (+ halo; pad rows are zeros), and a small LDS table maps position k of the chunk to its padded row
// — ds_read_b64_tr_b16 takes a row address per lane, so the nine shifted fragments are gathered through that table instead of walked.
template <int NW, bool DENSE>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void k_wgrad3x3_halo(const Wgrad3Args p, const Wgrad3Multi m) {
    constexpr int NT = NW * 64, RPP = NT / 8;                    // threads; tile rows staged per pass of all threads
    constexpr int YR = W3_CH / RPP, XRN = (DENSE ? W3D_MAXXR : W3_MAXXR) / RPP;        // dY / X rows per thread and chunk
    constexpr int TPW = NW == 4 ? 9 : 5;                         // taps per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sY = reinterpret_cast<bf16_t*>(smem_raw);            // [128][PITCH]
    bf16_t* sX = sY + W3_CH * W3_PITCH;                          // [XR][PITCH]
    int* sMap = reinterpret_cast<int*>(sX + (DENSE ? W3D_MAXXR : 0) * W3_PITCH);      // dense form: padded row (relative to the chunk's first pixel) of position k
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci_tiles = p.Ci >> 6;
    const int cot = blockIdx.y / ci_tiles, cit = blockIdx.y - cot * ci_tiles;
    const int co0 = cot * 64, ci0 = cit * 64;
    const int wq = wave & 3, tap0 = NW == 4 ? 0 : (wave >> 2) * 5;
    const int wco = (wq >> 1) * 32, wci = (wq & 1) * 32;
    const int chunk = tid & 7, r0 = tid >> 3;
    const bf16_t* gx = m.x[blockIdx.z];
    const bf16_t* gy = m.dy[blockIdx.z];

    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    int shift[TPW];                                              // row shift of this wave's taps (kh * WP + kw); wave-uniform
#pragma unroll
    for (int t = 0; t < TPW; ++t) { const int tt = tap0 + t < 9 ? tap0 + t : 8; shift[t] = (tt / 3) * p.WP + tt % 3; }

    const int c_begin = blockIdx.x * p.chunks_per_block;
    int c_end = c_begin + p.chunks_per_block;
    if (c_end > p.total_chunks) c_end = p.total_chunks;

    // padded index of real pixel g (n * Q + (y + 1) * WP + x + 1); exact for the sizes the launcher admits (< 2^24 pixels)
    auto padq = [&](int g) {
        const int n = g / p.P, rem = g - n * p.P;
        const int y = rem / p.W, x = rem - y * p.W;
        return n * p.Q + (y + 1) * p.WP + x + 1;
    };
    // one chunk's rows on their way from global memory to LDS: dY / X pieces of this thread, which of them are real, and (dense form) the padded
    // index of the chunk's first pixel and the rows of its X tile
    struct Staged { u32x4 vy[YR], vx[XRN]; unsigned ok; int q0, xr; };
    auto load_chunk = [&](Staged& g, int c) {
        const int q0 = c * W3_CH;
        g.ok = 0; g.q0 = q0; g.xr = p.XR;
        if (DENSE) {
            int plast = q0 + W3_CH - 1;
            if (plast > p.Ptot - 1) plast = p.Ptot - 1;
            g.q0 = padq(q0);
            g.xr = padq(plast) - g.q0 + 1 + 2 * (p.WP + 1);
        }
#pragma unroll
        for (int i = 0; i < YR; ++i) {
            long pix;
            if (DENSE) { const int px = q0 + r0 + RPP * i; pix = px < p.Ptot ? px : -1; }
            else pix = w3_pixel(p, q0 + r0 + RPP * i);
            const bool ok = pix >= 0;
            g.vy[i] = *reinterpret_cast<const u32x4*>(gy + (ok ? pix * p.Co + co0 + chunk * 8 : 0));
            g.ok |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < XRN; ++i) {
            const int rr = r0 + RPP * i;
            const long pix = rr < g.xr ? w3_pixel(p, g.q0 - (p.WP + 1) + rr) : -1;
            const bool ok = pix >= 0;
            g.vx[i] = *reinterpret_cast<const u32x4*>(gx + (ok ? pix * p.Ci + ci0 + chunk * 8 : 0));
            g.ok |= (ok ? 1u : 0u) << (8 + i);
        }
    };
    auto store_chunk = [&](const Staged& g, int c, bf16_t* bY, bf16_t* bX, int* bMap) {
#pragma unroll
        for (int i = 0; i < YR; ++i) {
            const bool ok = (g.ok >> i) & 1u;
            u32x4 v = g.vy[i];
            v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
            *reinterpret_cast<u32x4*>(bY + (r0 + RPP * i) * W3_PITCH + chunk * 8) = v;
        }
        if (DENSE && tid < W3_CH) {         // position k of the chunk -> its padded row inside the X tile (without the tap shift)
            int px = c * W3_CH + tid;
            if (px > p.Ptot - 1) px = p.Ptot - 1;            // (rows past the last pixel: dY is zero there, any readable row will do)
            bMap[tid] = padq(px) - g.q0;
        }
#pragma unroll
        for (int i = 0; i < XRN; ++i) {
            const int rr = r0 + RPP * i;
            const bool ok = (g.ok >> (8 + i)) & 1u;
            u32x4 v = g.vx[i];
            v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
            if (rr < g.xr) *reinterpret_cast<u32x4*>(bX + rr * W3_PITCH + chunk * 8) = v;
        }
    };
    auto contract = [&](const bf16_t* bY, const bf16_t* bX, const int* bMap) {
#pragma unroll 2
        for (int ks = 0; ks < W3_CH / 16; ++ks) {
            const bf16x8 fa = w3_frag_T(bY, wco, ks * 16, lane);
            int rowa = 0, rowb = 0;
            if (DENSE) {                           // this lane's two positions of the step (w3_frag_T's row pattern), as padded rows
                const int k0 = ks * 16 + ((lane >> 4) >> 1) * 8 + ((lane & 15) >> 2);
                rowa = bMap[k0]; rowb = bMap[k0 + 4];
            }
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                if (NW == 8 && t == TPW - 1 && tap0 + t >= 9) break;           // the second wave group has four taps (wave-uniform)
                const bf16x8 fb = DENSE ? w3_frag_T2(bX, wci, rowa + shift[t], rowb + shift[t], lane) : w3_frag_T(bX, wci, ks * 16 + shift[t], lane);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[t], 0, 0, 0);
            }
        }
    };

    {
        // (Measured in round 6 and dropped: two LDS tiles + two chunks in flight for the 8-wave form — one barrier per chunk, the store pass of chunk
        // c + 1 beside the contraction of chunk c: 73 -> 78 us at layer1, 72 -> 76 at layer2.  A chunk's ~4 us are not a load round trip: they are its
        // 384 KiB of fragment reads (1.2 KiB per MFMA: every X fragment feeds ONE MFMA) next to 1.2 us of MFMA time.)
        Staged g;
        if (c_begin < c_end) load_chunk(g, c_begin);
        for (int c = c_begin; c < c_end; ++c) {
            __syncthreads();                           // previous chunk's fragment reads are done
            store_chunk(g, c, sY, sX, sMap);
            __syncthreads();
            if (c + 1 < c_end) load_chunk(g, c + 1);   // in flight while this chunk is contracted
            contract(sY, sX, sMap);
        }
    }
    if (p.splits > 1) {
        f32x4* dst = reinterpret_cast<f32x4*>(p.part + (((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * W3_TILE_FLOATS);
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            if (tap0 + t >= 9) break;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                dst[(((tap0 + t) * 4 + wq) * 4 + rq) * 64 + lane] = f32x4{acc[t][rq * 4], acc[t][rq * 4 + 1], acc[t][rq * 4 + 2], acc[t][rq * 4 + 3]};
        }
        return;
    }
    // single split: D[row = co][col = ci] added to dW directly (one writer per element)
    const int ci = ci0 + wci + (lane & 31);
    float* dwp = m.dw[blockIdx.z];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        if (tap0 + t >= 9) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float* d = dwp + ((long)co * 9 + tap0 + t) * p.Ci + ci;
            *d += acc[t][r];
        }
    }
}

// dW[co][t][ci] += sum over the splits (in split order) of a task's tiles.  grid (9 * 16 blocks of 64 float4 groups, tasks, problems)
__global__ __launch_bounds__(256) void k_wgrad3_reduce(const float* __restrict__ part, const Wgrad3Multi m, int splits, int Ci, int ci_tiles) {
    __shared__ f32x4 sred[4][64];
    const int gl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int g = blockIdx.x * 64 + gl;
    const float* src = part + ((long)blockIdx.z * gridDim.y + blockIdx.y) * splits * W3_TILE_FLOATS;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    int s = sl;
    // eight loads in flight per thread, added in the lane's split order (two in flight made a lane's 64 slabs a chain of 32 dependent round
    // trips: 18 us alone, 30-130 us beside the main stream's HBM-bound passes)
    for (; s + 28 < splits; s += 32) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = reinterpret_cast<const f32x4*>(src + (long)(s + 4 * k) * W3_TILE_FLOATS)[g];
#pragma unroll
        for (int k = 0; k < 8; ++k) a = a + v[k];
    }
    for (; s + 4 < splits; s += 8) {
        const f32x4 v0 = reinterpret_cast<const f32x4*>(src + (long)s * W3_TILE_FLOATS)[g];
        const f32x4 v1 = reinterpret_cast<const f32x4*>(src + (long)(s + 4) * W3_TILE_FLOATS)[g];
        a = (a + v0) + v1;
    }
    for (; s < splits; s += 4) a = a + reinterpret_cast<const f32x4*>(src + (long)s * W3_TILE_FLOATS)[g];
    sred[sl][gl] = a;
    __syncthreads();
    if (sl != 0) return;
    a = ((sred[0][gl] + sred[1][gl]) + sred[2][gl]) + sred[3][gl];
    const int lane = g & 63, rq = (g >> 6) & 3, wave = (g >> 8) & 3, t = g >> 10;
    const int cot = blockIdx.y / ci_tiles, cit = blockIdx.y - cot * ci_tiles;
    const int ci = cit * 64 + (wave & 1) * 32 + (lane & 31);
    const int co = cot * 64 + (wave >> 1) * 32 + 8 * rq + 4 * (lane >> 5);
    float* dw = m.dw[blockIdx.z];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float* d = dw + ((long)(co + k) * 9 + t) * Ci + ci;
        *d += a[k];
    }
}

struct W3Plan { int splits, chunks_per_block, total_chunks, tasks, dense; };

// dense form (k_wgrad3x3_halo<NW, true>): do 128 consecutive real pixels, wherever they start, span at most W3D_MAXXR padded rows with the halo?
static bool w3_dense_ok(int H, int W) {
    if (svsr_tune_get(SVSR_TUNE_W3_DENSE) == 0) return false;
    const int WP = W + 2, Q = (H + 2) * WP, P = H * W;
    auto padq = [&](int g) { const int n = g / P, rem = g - n * P; return n * Q + (rem / W + 1) * WP + rem % W + 1; };
    int worst = 0;
    for (int s0 = 0; s0 < P; ++s0) { const int d = padq(s0 + W3_CH - 1) - padq(s0); if (d > worst) worst = d; }
    return worst + 1 + 2 * (WP + 1) <= W3D_MAXXR;
}

static W3Plan w3_plan(int Nimg, int H, int W, int Ci, int Co, int nprob = 1) {
    W3Plan pl;
    pl.dense = w3_dense_ok(H, W) ? 1 : 0;
    const long qtot = pl.dense ? (long)Nimg * H * W : (long)Nimg * (H + 2) * (W + 2);
    pl.total_chunks = (int)((qtot + W3_CH - 1) / W3_CH);
    pl.tasks = (Co / 64) * (Ci / 64);
    int target_blocks = svsr_tune_get(SVSR_TUNE_W3_BLOCKS);      // one round of 2 workgroups per CU (4-wave form) / 1 per CU (8-wave form), shared by the problems of the launch
    target_blocks = (target_blocks > 0 ? target_blocks : 512) / (nprob > 0 ? nprob : 1);
    if (svsr_tune_get(SVSR_TUNE_W3_WAVES) == 8) target_blocks /= 2;
    int splits = (target_blocks + pl.tasks - 1) / pl.tasks;       // every split costs a slab written and re-read
    if (splits > pl.total_chunks) splits = pl.total_chunks;
    if (splits < 1) splits = 1;
    pl.chunks_per_block = (pl.total_chunks + splits - 1) / splits;
    pl.splits = (pl.total_chunks + pl.chunks_per_block - 1) / pl.chunks_per_block;
    return pl;
}

/* workspace (floats) svsr_conv3x3_wgrad needs for this shape */
extern "C" int svsr_conv3x3_wgrad_plan(int Nimg, int H, int W, int Ci, int Co, int* splits, int64_t* part_floats) {
    if (Ci % 64 || Co % 64 || Ci <= 0 || Co <= 0 || H < 1 || W < 1 || Nimg < 1) return SVSR_ERR_ARG;
    const W3Plan pl = w3_plan(Nimg, H, W, Ci, Co);
    if (splits) *splits = pl.splits;
    if (part_floats) *part_floats = pl.splits > 1 ? (int64_t)pl.splits * pl.tasks * W3_TILE_FLOATS : 0;
    return SVSR_OK;
}

extern "C" int svsr_colsum_rows(const float* ws, int nrows, int64_t ld, float* out0, int64_t n0, float* out1, int64_t n1, int accumulate,
                                float scale, hipStream_t stream);

static int w3_fill_args(Wgrad3Args& a, int Nimg, int H, int W, int Ci, int Co) {
    if (Ci % 64 || Co % 64 || Ci <= 0 || Co <= 0 || W + 2 > (W3_MAXXR - W3_CH) / 2 - 1 || H < 1 || W < 1 || Nimg < 1) return SVSR_ERR_ARG;
    a.Nimg = Nimg; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
    a.WP = W + 2; a.Q = (H + 2) * (W + 2);
    const long qtot = (long)Nimg * a.Q;
    if (qtot >= (1L << 24)) return SVSR_ERR_ARG;
    a.Qtot = (int)qtot;
    a.XR = W3_CH + 2 * (a.WP + 1);
    a.inv_q = 1.0f / (float)a.Q; a.inv_wp = 1.0f / (float)a.WP;
    a.P = H * W; a.Ptot = Nimg * H * W;
    return SVSR_OK;
}

static int w3_launch(const Wgrad3Args& a0, const Wgrad3Multi& m, int n, const W3Plan& pl, float* part, int64_t part_floats, hipStream_t stream) {
    if (pl.splits > 1 && (part == nullptr || part_floats < (int64_t)n * pl.splits * pl.tasks * W3_TILE_FLOATS)) return SVSR_ERR_ARG;
    Wgrad3Args a = a0;
    a.total_chunks = pl.total_chunks; a.chunks_per_block = pl.chunks_per_block; a.splits = pl.splits; a.part = part;
    const bool w8 = svsr_tune_get(SVSR_TUNE_W3_WAVES) == 8;
    const size_t lds = pl.dense ? (size_t)(W3_CH + W3D_MAXXR) * W3_PITCH * sizeof(bf16_t) + W3_CH * sizeof(int)
                                : (size_t)(W3_CH + a.XR) * W3_PITCH * sizeof(bf16_t);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad3x3_halo<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad3x3_halo<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad3x3_halo<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad3x3_halo<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_set = lds;
    }
    if (w8 && pl.dense) hipLaunchKernelGGL((k_wgrad3x3_halo<8, true>), dim3(pl.splits, pl.tasks, n), dim3(512), lds, stream, a, m);
    else if (w8) hipLaunchKernelGGL((k_wgrad3x3_halo<8, false>), dim3(pl.splits, pl.tasks, n), dim3(512), lds, stream, a, m);
    else if (pl.dense) hipLaunchKernelGGL((k_wgrad3x3_halo<4, true>), dim3(pl.splits, pl.tasks, n), dim3(256), lds, stream, a, m);
    else hipLaunchKernelGGL((k_wgrad3x3_halo<4, false>), dim3(pl.splits, pl.tasks, n), dim3(256), lds, stream, a, m);
    int rc = svsr_check_launch();
    if (rc != SVSR_OK || pl.splits <= 1) return rc;
    hipLaunchKernelGGL(k_wgrad3_reduce, dim3(W3_TILE_FLOATS / 4 / 64, pl.tasks, n), dim3(256), 0, stream, (const float*)part, m, pl.splits, a.Ci, a.Ci >> 6);
    return svsr_check_launch();
}

extern "C" int svsr_conv3x3_wgrad(const void* x, const void* dy, float* dw, int Nimg, int H, int W, int Ci, int Co, float* part,
                                  int64_t part_floats, hipStream_t stream) {
    Wgrad3Args a;
    const int rc0 = w3_fill_args(a, Nimg, H, W, Ci, Co);
    if (rc0 != SVSR_OK) return rc0;
    a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.dw = dw;
    Wgrad3Multi m{};
    m.x[0] = a.x; m.dy[0] = a.dy; m.dw[0] = dw;
    return w3_launch(a, m, 1, w3_plan(Nimg, H, W, Ci, Co), part, part_floats, stream);
}
