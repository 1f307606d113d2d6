// Persistent 8-wave implicit-GEMM contraction for the trunk's stride-1 3x3 convolutions, forward and data gradient (gfx950).
//
// Replaces, for the large shapes, the same reference calls as igemm_fwd.hip: nn.Conv2d 3x3 of resnet.layer2/3 and its
// input-gradient (reference LRW/video/src/tcn/models/resnet.py:8-16,28-72; timm twin via lightning.py:55,114-117).
//
// Why another kernel: the 4-wave 128x128 kernel is bound by LDS-DMA instructions per MFMA (64 FLOP per staged
// byte), pays a 5 us epilogue and a cold prologue per 128x128 tile, and leans on two co-resident workgroups to overlap them.  Here
//   * ONE workgroup of 8 waves per CU walks a static list of 256 x 128 tiles (85 FLOP per staged byte; A tiles shared by the N tiles
//     of an M tile sit 8 items apart, i.e. on the same XCD's L2 under round-robin placement);
//   * the K loop runs on a 3-deep LDS ring of 64-deep K tiles (3 x 48 KiB) that is CONTINUOUS across tiles: the first two K tiles of
//     the next tile are requested during the last two of the current one, so a tile starts with a full ring and its epilogue runs
//     under the DMA flight of its successor; the next tile's row table and descriptor arrive by LDS-DMA too (no ordinary global load
//     ever sits in the loop: hipcc would drain the DMA queue in front of it);
//   * every K tile is two PHASES per wave: {8 ds_read_b128 of one 32-deep half + 3 of the 6 DMA pieces of K tile +2} -> barrier ->
//     {8 MFMA 32x32x16} -> barrier.  The two groups of four waves (one wave of each per SIMD) run these phases STAGGERED by one
//     barrier, so in every barrier interval one group feeds the matrix pipe while the other issues LDS reads and DMA; counted
//     s_waitcnt vmcnt(6) once per K tile keeps a whole K tile in flight across the barriers, never 0 inside the loop.
// Hazards (barrier k pairs group 0's k-th with group 1's k-th; group 1 starts one barrier late):
//   RAW  a K tile is waited for (own pieces) in phase p before the wave's mid-phase barrier and first read in phase p+1;
//   WAR  a ring slot is re-staged from the phase AFTER its last read; reads are retired (lgkmcnt(0)) before the mid-phase barrier.
// Epilogue: accumulators -> fp32 LDS (the ring slot that just went idle) in four 64-row passes -> (+addend | BatchNorm-backward
// masks) -> 16-byte bf16 stores; BatchNorm partial rows (one per M tile) in a fixed order.  No atomics: results are reproducible.
#include <string.h>
#include <type_traits>

#include "igemm_fwd.h"

namespace {

constexpr int BM = P8_BM, BN = P8_BN, BK = 64, NS = 3;
constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, S_ELEMS = A_ELEMS + B_ELEMS;     // bf16 elements per ring slot (48 KiB)
constexpr int RING_BYTES = NS * S_ELEMS * 2;                                          // 147456
constexpr int ROWTAB_OFF = RING_BYTES;                  // int  [2][256][2]
constexpr int DESC_OFF = ROWTAB_OFF + 2 * 256 * 2 * 4;  // int  [2][8 waves][64]
constexpr int RED_OFF = DESC_OFF + 2 * 8 * 64 * 4;      // float scratch, 8 KiB
constexpr int LDS_BYTES = RED_OFF + 8192;               // 163840 = all of a CU's LDS
static_assert(LDS_BYTES == 160 * 1024, "one workgroup owns the CU's LDS");

__device__ __forceinline__ void glds16(const void* src, void* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const void* src, void* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
}

#define P8_BARRIER() asm volatile("s_barrier" ::: "memory")
#define P8_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define P8_WAIT_VM(n) SVSR_WAIT_VM(n)      /* (vmcnt(0) in the SVSR_SYNC_DEBUG build: common.h) */
#define P8_SYNC_ALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define P8_SYNC_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")       /* LDS traffic only: global stores stay in flight */

struct TileCtx {
    unsigned a_off[4];          // BYTE offset from p.in of row rr + 64 i's centre pixel + this lane's (swizzled) 16-byte chunk; the per-K-tile
    unsigned b_off[2];          // part of an address (tap shift, channel offset) is wave-uniform and lives in the scalar base.  b: weight row n0 + rr + 64 i
    int desc;                   // lane l holds descriptor word l
    int KT;                     // K tiles of the tile = taps * Ci / 64
    int m_tile, n0;
};

// work item r of workgroup w: rounds alternate direction, so the workgroups that drew the longest tiles of one round draw the
// shortest of the next (tiles are ordered by decreasing tap count)
// rev > 0 (= this workgroup's round count): the rounds are walked last to first — half the workgroups start with their SHORT tile, so
// the first epilogues of a launch (an HBM burst when they coincide) do not all fall at the same moment
__device__ __forceinline__ bool p8_item(int gy, int items, int w, int G, int r, int& m_tile, int& n0, int rev = 0, int bn = BN) {
    if (rev > 0) { if (r >= rev) return false; r = rev - 1 - r; }
    const int q = r * G + ((r & 1) ? G - 1 - w : w);
    if (q >= items) return false;
    const int grp = q / (8 * gy), rem = q - grp * 8 * gy;
    m_tile = grp * 8 + (rem & 7);
    n0 = (rem >> 3) * bn;
    return true;          // (m_tile may be a hole >= tiles_m: the caller skips it)
}

__device__ __forceinline__ bool p8_next_item(int tiles_m, int gy, int items, int w, int G, int& r, int& m_tile, int& n0, int rev = 0, int bn = BN) {
    for (;;) {
        ++r;
        if (!p8_item(gy, items, w, G, r, m_tile, n0, rev, bn)) return false;
        if (m_tile < tiles_m) return true;
    }
}

#ifdef SVSR_P8_STAMP
__device__ long long g_p8_phase[8];             // wave 0 of every workgroup adds: [0] epilogue head (statistics of the accumulators, offsets, constants), [1] row block 0, [2] row block 1
__device__ long long g_p8_stamp[2048];          // [workgroup][4]: cycles inside K loops (wave 0), K tiles walked, cycles inside epilogues, tiles — build variant "p8stamp" only
#endif

}  // namespace

// EPI 0: out = acc (+ addend), optional BatchNorm partials of the fp32 accumulators;  EPI 1: BatchNorm-backward fusion (bnb_*)
// PH: phases per K tile (2: the 32-deep halves, 8 MFMAs between barriers; 1: the whole K tile, 16 MFMAs between barriers)
// NJ: 32-column blocks of a wave's tile — 2: the 256 x 128 tile (wave tile 64 x 64); 1: a 256 x 64 tile (wave tile 64 x 32) for launches whose
// 128-wide items would leave half the CUs idle (layer4: 33 row tiles x 512 channels = 132 items of 128 columns, 264 of 64)
// WIDE (round 6, NJ = 2): the epilogue's global accesses as FULL 128-byte lines.  The patch holds 16 rows of BOTH 32-column fragments of a wave
// (fp32 [16][64], the same 4 KiB), a lane reads 8 consecutive channels of a row and stores 16 bytes: a wave instruction covers 8 rows x 128 bytes
// (its addend / x / y loads likewise) instead of 8 rows x 64 bytes.  Same arithmetic per element, same order of every sum: bit-identical results.
template <int EPI, int PH = 2, int NJ = 2, bool WIDE = false>
__global__ __launch_bounds__(512, 2) void k_igemm_p8(const IgemmFwdArgs p, int stagger) {
    constexpr int BNT = 64 * NJ, NPIECE = 4 + NJ;          // tile columns; DMA pieces per thread and K tile
    static_assert(NJ == 2 || PH == 1, "the 64-column tile has five pieces per K tile: one phase");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* ring = reinterpret_cast<bf16_t*>(smem);
    int* sRowTab = reinterpret_cast<int*>(smem + ROWTAB_OFF);
    int* sDesc = reinterpret_cast<int*>(smem + DESC_OFF);
    float* sRed = reinterpret_cast<float*>(smem + RED_OFF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;              // wave tile: rows wm*64.., columns wn*64..; wn = stagger group
    const int slot = tid & 7, rr = tid >> 3;              // DMA: lane writes LDS chunk `slot` of row rr + 64 i ...
    const int csw = slot ^ ((rr >> 1) & 7);               // ... which must hold global chunk csw (swizzle on the source)
    const int G = gridDim.x, w = blockIdx.x;
    const int* hdr = p.plan;
    const int tiles_m = hdr[1], gy = hdr[2], items = hdr[5];
    const int* descw = p.plan + hdr[3];
    const int* rowsw = p.plan + hdr[4];
    const int spt = p.Ci / BK;                            // K tiles per tap

    // ---- helpers -----------------------------------------------------------------------------------------------------------
    auto meta_dma = [&](int m_tile, int par) {            // row table (one dword per thread) + descriptor (one copy per wave)
        glds4(rowsw + (long)m_tile * 512 + tid, sRowTab + par * 512 + wave * 64);
        glds4(descw + (long)m_tile * P8_DESC_WORDS + lane, sDesc + (par * 8 + wave) * 64);
    };
    auto make_ctx = [&](TileCtx& c, int m_tile, int n0, int par) {
        c.m_tile = m_tile; c.n0 = n0;
        c.desc = sDesc[(par * 8 + wave) * 64 + lane];
        c.KT = __builtin_amdgcn_readlane(c.desc, 0) * spt;
#pragma unroll
        for (int i = 0; i < 4; ++i)       // (rows past the end of a class repeat the tile's first row: always a readable pixel; their results are dropped)
            c.a_off[i] = ((unsigned)sRowTab[par * 512 + (rr + 64 * i) * 2] * (unsigned)p.in_pitch + csw * 8) * 2u;
#pragma unroll
        for (int i = 0; i < NJ; ++i) c.b_off[i] = ((unsigned)(n0 + rr + 64 * i) * (unsigned)(p.wt_taps * p.Ci) + csw * 8) * 2u;
    };
    auto dummy_ctx = [&](TileCtx& c) {                    // past the last tile: the staging stream keeps its rhythm on a harmless source
        c.m_tile = -1; c.n0 = 0; c.desc = 0; c.KT = 1 << 30;
#pragma unroll
        for (int i = 0; i < 4; ++i) c.a_off[i] = (unsigned)(csw * 16);
        c.b_off[0] = c.b_off[1] = (unsigned)(csw * 16);
    };
    // DMA pieces [LO, HI) of one K tile (0..3: A rows rr + 64 i, 4..5: B rows) into ring slot `dst`; abase / bbase: wave-uniform
    auto stage_pieces = [&](const TileCtx& c, const char* abase, const char* bbase, bf16_t* dst, auto lo_c, auto hi_c) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
#pragma unroll
        for (int i = LO; i < HI; ++i) {
            if (i < 4) glds16(abase + c.a_off[i], dst + (wave * 8 + 64 * i) * 64);
            else glds16(bbase + c.b_off[i - 4], dst + A_ELEMS + (wave * 8 + 64 * (i - 4)) * 64);
        }
    };

    // ---- first item of this workgroup ---------------------------------------------------------------------------------------
    int rev = 0;
    if ((stagger & 2) && (w & 1)) {
        int mt, nn;
        while (p8_item(gy, items, w, G, rev, mt, nn, 0, BNT)) ++rev;      // rounds of this workgroup
    }
    int r = -1, m_tile = 0, n0 = 0;
    if (!p8_next_item(tiles_m, gy, items, w, G, r, m_tile, n0, rev, BNT)) return;
    int par = 0;
    meta_dma(m_tile, par);
    P8_SYNC_ALL();
    TileCtx cur, nxt, str;                                // str: the tile the staging stream is in (a copy of cur, later of nxt)
    make_ctx(cur, m_tile, n0, par);
    str = cur;
    // staging stream: K tile st_kt of `str`, tap st_t, channel offset st_c
    int st_kt = 0, st_t = 0, st_c = 0;
    bool st_next = false;
    auto stream_bases = [&](const char*& abase, const char*& bbase) {
        const int d = __builtin_amdgcn_readlane(str.desc, 2 + st_t), tw = __builtin_amdgcn_readlane(str.desc, 11 + st_t);
        abase = reinterpret_cast<const char*>(p.in) + ((long)d * p.in_pitch + st_c) * 2;
        bbase = reinterpret_cast<const char*>(p.wt) + ((long)tw * p.Ci + st_c) * 2;
    };
    auto stream_advance = [&]() {
        st_c += BK;
        if (st_c >= p.Ci) { st_c = 0; ++st_t; }
        ++st_kt;
    };
    int stg = 0;                                          // ring slot of the K tile being computed
    {   // prologue: K tiles 0 and 1 of the first tile (KT >= 4 is guaranteed by the host)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const char *abase, *bbase;
            stream_bases(abase, bbase);
            stage_pieces(str, abase, bbase, ring + s * S_ELEMS, std::integral_constant<int, 0>{}, std::integral_constant<int, NPIECE>{});
            stream_advance();
        }
        P8_WAIT_VM(NPIECE);
        P8_BARRIER();
    }

    f32x16 acc[2][NJ];
    for (;;) {
        // ---- the following item (its meta data is requested during K tile 0, its pointers are built at K tile 2) ----------
        int r2 = r, m2 = 0, n2 = 0;
        const bool has_next = p8_next_item(tiles_m, gy, items, w, G, r2, m2, n2, rev, BNT);
        if (!has_next) dummy_ctx(nxt);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

        if ((stagger & 1) && wn == 1) P8_BARRIER();             // stagger: group 1 runs one barrier behind group 0
        const int KT = cur.KT;
#ifdef SVSR_P8_STAMP
        const long long t_k0 = __builtin_amdgcn_s_memtime();
#endif
        for (int kt = 0; kt < KT; ++kt) {
            if (kt == 2 && has_next) make_ctx(nxt, m2, n2, par ^ 1);
            if (!st_next && st_kt == KT) { st_next = true; st_kt = 0; st_t = 0; st_c = 0; str = nxt; }
            const char *abase, *bbase;
            stream_bases(abase, bbase);
            const bf16_t* cA = ring + stg * S_ELEMS;
            const bf16_t* cB = cA + A_ELEMS;
            const int tgt = stg == 0 ? 2 : stg - 1;       // == (stg + 2) % 3
            bf16_t* dst = ring + tgt * S_ELEMS;
            constexpr int KF = 4 / PH, NP = NPIECE / PH;   // 16-deep fragments and DMA pieces per phase
            bf16x8 fa[KF][2], fb[KF][NJ];
#pragma unroll
            for (int h = 0; h < PH; ++h) {
                // -- load section: NP DMA pieces of K tile kt + 2, then the fragments of this phase ---------------------------------
                // (K tile 0 of a tile: the queue still holds the previous epilogue's stores and K tiles 0 / 1, requested two K tiles
                // ago: drained here, before anything new is issued — from then on the queue holds DMA pieces only)
                if (kt == 0 && h == 0) P8_WAIT_VM(0);
                if (h == 0) stage_pieces(str, abase, bbase, dst, std::integral_constant<int, 0>{}, std::integral_constant<int, NP>{});
                else stage_pieces(str, abase, bbase, dst, std::integral_constant<int, NP>{}, std::integral_constant<int, NPIECE>{});
#pragma unroll
                for (int kf = 0; kf < KF; ++kf) {
                    const int ch = (KF * h + kf) * 2 + (lane >> 5);
#pragma unroll
                    for (int i = 0; i < 2; ++i) fa[kf][i] = *reinterpret_cast<const bf16x8*>(cA + LDS_SWZ(wm * 64 + i * 32 + (lane & 31), ch));
#pragma unroll
                    for (int j = 0; j < NJ; ++j) fb[kf][j] = *reinterpret_cast<const bf16x8*>(cB + LDS_SWZ(wn * 32 * NJ + j * 32 + (lane & 31), ch));
                }
                if (h == PH - 1) {
                    // K tile kt + 1 has landed once only K tile kt + 2's six pieces (and, at kt = 0, the two meta pieces) are outstanding
                    if (kt == 0) { if (has_next) meta_dma(m2, par ^ 1); }
                    else P8_WAIT_VM(NPIECE);
                }
                P8_WAIT_LGKM0();
                P8_BARRIER();
                // -- matrix section ---------------------------------------------------------------------------------------------
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kf = 0; kf < KF; ++kf)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kf][i], fb[kf][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                P8_BARRIER();
            }
            stream_advance();
            stg = stg == 2 ? 0 : stg + 1;
        }
#ifdef SVSR_P8_STAMP
        const long long t_e0 = __builtin_amdgcn_s_memtime();
        if (tid == 0) { g_p8_stamp[4 * w] += t_e0 - t_k0; g_p8_stamp[4 * w + 1] += KT; }
#endif
        if ((stagger & 1) && wn == 0) P8_BARRIER();             // the groups meet again

        // ---- epilogue of the tile: the ring slot read last is idle until K tile 2 of the next tile is staged ----------------------
        // WAVE-PRIVATE: every wave turns its own four 32x32 accumulator fragments round through a 4 KiB patch of that slot (fp32
        // [32 rows][32 columns]: written in the MFMA layout, one column per lane, read back as rows of four channels per lane) and
        // stores 8 bytes per lane, 64 contiguous bytes per row.  No barrier and no idle wave until the reductions at the end.
        {
            const int idle = stg == 0 ? 2 : stg - 1;      // slot of the last K tile
            float* sW = reinterpret_cast<float*>(ring + idle * S_ELEMS) + wave * 1024;
            const int* rowtab = sRowTab + par * 512;
            const int n0t = cur.n0;
            const int c4 = lane & 7, rq = lane >> 3;      // this lane's four channels of a fragment, its row (+ 8 k) of a fragment
            float st_s[NJ], st_q[NJ];
            if (EPI == 0 && p.stats != nullptr) {
                // rows past the end of the class carry the first row's data, not zeros: a partial tile (wave-uniform test) masks them
                const int valid = __builtin_amdgcn_readlane(cur.desc, 1);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    float s = 0.f, q = 0.f;
                    if (valid >= BM) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int e = 0; e < 16; ++e) { const float v = acc[i][j][e]; s += v; q += v * v; }
                    } else {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int row = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                                const float v = row < valid ? acc[i][j][e] : 0.f;
                                s += v; q += v * v;
                            }
                    }
                    st_s[j] = s + __shfl_xor(s, 32, 64);
                    st_q[j] = q + __shfl_xor(q, 32, 64);
                }
            }
            const bool from_x = p.bnb_y == nullptr, swish_act = p.bnb_act == 2, has_add = p.addend != nullptr;
            float bs1[NJ][4], bs2[NJ][4];          // (WIDE: the same 8 accumulators, [c >> 2][c & 3] over the lane's 8 channels c)
            if (EPI == 1) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { bs1[j][k] = 0.f; bs2[j][k] = 0.f; }
            }
            if constexpr (WIDE) {
                static_assert(NJ == 2, "the full-line epilogue covers the two 32-column fragments of a wave together");
                const int c8 = lane & 7;                   // this lane's 8 channels: columns c8 * 8 .. + 7 of the wave's 64
                const int n = n0t + wn * 64 + c8 * 8;
                // target offsets of this lane's rows: row block i, 16-row half h, read kk -> row wm*64 + i*32 + 16 h + rq + 8 kk
                int offs[2][2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            const int dstpix = rowtab[(wm * 64 + i * 32 + 16 * h + rq + 8 * kk) * 2 + 1];
                            offs[i][h][kk] = dstpix >= 0 ? dstpix * p.out_pitch : -1;
                        }
                float mu[8], rs[8], sc[8], sh[8];
                if (EPI == 1) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) { mu[k] = p.bnb_mean[n + k]; rs[k] = p.bnb_rstd[n + k]; sc[k] = 0.f; sh[k] = 0.f; }
                    if (from_x || swish_act) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) { sc[k] = p.bnb_gamma[n + k] * rs[k]; sh[k] = __builtin_fmaf(-mu[k], sc[k], p.bnb_beta[n + k]); }
                    }
                }
                P8_WAIT_LGKM0();
#ifdef SVSR_P8_STAMP
                if (tid == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&g_p8_phase[0]), (unsigned long long)(__builtin_amdgcn_s_memtime() - t_e0));
                long long t_ph = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    // the global operands of this row block (4 rows x 16 bytes per tensor) are requested HERE, before the first is used (see the 8-byte form below)
                    u32x4 add_all[2][2], x_all[2][2], y_all[2][2];
                    if (has_add || EPI == 1) {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int kk = 0; kk < 2; ++kk) {
                                const long o = (long)((offs[i][h][kk] >= 0 ? offs[i][h][kk] : 0) + n);
                                if (has_add) add_all[h][kk] = *reinterpret_cast<const u32x4*>(p.addend + o);
                                if (EPI == 1) {
                                    x_all[h][kk] = *reinterpret_cast<const u32x4*>(p.bnb_x + o);
                                    if (!from_x) y_all[h][kk] = *reinterpret_cast<const u32x4*>(p.bnb_y + o);
                                }
                            }
                    }
                    if (has_add) {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int kk = 0; kk < 2; ++kk) asm volatile("" : "+v"(add_all[h][kk].x), "+v"(add_all[h][kk].y), "+v"(add_all[h][kk].z), "+v"(add_all[h][kk].w));
                    }
                    if (EPI == 1) {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int kk = 0; kk < 2; ++kk) {
                                asm volatile("" : "+v"(x_all[h][kk].x), "+v"(x_all[h][kk].y), "+v"(x_all[h][kk].z), "+v"(x_all[h][kk].w));
                                if (!from_x) asm volatile("" : "+v"(y_all[h][kk].x), "+v"(y_all[h][kk].y), "+v"(y_all[h][kk].z), "+v"(y_all[h][kk].w));
                            }
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        // rows 16 h .. 16 h + 15 of this row block: accumulator registers 8 h .. 8 h + 7 of both fragments -> patch [16][64]
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int e = 0; e < 8; ++e) sW[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * 64 + j * 32 + (lane & 31)] = acc[i][j][8 * h + e];
                        P8_WAIT_LGKM0();
                        // a lane's 8 channels are two 16-byte reads; 16 lanes of a ds_read_b128 group cover 4 rows x 4 lanes, and with every lane
                        // reading its LOW half first two of those rows share their banks (2-way).  Rows with bit 1 set read their HIGH half first:
                        // the 16 accesses of a group then fall on 16 different 16-byte slots in both instructions.
                        const int hb = (rq >> 1) & 1;
                        f32x4 rowv[2][2];
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            const f32x4 r0 = *reinterpret_cast<const f32x4*>(sW + (rq + 8 * kk) * 64 + c8 * 8 + 4 * hb);
                            const f32x4 r1 = *reinterpret_cast<const f32x4*>(sW + (rq + 8 * kk) * 64 + c8 * 8 + 4 * (1 - hb));
                            rowv[kk][0] = hb ? r1 : r0;
                            rowv[kk][1] = hb ? r0 : r1;
                        }
                        P8_WAIT_LGKM0();
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            const bool live = offs[i][h][kk] >= 0;
                            float v[8] = {rowv[kk][0][0], rowv[kk][0][1], rowv[kk][0][2], rowv[kk][0][3], rowv[kk][1][0], rowv[kk][1][1], rowv[kk][1][2], rowv[kk][1][3]};
                            if (has_add) {
                                float a8[8];
                                unpack8(add_all[h][kk], a8);
#pragma unroll
                                for (int c = 0; c < 8; ++c) v[c] += a8[c];
                            }
                            if (EPI == 1) {
                                float xv[8], yv[8];
                                unpack8(x_all[h][kk], xv);
                                if (from_x) {
#pragma unroll
                                    for (int c = 0; c < 8; ++c) yv[c] = __builtin_fmaf(xv[c], sc[c], sh[c]);
                                } else unpack8(y_all[h][kk], yv);
                                if (swish_act) {
#pragma unroll
                                    for (int c = 0; c < 8; ++c) {
                                        const float z = __builtin_fmaf(xv[c], sc[c], sh[c]) + (from_x ? 0.f : yv[c]);
                                        v[c] = live ? bf2f(f2bf(bf2f(f2bf(v[c])) * swish_grad(z))) : 0.f;
                                        bs1[c >> 2][c & 3] += v[c];
                                        bs2[c >> 2][c & 3] += v[c] * (xv[c] - mu[c]) * rs[c];
                                    }
                                } else {
#pragma unroll
                                    for (int c = 0; c < 8; ++c) {
                                        v[c] = (live && yv[c] > 0.f) ? bf2f(f2bf(v[c])) : 0.f;
                                        bs1[c >> 2][c & 3] += v[c];
                                        bs2[c >> 2][c & 3] += v[c] * (xv[c] - mu[c]) * rs[c];
                                    }
                                }
                            }
                            if (live) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.out) + (long)(offs[i][h][kk] + n)) = pack8(v);
                        }
                    }
#ifdef SVSR_P8_STAMP
                    { const long long t_n = __builtin_amdgcn_s_memtime(); if (tid == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&g_p8_phase[1 + i]), (unsigned long long)(t_n - t_ph)); t_ph = t_n; }
#endif
                }
            } else {
            // target offsets of this lane's rows: fragment i, read k -> row wm*64 + i*32 + rq + 8k
            int offs[2][4];          // elements (the launcher bounds rows * out_pitch below 2^31)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int dstpix = rowtab[(wm * 64 + i * 32 + rq + 8 * k) * 2 + 1];
                    offs[i][k] = dstpix >= 0 ? dstpix * p.out_pitch : -1;
                }
            P8_WAIT_LGKM0();
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = n0t + wn * 32 * NJ + j * 32 + c4 * 4;      // first of this lane's four channels
                // the global operands of this 32-column half of the wave tile (addend, x, y: 8 bytes per row and fragment) are requested
                // HERE, before the first is used: with LDS-DMA pieces of the next tile in flight hipcc turns every wait for an ordinary
                // load into vmcnt(0), so per-fragment requests would cost one full round trip each
                uint2 add_all[2][4], x_all[2][4], y_all[2][4];
                if (has_add || EPI == 1) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const long o = (long)((offs[i][k] >= 0 ? offs[i][k] : 0) + n);
                            if (has_add) add_all[i][k] = *reinterpret_cast<const uint2*>(p.addend + o);
                            if (EPI == 1) {
                                x_all[i][k] = *reinterpret_cast<const uint2*>(p.bnb_x + o);
                                if (!from_x) y_all[i][k] = *reinterpret_cast<const uint2*>(p.bnb_y + o);
                            }
                        }
                }
                float mu[4], rs[4], sc[4], sh[4];
                if (EPI == 1) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { mu[k] = p.bnb_mean[n + k]; rs[k] = p.bnb_rstd[n + k]; sc[k] = 0.f; sh[k] = 0.f; }
                    if (from_x || swish_act) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { sc[k] = p.bnb_gamma[n + k] * rs[k]; sh[k] = __builtin_fmaf(-mu[k], sc[k], p.bnb_beta[n + k]); }
                    }
                }
                // ARRIVAL POINT of the operands: a first use in straight-line code.  hipcc tracks outstanding loads per control-flow path
                // and, with LDS-DMA in flight, waits with vmcnt(0); left to find the first use inside a row's divergent `if (live)` it
                // re-waits in EVERY row, and on gfx950 vmcnt(0) also waits for the previous row's store.
                if (has_add) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(add_all[i][k].x), "+v"(add_all[i][k].y));
                }
                if (EPI == 1) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            asm volatile("" : "+v"(x_all[i][k].x), "+v"(x_all[i][k].y));
                            if (!from_x) asm volatile("" : "+v"(y_all[i][k].x), "+v"(y_all[i][k].y));
                        }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint2 (&add4)[4] = add_all[i];
                    const uint2 (&x4)[4] = x_all[i];
                    const uint2 (&y4)[4] = y_all[i];
#pragma unroll
                    for (int e = 0; e < 16; ++e) sW[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[i][j][e];
                    P8_WAIT_LGKM0();
                    f32x4 rowv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) rowv[k] = *reinterpret_cast<const f32x4*>(sW + (rq + 8 * k) * 32 + c4 * 4);
                    P8_WAIT_LGKM0();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // (no early exit for a padding row: a divergent skip in front of the first use of the hoisted operands makes hipcc
                        // re-wait vmcnt(0) in every row, and on gfx950 that counter includes the previous row's STORE — one store round
                        // trip per row, 6 us per tile.  Padding rows compute on row 0's operands and are masked out of sums and stores.)
                        const bool live = offs[i][k] >= 0;
                        float v[4] = {rowv[k][0], rowv[k][1], rowv[k][2], rowv[k][3]};
                        {   // (selects, not a branch: a uniform branch in every row has the same effect on hipcc's waits as the early exit)
                            const unsigned ax = has_add ? add4[k].x : 0u, ay = has_add ? add4[k].y : 0u;
                            v[0] += __uint_as_float(ax << 16); v[1] += __uint_as_float(ax & 0xffff0000u);
                            v[2] += __uint_as_float(ay << 16); v[3] += __uint_as_float(ay & 0xffff0000u);
                        }
                        if (EPI == 1) {
                            const float xv[4] = {__uint_as_float(x4[k].x << 16), __uint_as_float(x4[k].x & 0xffff0000u),
                                                 __uint_as_float(x4[k].y << 16), __uint_as_float(x4[k].y & 0xffff0000u)};
                            float yv[4];
                            if (from_x) {
#pragma unroll
                                for (int c = 0; c < 4; ++c) yv[c] = __builtin_fmaf(xv[c], sc[c], sh[c]);
                            } else {
                                yv[0] = __uint_as_float(y4[k].x << 16); yv[1] = __uint_as_float(y4[k].x & 0xffff0000u);
                                yv[2] = __uint_as_float(y4[k].y << 16); yv[3] = __uint_as_float(y4[k].y & 0xffff0000u);
                            }
                            // sums over the values the apply pass reads back (rounded to bf16): mean(g) is then the mean of what it is subtracted from
                            if (swish_act) {
#pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    const float z = __builtin_fmaf(xv[c], sc[c], sh[c]) + (from_x ? 0.f : yv[c]);
                                    v[c] = live ? bf2f(f2bf(bf2f(f2bf(v[c])) * swish_grad(z))) : 0.f;
                                    bs1[j][c] += v[c];
                                    bs2[j][c] += v[c] * (xv[c] - mu[c]) * rs[c];
                                }
                            } else {
#pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    v[c] = (live && yv[c] > 0.f) ? bf2f(f2bf(v[c])) : 0.f;
                                    bs1[j][c] += v[c];
                                    bs2[j][c] += v[c] * (xv[c] - mu[c]) * rs[c];
                                }
                            }
                        }
                        uint2 o2;
                        o2.x = pack2bf(v[0], v[1]); o2.y = pack2bf(v[2], v[3]);
                        if (live) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (long)(offs[i][k] + n)) = o2;
                    }
                }
            }
            }
            if (EPI == 1) {
                // lanes with the same c4 (8 row lanes rq) in the waves wm = 0..3 of a column group share their channels: partial sums
                // through LDS, added in the fixed order (wm, rq).  red: [8 waves][64 lanes][16] floats = 32 KiB of the idle slot.
                P8_SYNC_LDS();                               // every wave is done with its patch
                float* red = reinterpret_cast<float*>(ring + idle * S_ELEMS);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { red[(wave * 64 + lane) * 16 + j * 4 + k] = bs1[j][k]; red[(wave * 64 + lane) * 16 + 8 + j * 4 + k] = bs2[j][k]; }
                P8_SYNC_LDS();
                if (tid < 2 * BNT) {
                    const int which = tid / BNT, cc = tid - which * BNT;     // cc = wn*32*NJ + j*32 + c4*4 + k   (WIDE: wn*64 + c8*8 + 4 j + k, the lane's channel c = 4 j + k)
                    const int g = cc / (32 * NJ);
                    const int j = WIDE ? (cc >> 2) & 1 : (cc >> 5) % NJ, c4r = WIDE ? (cc >> 3) & 7 : (cc >> 2) & 7, k = cc & 3;
                    float s = 0.f;
                    for (int m = 0; m < 4; ++m)
                        for (int q = 0; q < 8; ++q) s += red[((g * 4 + m) * 64 + q * 8 + c4r) * 16 + which * 8 + j * 4 + k];
                    p.stats[((long)cur.m_tile * 2 + which) * p.Co + n0t + cc] = s;
                }
            } else if (p.stats != nullptr) {
                // sRed: [8 waves][64 columns][2]; waves (wm, wn), wm = 0..3, own the columns of wn: added in the order of wm
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (lane < 32) {
                        sRed[(wave * 64 + j * 32 + lane) * 2 + 0] = st_s[j];
                        sRed[(wave * 64 + j * 32 + lane) * 2 + 1] = st_q[j];
                    }
                P8_SYNC_LDS();
                if (tid < BNT) {
                    const int g = tid / (32 * NJ), cc = tid % (32 * NJ);
                    float s = 0.f, q = 0.f;
#pragma unroll
                    for (int m = 0; m < 4; ++m) { s += sRed[((g * 4 + m) * 64 + cc) * 2 + 0]; q += sRed[((g * 4 + m) * 64 + cc) * 2 + 1]; }
                    p.stats[((long)cur.m_tile * 2 + 0) * p.Co + n0t + tid] = s;
                    p.stats[((long)cur.m_tile * 2 + 1) * p.Co + n0t + tid] = q;
                }
            }
            // The epilogue's global stores stay in flight into the next tile: its first phase drains the VM queue (vmcnt(0)) BEFORE it
            // issues new DMA, so a counted wait only ever sees DMA pieces in the queue.  The LDS scratch (this ring slot, sRed) is free
            // again once everybody passed this barrier.
            P8_SYNC_LDS();
        }
#ifdef SVSR_P8_STAMP
        if (tid == 0) { g_p8_stamp[4 * w + 2] += __builtin_amdgcn_s_memtime() - t_e0; g_p8_stamp[4 * w + 3] += 1; }
#endif
        if (!has_next) return;
        cur = nxt;
        str = nxt;
        r = r2; par ^= 1;
        st_next = false;            // the stream is at K tile 2 of what is now the current tile
    }
}

#ifdef SVSR_P8_STAMP
/* build variant p8stamp (python -m syncvsr_amd.build --variant p8stamp; scripts/probes/p8_stamps.py): sums over the workgroups of {cycles inside
 * the K loops, K tiles, cycles inside the epilogues, tiles} since the last call; resets the counters */
extern "C" int svsr_debug_p8_stamps(long long* out4) {        /* out4: 8 words */
    static long long h[2048];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_p8_stamp), sizeof h) != hipSuccess) return SVSR_ERR_LAUNCH;
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    for (int i = 0; i < 512; ++i) for (int k = 0; k < 4; ++k) out4[k] += h[4 * i + k];
    long long ph[8];
    if (hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_p8_phase), sizeof ph) != hipSuccess) return SVSR_ERR_LAUNCH;
    for (int k = 0; k < 4; ++k) out4[4 + k] = ph[k];
    memset(ph, 0, sizeof ph);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_p8_phase), ph, sizeof ph);
    memset(h, 0, sizeof h);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_p8_stamp), h, sizeof h);
}
#endif

// ---------------------------------------------------------------------------------------------------------------------------------
int igemm_p8_launch(const IgemmFwdArgs& a, const int* meta, hipStream_t stream) {
    // meta: {256, 128, 3, M tiles, gy, classes, max taps, rows}; the epilogues this kernel has: (+addend | BatchNorm-backward), BatchNorm partials
    if (a.act == 1 || a.out_f32 || a.out_pre != nullptr) return SVSR_ERR_ARG;
    if (a.bias != nullptr || a.act == 2 || a.alpha != 1.f || a.drop.seed != nullptr) return SVSR_ERR_ARG;      // (the dense layers' epilogue lives in k_igemm_fwd_glds)
    const int bn = meta[1];                     // 128, or 64 (plans whose 128-wide items would not fill the chip)
    if ((bn != BN && bn != 64) || a.Co % bn != 0 || a.out_pitch % 8 != 0 || a.Ci % BK != 0) return SVSR_ERR_ARG;
    // per-lane addresses are 32-bit byte offsets from the tensor bases
    if ((long)a.Nimg * a.in_pix * a.in_pitch * 2 >= (1L << 32) || (long)a.Co * a.wt_taps * a.Ci * 2 >= (1L << 32)) return SVSR_ERR_ARG;
    if ((long)a.Nimg * a.out_pix * a.out_pitch >= (1L << 31)) return SVSR_ERR_ARG;          // the epilogue's row offsets are 32-bit element counts
    const int cus = svsr_stream_cus(stream);          // (a CU-masked stream: one workgroup per compute unit it may use)
    const int tiles_m = meta[3], gy = a.Co / bn;
    const int items = (tiles_m + 7) / 8 * 8 * gy;
    int G = items < cus ? items : cus;
    const int forced = svsr_tune_get(SVSR_TUNE_P8_GRID);
    if (forced > 0) G = forced < items ? forced : items;          // (above the CU count: one tile per workgroup, handed out by the dispatcher)
    const int ph = svsr_tune_get(SVSR_TUNE_P8_PH) == 2 ? 2 : 1, stagger = svsr_tune_get(SVSR_TUNE_P8_STAGGER) & 3;      // bit 0: wave groups one barrier apart, bit 1: odd workgroups walk their rounds backwards
    const bool wide = svsr_tune_get(SVSR_TUNE_P8_WIDE) != 0;
#define P8_LAUNCH(...) do { static bool set_ = false; \
        if (!set_) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_igemm_p8<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); set_ = true; } \
        hipLaunchKernelGGL((k_igemm_p8<__VA_ARGS__>), dim3(G), dim3(512), LDS_BYTES, stream, a, stagger); } while (0)
    if (bn == 64) { if (a.bnb_x != nullptr) P8_LAUNCH(1, 1, 1); else P8_LAUNCH(0, 1, 1); }
    // (the BatchNorm-backward epilogue in this form — p8_wide = 2 — shortens its epilogue by 19 % and lengthens its K loop by 7-12 %: 32 more
    // live registers of per-channel constants; layer2 62.3 -> 60.4 us, layer3 52.6 -> 53.9: the plain epilogue only by default)
    else if (wide && (a.bnb_x == nullptr || svsr_tune_get(SVSR_TUNE_P8_WIDE) == 2)) {
        if (a.bnb_x != nullptr) { if (ph == 2) P8_LAUNCH(1, 2, 2, true); else P8_LAUNCH(1, 1, 2, true); }
        else { if (ph == 2) P8_LAUNCH(0, 2, 2, true); else P8_LAUNCH(0, 1, 2, true); }
    }
    else if (a.bnb_x != nullptr) { if (ph == 2) P8_LAUNCH(1, 2); else P8_LAUNCH(1, 1); }
    else { if (ph == 2) P8_LAUNCH(0, 2); else P8_LAUNCH(0, 1); }
#undef P8_LAUNCH
    return svsr_check_launch();
}
