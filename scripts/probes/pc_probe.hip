// Producer / consumer contraction core, priced on a plain GEMM before it is built into the library (round 6, review item 1):
//   C[M][N] (bf16) = A[M][K] x B[N][K]^T, both operands K-contiguous bf16, persistent 256 x 128 tiles, one 8-wave workgroup per CU.
//   waves 0-3  CONSUMERS: wave tile 128 x 64 (4 x 2 blocks of mfma_f32_32x32x16_bf16): 24 ds_read_b128 per 32 MFMAs = 0.75 KiB of
//              LDS fragment reads per MFMA (the 64 x 64 wave tile of k_igemm_p8 reads 1 KiB); they never issue a vector-memory load
//              inside the K loop.
//   waves 4-7  PRODUCERS: all LDS-DMA (global_load_lds_dwordx4) of the 3-deep ring of 64-deep K tiles (48 KiB each), 12 pieces per
//              wave and K tile, two K tiles in flight.
//   hand-over  through LDS flag words, no s_barrier in the K loop: pfull[p] = K tiles of producer p that have landed (written after
//              its counted s_waitcnt vmcnt), cfree[c] = K tiles consumer c has finished reading (written right behind its last
//              fragment read of the tile: the LDS executes a wave's operations in order).  The K-tile counter runs on across tiles,
//              so the ring is continuous over tile boundaries.
// Reports: in-loop cycles per K tile (s_memtime stamps of consumer wave 0), launch time, TFLOP/s; checks sampled outputs on the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/pc_probe.hip -o scripts/probes/pc_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int BM = 256, BN = 128;
constexpr int RING_BYTES = 144 * 1024;
constexpr int FLAG_OFF = RING_BYTES;                  // int pfull[4] | int cfree[4]
constexpr int LDS_BYTES = RING_BYTES + 64;

__device__ __forceinline__ void glds16(const void* src, void* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__device__ __forceinline__ unsigned hash32(unsigned h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
__host__ __device__ inline float bf2f(bf16_t u) { union { unsigned i; float f; } v; v.i = ((unsigned)u) << 16; return v.f; }
__host__ __device__ inline bf16_t f2bf(float f) { union { unsigned i; float f; } v; v.f = f; unsigned u = v.i; u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
__host__ __device__ inline unsigned hash32h(unsigned h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
__host__ __device__ inline bf16_t gen(unsigned seed, size_t idx) {       // uniform in [-1, 1), full sign range (guide rule 25: no zero fill)
    const unsigned h = hash32h((unsigned)idx * 2654435761u + seed);
    return f2bf((float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f);
}
__global__ void k_fill(bf16_t* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = gen(seed, i);
}

// LDS flag words are touched with explicit DS instructions (a volatile generic pointer makes hipcc emit FLAT accesses + vmcnt(0) waits)
typedef __attribute__((address_space(3))) const i32x4* lds_i32x4_p;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void flag_store(unsigned addr, int val) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(val) : "memory"); }
__device__ __forceinline__ int min4(const i32x4 v) {
    int m = v[0] < v[1] ? v[0] : v[1];
    const int m2 = v[2] < v[3] ? v[2] : v[3];
    return __builtin_amdgcn_readfirstlane(m < m2 ? m : m2);
}
// smallest of the four flag words at LDS byte address `addr` (one broadcast ds_read_b128), as a scalar; waits for the LDS queue
__device__ __forceinline__ int flags_min(unsigned addr) {
    i32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return min4(v);
}

struct Args {
    const bf16_t* A; const bf16_t* B; bf16_t* C;
    int M, N, K;
    long long* stamps;      // [grid][2]: loop cycles of consumer wave 0, K tiles walked
};

// MODE 0: the full kernel.  MODE 1: consumers skip the MFMAs (reads + hand-over only).  MODE 2: consumers skip the fragment reads (MFMA on stale
// registers + hand-over).  MODE 3: producers issue nothing (flags only: the hand-over skeleton and the consumers' block alone).
// BK: depth of a K tile (64: 3 ring slots of 48 KiB; 32: 6 slots of 24 KiB, 64-byte LDS rows).  D: K tiles a producer leaves in flight behind the
// one it just issued (D <= NS - 2, or producers and consumers can wait for each other).
template <int MODE, int BK, int D, int PRIO>
__global__ __launch_bounds__(512, 2) void k_pc(const Args p) {
    constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, S_ELEMS = A_ELEMS + B_ELEMS;
    constexpr int NS = RING_BYTES / (S_ELEMS * 2);
    constexpr int KF = BK / 16;                        // 16-deep MFMA steps per K tile
    constexpr int CPR = BK / 8;                        // 16-byte chunks per LDS row
    constexpr int PROWS = 64 / CPR;                    // rows per 1-KiB DMA piece
    constexpr int NPA = BM / PROWS / 4, NPB = BN / PROWS / 4;      // pieces per producer wave and K tile
    constexpr int NP = NPA + NPB;
    static_assert(D <= NS - 2 && KF % 2 == 0, "ring depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* ring = reinterpret_cast<bf16_t*>(smem);
    const unsigned pfull = lds_addr(smem + FLAG_OFF), cfree = pfull + 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x, w = blockIdx.x;
    const int tiles_m = p.M / BM, gy = p.N / BN, items = tiles_m * gy;
    const int KT = p.K / BK;
    if (tid < 8) flag_store(pfull + 4 * tid, 0);
    __syncthreads();
    auto swz = [](int row) { return BK == 64 ? (row >> 1) & 7 : (row >> 2) & 3; };

    if (wave >= 4) {
        // ------------------------------------------------------------ producers ------------------------------------------------------------
        const int pw = wave - 4;
        if (PRIO & 3) __builtin_amdgcn_s_setprio(PRIO & 3);          // producers ahead of the matrix waves at the issue port
        const int slot = lane % CPR, rr = lane / CPR;
        const int rbase = pw * PROWS + rr;                       // rows rbase + 4 PROWS i
        const int csw = slot ^ swz(rbase);                       // the LDS chunk `slot` of that row holds global chunk csw (swizzle on the source)
        int g = 0, ps = 0;
        for (int q = w; q < items; q += G) {
            const int m_tile = q / gy, n0 = (q - m_tile * gy) * BN;
            unsigned a_off[NPA], b_off[NPB];
#pragma unroll
            for (int i = 0; i < NPA; ++i) a_off[i] = (unsigned)(((long)(m_tile * BM + rbase + 4 * PROWS * i) * p.K + csw * 8) * 2);
#pragma unroll
            for (int i = 0; i < NPB; ++i) b_off[i] = (unsigned)(((long)(n0 + rbase + 4 * PROWS * i) * p.K + csw * 8) * 2);
            for (int kt = 0; kt < KT; ++kt, ++g) {
                const int s = ps;
                ps = ps == NS - 1 ? 0 : ps + 1;
                if (g >= NS) {                                   // the slot's previous K tile (g - NS) has been read by every consumer
                    while (flags_min(cfree) < g - NS + 1) __builtin_amdgcn_s_sleep(1);
                }
                bf16_t* dst = ring + s * S_ELEMS;
                const char* abase = reinterpret_cast<const char*>(p.A) + (long)kt * BK * 2;
                const char* bbase = reinterpret_cast<const char*>(p.B) + (long)kt * BK * 2;
                if (MODE != 3) {
#pragma unroll
                    for (int i = 0; i < NPA; ++i) glds16(abase + a_off[i], dst + (pw * PROWS + 4 * PROWS * i) * BK);
#pragma unroll
                    for (int i = 0; i < NPB; ++i) glds16(bbase + b_off[i], dst + A_ELEMS + (pw * PROWS + 4 * PROWS * i) * BK);
                    if (g >= D) { WAIT_VM(NP * D); flag_store(pfull + 4 * pw, g - D + 1); }          // K tiles <= g - D of this producer have landed
                } else if (g >= D) flag_store(pfull + 4 * pw, g - D + 1);
            }
        }
        WAIT_VM(0);
        flag_store(pfull + 4 * pw, g);
        return;
    }

    // ---------------------------------------------------------------- consumers ----------------------------------------------------------------
    const int wm = wave & 1, wn = wave >> 1;
    if (PRIO & 4) __builtin_amdgcn_s_setprio(1);                     // (the other way round: matrix waves first)
    const int hi = lane >> 5, l31 = lane & 31;
    const int sw = swz(l31);
    const int a_row = (wm * 128 + l31) * BK, b_row = A_ELEMS + (wn * 64 + l31) * BK;
    int coff[KF];
#pragma unroll
    for (int kf = 0; kf < KF; ++kf) coff[kf] = ((2 * kf + hi) ^ sw) << 3;
    bf16x8 fa0[4], fb0[2], fa1[4], fb1[2];
    f32x16 acc[4][2];
    auto loadf = [&](bf16x8 (&fa)[4], bf16x8 (&fb)[2], const bf16_t* sb, int co) {
        if (MODE == 2) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sb + a_row + i * 32 * BK + co);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(sb + b_row + j * 32 * BK + co);
    };
    auto mma = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[2]) {
        if (MODE == 1) { asm volatile("" ::"v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fb[0]), "v"(fb[1])); return; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    int g = 0, cs = 0;
    long long t_loop = 0, n_kt = 0;
    // first K tile of the first item
    if (w < items) {
        while (flags_min(pfull) < 1) __builtin_amdgcn_s_sleep(1);
        loadf(fa0, fb0, ring, coff[0]);
    }
    for (int q = w; q < items; q += G) {
        const int m_tile = q / gy, n0 = (q - m_tile * gy) * BN;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const bool more = q + G < items;
        const long long t0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0), stated where hipcc's own wait counting sees it: nothing (no LDS read, no s_memtime) is outstanding at the loop head
        for (int kt = 0; kt < KT; ++kt, ++g) {
            const bf16_t* sb = ring + cs * S_ELEMS;
            cs = cs == NS - 1 ? 0 : cs + 1;
            const bf16_t* nb = ring + cs * S_ELEMS;
            const bool ahead = kt + 1 < KT || more;
            i32x4 pv;
            // invariant: the fragments of (g, 0) are in set 0 (requested during the previous K tile, waited for at its end)
#pragma unroll
            for (int kf = 0; kf < KF; ++kf) {
                if (kf + 1 < KF) {
                    if (kf & 1) loadf(fa0, fb0, sb, coff[(kf + 1) % KF]); else loadf(fa1, fb1, sb, coff[(kf + 1) % KF]);
                    if (kf + 2 == KF) {
                        flag_store(cfree + 4 * wave, g + 1);              // behind this wave's last read of the slot (the LDS runs a wave's operations in order)
                        pv = *(lds_i32x4_p)(unsigned long)pfull;          // requested before the MFMA block, looked at behind it
                    }
                } else {
                    const int have = min4(pv);       // (looked at on every path: hipcc's wait counting then sees the same queue on both)
                    if (ahead && have < g + 2) { while (flags_min(pfull) < g + 2) __builtin_amdgcn_s_sleep(1); }
                    loadf(fa0, fb0, nb, coff[0]);     // (unconditional: behind the last K tile of the workgroup a harmless read of a stale slot)
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kf & 1) mma(fa1, fb1); else mma(fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);          // the read-ahead has had a whole MFMA block to land
        }
        if (wave == 0) { t_loop += __builtin_amdgcn_s_memtime() - t0; n_kt += KT * (BK / 32); }
        // epilogue (probe: 2-byte stores straight from the MFMA layout; the library kernel turns fragments round through LDS)
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = m_tile * BM + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                        const int col = n0 + wn * 64 + j * 32 + l31;
                        p.C[(long)row * p.N + col] = f2bf(acc[i][j][e]);
                    }
        } else {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) s += acc[i][j][e];
            if (s == 12345.678f) p.C[0] = 1;
        }
    }
    if (wave == 0 && lane == 0) { p.stamps[w * 2] = t_loop; p.stamps[w * 2 + 1] = n_kt; }
}

template <int MODE, int BK, int D, int PRIO>
static void run(const char* name, int M, int N, int K, bool check) {
    bf16_t *A, *B, *C;
    long long* stamps;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
    hipMalloc(&stamps, 256 * 2 * 8);
    hipMemset(stamps, 0, 256 * 2 * 8);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, A, (size_t)M * K, 1u);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, B, (size_t)N * K, 2u);
    hipMemset(C, 0, (size_t)M * N * 2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_pc<MODE, BK, D, PRIO>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    const int items = (M / BM) * (N / BN);
    const int G = items < 256 ? items : 256;
    Args a{A, B, C, M, N, K, stamps};
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_pc<MODE, BK, D, PRIO>), dim3(G), dim3(512), LDS_BYTES, 0, a);
        hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); exit(1); }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    std::vector<long long> st(512);
    hipMemcpy(st.data(), stamps, 512 * 8, hipMemcpyDeviceToHost);
    double cyc = 0, kts = 0, worst = 0;
    for (int i = 0; i < G; ++i) { cyc += (double)st[2 * i]; kts += (double)st[2 * i + 1]; if (st[2 * i + 1] > 0 && st[2 * i] / (double)st[2 * i + 1] > worst) worst = st[2 * i] / (double)st[2 * i + 1]; }
    const double flops = 2.0 * M * N * K;
    printf("%-10s BK%d D%d P%d %6d x %4d x %5d  items %4d  %8.2f us  %7.1f TFLOP/s   in-loop %6.0f cycles per 32-deep half K tile (worst wg %6.0f; 512 = matrix pipe)\n",
           name, BK, D, PRIO, M, N, K, items, best * 1e3, flops / (best * 1e-3) / 1e12, kts > 0 ? cyc / kts : 0.0, worst);
    if (check && MODE == 0) {
        std::vector<bf16_t> hc((size_t)M * N);
        hipMemcpy(hc.data(), C, (size_t)M * N * 2, hipMemcpyDeviceToHost);
        double maxerr = 0;
        int bad = 0;
        for (int s = 0; s < 4000; ++s) {
            const int r = (int)(hash32h(s * 7 + 1) % (unsigned)M), c = (int)(hash32h(s * 13 + 5) % (unsigned)N);
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)bf2f(gen(1u, (size_t)r * K + k)) * (double)bf2f(gen(2u, (size_t)c * K + k));
            const double got = bf2f(hc[(size_t)r * N + c]);
            const double err = fabs(got - ref);
            if (err > maxerr) maxerr = err;
            if (err > 0.02 * fabs(ref) + 0.05 * sqrt((double)K) * 0.02) ++bad;
        }
        printf("             check: 4000 sampled outputs, max |err| %.4f, %d outside the bf16 bound%s\n", maxerr, bad, bad ? "  <-- WRONG" : "");
    }
    hipFree(A); hipFree(B); hipFree(C); hipFree(stamps);
}

template <int BK, int D, int PRIO>
static void sweep(int M, int N, int K) {
    run<0, BK, D, PRIO>("full", M, N, K, true);
    run<2, BK, D, PRIO>("no reads", M, N, K, false);
}

int main(int argc, char** argv) {
    const int shapes[][3] = {{2560, 3072, 768}, {8192, 256, 2304}, {4096, 4096, 4096}};
    for (auto& s : shapes) {
        sweep<64, 1, 0>(s[0], s[1], s[2]);
        sweep<64, 1, 1>(s[0], s[1], s[2]);
        sweep<64, 1, 3>(s[0], s[1], s[2]);
        sweep<64, 1, 4>(s[0], s[1], s[2]);
    }
    return 0;
}
