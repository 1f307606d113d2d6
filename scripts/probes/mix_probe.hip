// Steady-state K loop of a 128 x 128 x 64 MFMA tile (4 waves, 64 x 64 wave tiles, two workgroups per CU) with the operands coming from an
// L2 / MALL-resident matrix, in two forms:
//   MODE 0  both operands through LDS (LDS-DMA, 2-deep ring, vmcnt(0) + one barrier per K step) — the structure of k_igemm_fwd_glds<128,128,2>
//   MODE 1  A through LDS (LDS-DMA, 3-deep ring), B (the weights) STRAIGHT FROM GLOBAL MEMORY INTO REGISTERS as MFMA fragments, one K step ahead
//           (inline-asm global_load_dwordx4 + hand-counted vmcnt), so the LDS carries half the bytes per MFMA
// Question: the LDS sustains ~115-120 B/clk per CU in these kernels and the 64 x 64 wave tile needs 1 KiB of fragment reads per MFMA plus
// the DMA writes of both tiles — is the K loop faster when the vector-memory path carries one operand?
// Measured (round 5, 4096^3 operands, 256 K steps, constant data): MODE 0 0.94 us per K step and workgroup at two workgroups per CU = 1,147 TFLOP/s
// (0.64 us / 840 TFLOP/s at one per CU); MODE 1 2.20 us = 488 TFLOP/s (1.25 us / 430 at one per CU): fragment-shaped global loads (16 bytes per lane,
// 32 rows per instruction) are 2.3x SLOWER than the LDS path — the answer is no.  It also says the 128 x 128 K loop itself runs at 46 % of the
// MFMA peak in steady state: a 2,560 x 3,072 x 768 dense layer (12 K steps = 11.3 us of the 22-24 us launch) is bound by its prologue, its 15.7 MB
// output burst and the launch, not by the loop.
//   hipcc --offload-arch=gfx950 -O3 -w scripts/probes/mix_probe.hip -o scripts/probes/mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short bf16_t;
#define SWZ(row, chunk) ((row) * 64 + ((((chunk) ^ (((row) >> 1) & 7))) << 3))

__device__ __forceinline__ void glds16(const void* src, void* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const bf16_t* A, const bf16_t* B, float* sink, int K, int ksteps, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sS = reinterpret_cast<bf16_t*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int bm = blockIdx.x / tiles_n, bn = blockIdx.x % tiles_n;
    const int slot = tid & 7, r0 = tid >> 3, csw = slot ^ ((r0 >> 1) & 7), wrow = wave * 8;
    const bf16_t* a_ptr = A + (long)(bm * 128 + r0) * K + csw * 8;        // rows r0 + 32 i
    const bf16_t* b_ptr = B + (long)(bn * 128 + r0) * K + csw * 8;
    const int kmask = K / 64 - 1;                                        // K/64 is a power of two: the loop walks K cyclically
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (MODE == 0) {
        constexpr int S = 256 * 64;                                      // [A 128 rows | B 128 rows] x 64
        auto stage = [&](int kt, int buf) {
            const int c0 = (kt & kmask) * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(a_ptr + (long)i * 32 * K + c0, sS + buf * S + (wrow + 32 * i) * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(b_ptr + (long)i * 32 * K + c0, sS + buf * S + 128 * 64 + (wrow + 32 * i) * 64);
        };
        stage(0, 0);
        for (int it = 0; it < ksteps; ++it) {
            const int buf = it & 1;
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (it + 1 < ksteps) stage(it + 1, buf ^ 1);
            const bf16_t* cA = sS + buf * S;
            const bf16_t* cB = cA + 128 * 64;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ch = ks * 2 + (lane >> 5);
                bf16x8 fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(cA + SWZ(wm0 + i * 32 + (lane & 31), ch));
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(cB + SWZ(wn0 + j * 32 + (lane & 31), ch));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
    } else {
        constexpr int S = 128 * 64, NS = 3;
        auto stageA = [&](int kt) {
            const int c0 = (kt & kmask) * 64;
            bf16_t* d = sS + (kt % NS) * S;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(a_ptr + (long)i * 32 * K + c0, d + (wrow + 32 * i) * 64);
        };
        // B fragments of one K step: [ks][j], lane l = column bn*128 + wn0 + j*32 + (l & 31), k = ks*16 + (l >> 5)*8 .. +7
        const unsigned voff0 = (unsigned)(((long)(wn0 + (lane & 31)) * K + (lane >> 5) * 8) * 2);
        const unsigned voff1 = voff0 + (unsigned)(32 * K * 2);
        const bf16_t* bbase = B + (long)bn * 128 * K;
        bf16x8 b0[4][2], b1[4][2];
#define BLOAD(dst, kt)                                                                                                         \
        {                                                                                                                      \
            const bf16_t* sb = bbase + ((kt) & kmask) * 64;                                                                    \
            asm volatile("global_load_dwordx4 %0, %8, %10\n\tglobal_load_dwordx4 %1, %9, %10\n\t"                             \
                         "global_load_dwordx4 %2, %8, %10 offset:32\n\tglobal_load_dwordx4 %3, %9, %10 offset:32\n\t"        \
                         "global_load_dwordx4 %4, %8, %10 offset:64\n\tglobal_load_dwordx4 %5, %9, %10 offset:64\n\t"        \
                         "global_load_dwordx4 %6, %8, %10 offset:96\n\tglobal_load_dwordx4 %7, %9, %10 offset:96"            \
                         : "=&v"(dst[0][0]), "=&v"(dst[0][1]), "=&v"(dst[1][0]), "=&v"(dst[1][1]), "=&v"(dst[2][0]), "=&v"(dst[2][1]),      \
                           "=&v"(dst[3][0]), "=&v"(dst[3][1])                                                                  \
                         : "v"(voff0), "v"(voff1), "s"(sb)                                                                     \
                         : "memory");                                                                                          \
        }
#define BWAIT(dst, n)                                                                                                          \
        asm volatile("s_waitcnt vmcnt(" #n ")\n\ts_barrier"                                                                   \
                     : "+v"(dst[0][0]), "+v"(dst[0][1]), "+v"(dst[1][0]), "+v"(dst[1][1]), "+v"(dst[2][0]), "+v"(dst[2][1]),  \
                       "+v"(dst[3][0]), "+v"(dst[3][1])                                                                        \
                     :: "memory");
#define COMPUTE(bb, it)                                                                                                        \
        {                                                                                                                      \
            const bf16_t* cA = sS + ((it) % NS) * S;                                                                           \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                 \
                const int ch = ks * 2 + (lane >> 5);                                                                           \
                bf16x8 fa[2];                                                                                                  \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(cA + SWZ(wm0 + i * 32 + (lane & 31), ch)); \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                  \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], bb[ks][j], acc[i][j], 0, 0, 0); \
            }                                                                                                                  \
        }
        // issue order per step: B(it + 1) then A(it + 2): at the next step's wait B(it + 1) and A(it + 1) are older than the four pieces of A(it + 2)
        stageA(0);
        BLOAD(b0, 0);
        stageA(1);
        for (int it = 0; it < ksteps; it += 2) {
            BWAIT(b0, 4);
            BLOAD(b1, it + 1);
            stageA(it + 2);
            COMPUTE(b0, it);
            BWAIT(b1, 4);
            BLOAD(b0, it + 2);
            stageA(it + 3);
            COMPUTE(b1, it + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" :: "v"(b0[0][0]), "v"(b1[0][0]));
    }
    float t = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    sink[blockIdx.x * 256 + tid] = t;
}

int main(int argc, char** argv) {
    const int M = 4096, N = 4096, K = 4096;
    const int ksteps = argc > 1 ? atoi(argv[1]) : 256;
    bf16_t *A, *B; float* sink;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&sink, 1024 * 256 * 4);
    hipMemset(A, 0x3c, (size_t)M * K * 2); hipMemset(B, 0x3c, (size_t)N * K * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int grid : {512, 256}) {
        for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 2; ++mode) {
            const size_t lds = mode == 0 ? 2 * 256 * 64 * 2 : 3 * 128 * 64 * 2;
            for (int w = 0; w < 2; ++w) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), lds, 0, A, B, sink, K, ksteps, 32);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), lds, 0, A, B, sink, K, ksteps, 32);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = 2.0 * 128 * 128 * 64 * (double)ksteps * grid;
            printf("grid %d mode %d (%s): %.1f us, %.0f TFLOP/s, %.2f us per K step and workgroup\n", grid, mode, mode == 0 ? "A, B via LDS" : "A via LDS, B direct",
                   ms * 1e3, fl / ms / 1e9, ms * 1e3 / ksteps);
        }
    }
    hipError_t e = hipGetLastError();
    printf("status %s\n", hipGetErrorString(e));
    return 0;
}
