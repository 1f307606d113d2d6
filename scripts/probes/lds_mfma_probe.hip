// How fast can 4 / 8 waves per CU run the fragment-read + MFMA block of the 128x128x64 K step when NOTHING else happens (no global traffic,
// no barriers)?  64x64 wave tile: 16 ds_read_b128 (A and B fragments, swizzled 128-byte rows as in igemm_fwd.hip) + 16 MFMA 32x32x16 per step.
//   hipcc --offload-arch=gfx950 -O3 -w scripts/probes/lds_mfma_probe.hip -o scripts/probes/lds_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define SWZ(row, chunk) ((row) * 64 + ((((chunk) ^ (((row) >> 1) & 7))) << 3))

template <int MODE>   // 0: reads + MFMA (compiler schedule), 1: MFMA only (fragments loaded once), 2: reads only, 3: reads one sub-step ahead (two register sets)
__global__ __launch_bounds__(256) void k(float* sink, int iters, int lds_extra) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    for (int i = threadIdx.x; i < 2 * 256 * 64; i += 256) smem[i] = (unsigned short)(i * 7);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[2], fb[2];
    for (int i = 0; i < 2; ++i) { fa[i] = *reinterpret_cast<const bf16x8*>(smem + SWZ(wm0 + i * 32 + (lane & 31), lane >> 5)); fb[i] = fa[i]; }
    if (MODE == 3) {
        bf16x8 ga[2][2], gb[2][2];
        auto ld = [&](const unsigned short* cA, const unsigned short* cB, int ks, int set) {
            const int ch = ks * 2 + (lane >> 5);
#pragma unroll
            for (int i = 0; i < 2; ++i) ga[set][i] = *reinterpret_cast<const bf16x8*>(cA + SWZ(wm0 + i * 32 + (lane & 31), ch));
#pragma unroll
            for (int j = 0; j < 2; ++j) gb[set][j] = *reinterpret_cast<const bf16x8*>(cB + SWZ(wn0 + j * 32 + (lane & 31), ch));
        };
        ld(smem, smem + 128 * 64, 0, 0);
        for (int it = 0; it < iters; ++it) {
            const unsigned short* cA = smem + (it & 1) * 256 * 64;
            const unsigned short* cB = cA + 128 * 64;
            const unsigned short* nA = smem + ((it + 1) & 1) * 256 * 64;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) ld(cA, cB, ks + 1, (ks + 1) & 1); else ld(nA, nA + 128 * 64, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[ks & 1][i], gb[ks & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else
    for (int it = 0; it < iters; ++it) {
        const unsigned short* cA = smem + (it & 1) * 256 * 64;
        const unsigned short* cB = cA + 128 * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ch = ks * 2 + (lane >> 5);
            if (MODE != 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(cA + SWZ(wm0 + i * 32 + (lane & 31), ch));
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(cB + SWZ(wn0 + j * 32 + (lane & 31), ch));
            }
            if (MODE != 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            } else {
                asm volatile("" :: "v"(fa[0]), "v"(fa[1]), "v"(fb[0]), "v"(fb[1]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.f) sink[0] = s;
}

template <int MODE>
static void run(const char* name, int blocks_per_cu, int iters) {
    float* sink; hipMalloc(&sink, 4);
    const size_t lds = blocks_per_cu == 1 ? 120 * 1024 : 66 * 1024;          // forces 1 or 2 workgroups per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * blocks_per_cu;
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, sink, iters, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double steps = (double)blocks * iters;                          // 128x128x64 steps
        printf("%-22s %d wg/CU: %7.3f ms  %6.0f cycles per step per wg @2.1GHz   %7.1f TFLOP/s-equivalent\n", name, blocks_per_cu, ms,
               ms * 1e-3 * 2.1e9 / iters, steps * 2.0 * 128 * 128 * 64 / (ms * 1e-3) / 1e12);
    }
    hipFree(sink);
}

int main() {
    run<1>("MFMA only", 1, 20000); run<1>("MFMA only", 2, 20000);
    run<2>("LDS reads only", 1, 20000); run<2>("LDS reads only", 2, 20000);
    run<0>("reads + MFMA", 1, 20000); run<0>("reads + MFMA", 2, 20000);
    run<3>("reads 1 ahead + MFMA", 1, 20000); run<3>("reads 1 ahead + MFMA", 2, 20000);
    return 0;
}
