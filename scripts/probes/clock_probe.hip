// Effective shader clock under load (gfx950): clock64() counts shader cycles (s_memtime), wall_clock64() counts the constant
// 100 MHz reference (s_memrealtime).  Their ratio inside a kernel is the clock the CUs actually ran at.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/clock_probe.hip -o scripts/probes/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>   // 0: MFMA only, 1: VALU fma only, 2: idle spin (s_sleep)
__global__ __launch_bounds__(256) void k_probe(long* out, float* sink, int iters) {
    const long c0 = clock64(), w0 = wall_clock64();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (short)(threadIdx.x + k); b[k] = (short)(threadIdx.x * 3 + k); }
    float v = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v = v * 1.0001f + 0.5f;
        } else {
            __builtin_amdgcn_s_sleep(8);
        }
    }
    const long c1 = clock64(), w1 = wall_clock64();
    float s = v;
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = w1 - w0; }
}

template <int MODE>
static void run(const char* name, int blocks, int iters) {
    long* d; float* sink;
    hipMalloc(&d, blocks * 2 * sizeof(long)); hipMalloc(&sink, 4);
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, sink, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long> h(blocks * 2);
        hipMemcpy(h.data(), d, blocks * 2 * sizeof(long), hipMemcpyDeviceToHost);
        double cs = 0, ws = 0;
        for (int i = 0; i < blocks; ++i) { cs += h[2 * i]; ws += h[2 * i + 1]; }
        const double mhz = cs / ws * 100.0;
        double tf = 0;
        if (MODE == 0) tf = (double)blocks * 4 /*waves*/ * iters * 4 * 32768.0 / (ms * 1e-3) / 1e12;
        printf("%-10s blocks %5d iters %7d: %8.3f ms  shader clock %7.1f MHz (cycles/block %.0f)  %s%.0f%s\n", name, blocks, iters, ms, mhz,
               cs / blocks, MODE == 0 ? "MFMA " : "", tf, MODE == 0 ? " TFLOP/s" : "");
    }
    hipFree(d); hipFree(sink);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s  CUs %d  clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run<2>("idle", 256, 20000);
    run<1>("valu", 2048, 200000);
    run<0>("mfma 1w/simd", 256, 200000);       // 4 waves per CU: one per SIMD
    run<0>("mfma 2w/simd", 512, 200000);
    run<0>("mfma long", 2048, 400000);
    return 0;
}
