// LDS fill rate per CU (gfx950): the same L2-resident tile stream brought into a 3 x 48 KiB LDS ring by (A) LDS-DMA
// (global_load_lds_dwordx4: what every contraction kernel of this library stages with) and (B) ordinary global_load_dwordx4 into
// registers + ds_write_b128.  512 threads per workgroup, one workgroup per CU, two tiles in flight, nothing consumes the tiles.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/fill_rate_probe.hip -o /tmp/fill_rate_probe && /tmp/fill_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int TILE = 48 * 1024, PIECES = TILE / (512 * 16);      // 6 pieces of 16 bytes per thread

template <int MODE, int WAVES>
__global__ __launch_bounds__(512) void k_fill(const char* __restrict__ src, long span, int tiles, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6;
    const long b0 = (long)blockIdx.x * 131072;
    const char* base = src;
    unsigned acc = 0;
    if (wave >= WAVES) return;                                      // only WAVES waves of the workgroup issue
    constexpr int NT = WAVES * 64, PP = TILE / (NT * 16);           // pieces per issuing thread and tile
    if (MODE == 0) {
        auto stage = [&](int t) {
            const char* s = base + (b0 + (long)t * TILE) % (span - TILE);
            unsigned char* d = smem + (t % 3) * TILE;
#pragma unroll
            for (int i = 0; i < PP; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + (i * NT + tid) * 16),
                                                 (__attribute__((address_space(3))) void*)(d + (i * NT + (tid & ~63)) * 16), 16, 0, 0);
        };
        stage(0); stage(1);
        for (int t = 0; t < tiles; ++t) {
            if (t + 2 < tiles) stage(t + 2);
            if (t + 2 < tiles) { if (PP == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else if (PP == 12) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        acc = smem[tid * 4];
    } else {
        u32x4 r[2][PP];
        auto load = [&](int t, int b) {
            const char* s = base + (b0 + (long)t * TILE) % (span - TILE);
#pragma unroll
            for (int i = 0; i < PP; ++i) r[b][i] = *reinterpret_cast<const u32x4*>(s + (i * NT + tid) * 16);
        };
        load(0, 0);
        for (int t = 0; t < tiles; t += 2) {
            load(t + 1, 1);
            unsigned char* d0 = smem + (t % 3) * TILE;
#pragma unroll
            for (int i = 0; i < PP; ++i) *reinterpret_cast<u32x4*>(d0 + (i * NT + tid) * 16) = r[0][i];
            load(t + 2, 0);
            unsigned char* d1 = smem + ((t + 1) % 3) * TILE;
#pragma unroll
            for (int i = 0; i < PP; ++i) *reinterpret_cast<u32x4*>(d1 + (i * NT + tid) * 16) = r[1][i];
        }
        acc = r[0][0].x + smem[tid * 4];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int WAVES>
int run(const char* name, const char* src, long span, unsigned* sink) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fill<MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * TILE));
    const int tiles = 2048;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k_fill<MODE, WAVES>), dim3(256), dim3(512), 3 * TILE, 0, src, span, tiles, sink);
        CK(hipEventRecord(b, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep == 1) printf("%-44s %d issuing waves: %7.1f GB/s per CU, %6.2f TB/s chip\n", name, WAVES, (double)tiles * TILE / (ms * 1e-3) / 1e9, 256.0 * tiles * TILE / (ms * 1e-3) / 1e12);
    }
    return 0;
}

int main(int argc, char** argv) {
    // source span: 64 MB (default) streams from the Infinity Cache / HBM; 2 MB (argument "2") stays in every XCD's L2
    const long span = (argc > 1 ? atol(argv[1]) : 64l) << 20;
    printf("source span %ld MB\n", span >> 20);
    char* src; unsigned* sink;
    CK(hipMalloc(&src, span)); CK(hipMalloc(&sink, 256)); CK(hipMemset(src, 1, span));
    if (run<0, 8>("LDS-DMA (global_load_lds_dwordx4)", src, span, sink)) return 1;
    if (run<0, 4>("LDS-DMA (global_load_lds_dwordx4)", src, span, sink)) return 1;
    if (run<0, 1>("LDS-DMA (global_load_lds_dwordx4)", src, span, sink)) return 1;
    if (run<1, 8>("global_load_dwordx4 + ds_write_b128", src, span, sink)) return 1;
    if (run<1, 4>("global_load_dwordx4 + ds_write_b128", src, span, sink)) return 1;
    return 0;
}
