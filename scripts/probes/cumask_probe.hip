// Which compute units does bit i of a hipExtStreamCreateWithCUMask mask stand for?  (gfx950)
// For a set of masks, a census launch (many long-lived 160-KiB-LDS workgroups: one per compute unit) records (XCC_ID, SE_ID, CU_ID) of every
// workgroup; the program prints the distinct compute units per mask and, per XCD, how many it saw.
//   hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip && ./cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>
__global__ void census(unsigned* out, int spin) {
    extern __shared__ char lds[];
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) {
        lds[0] = 1;
        out[blockIdx.x * 2] = xcc & 0xf;
        out[blockIdx.x * 2 + 1] = hw;
        for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);      // stay resident so that the next workgroup goes elsewhere
    }
    __syncthreads();
}
static void run(const char* name, const std::vector<int>& bits, int total) {
    uint32_t mask[8] = {0};
    for (int b : bits) mask[b / 32] |= 1u << (b % 32);
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask);
    if (e != hipSuccess) { printf("%s: create failed %d\n", name, (int)e); return; }
    const int G = 1024;
    unsigned* d; hipMalloc(&d, G * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(census), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(census, dim3(G), dim3(64), 160 * 1024, s, d, 2000);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(G * 2);
    hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
    std::set<unsigned> cus; int per_xcc[16] = {0};
    std::set<unsigned> per_xcc_set[16];
    for (int i = 0; i < G; ++i) {
        const unsigned xcc = h[2 * i], hw = h[2 * i + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        cus.insert(key); per_xcc_set[xcc & 15].insert(key);
    }
    printf("%s: %zu bits -> %zu distinct CUs; per XCD:", name, bits.size(), cus.size());
    for (int x = 0; x < 8; ++x) printf(" %zu", per_xcc_set[x].size());
    printf("\n");
    if (bits.size() <= 8) { for (unsigned k : cus) printf("   xcc %u se %u sh %u cu %u\n", k >> 12, (k >> 8) & 7, (k >> 4) & 1, k & 15); }
    hipFree(d); hipStreamDestroy(s);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int total = p.multiProcessorCount;
    printf("device CUs %d\n", total);
    std::vector<int> all; for (int i = 0; i < total; ++i) all.push_back(i);
    run("all", all, total);
    for (int b : {0, 1, 7, 8, 9, 32, 255}) { char n[32]; snprintf(n, sizeof n, "bit %d", b); run(n, {b}, total); }
    std::vector<int> low, top, x7, nx7;
    for (int i = 0; i < total; ++i) { (i < total - 32 ? low : top).push_back(i); (i % 8 == 7 ? x7 : nx7).push_back(i); }
    run("low 224 (spread main)", low, total);
    run("top 32 (spread side)", top, total);
    run("i%8==7 (xcd side)", x7, total);
    run("i%8!=7 (xcd main)", nx7, total);
    return 0;
}
