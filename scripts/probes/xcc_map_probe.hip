// Which XCD does workgroup (x, y) of a 2-D grid run on?  (gfx950: s_getreg_b32 HW_REG_XCC_ID)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x + gridDim.x * blockIdx.y] = (int)(v & 0xf);
}
int main() {
    const int gx = 20, gy = 24;
    int* d; hipMalloc(&d, gx * gy * 4);
    hipLaunchKernelGGL(k, dim3(gx, gy), dim3(256), 0, 0, d);
    int h[gx * gy]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int ok = 0;
    for (int i = 0; i < gx * gy; ++i) ok += (h[i] == (i & 7));
    printf("grid %d x %d: workgroups whose XCC id == linear index %% 8: %d of %d\nfirst 40 ids:", gx, gy, ok, gx * gy);
    for (int i = 0; i < 40; ++i) printf(" %d", h[i]);
    printf("\n");
    return 0;
}
