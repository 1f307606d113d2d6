// What a cross-stream hand-over costs the SIGNALLING stream (gfx950, ROCm 7.2): a chain of dependent kernels on the main stream with a
// side-stream launch handed over after every kernel, by (1) hipEventRecord + hipStreamWaitEvent with different event flags,
// (2) hipStreamWriteValue32 + hipStreamWaitValue32, (3) the NEXT main-stream kernel writing a word at its start + hipStreamWaitValue32.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/handover_probe.hip -o /tmp/handover_probe && /tmp/handover_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_stream(float* p, long n, unsigned* flag, unsigned value) {
    if (flag != nullptr && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x, s = (long)gridDim.x * blockDim.x;
    for (; i < n; i += s) p[i] = p[i] * 1.0001f + 1.0f;
}

int main() {
    const long nA = 32l << 20, nS = 8l << 20;       // main kernels ~128 MB r+w (~45 us), side ~32 MB: the host stays ahead of the device
    float *a, *s;
    unsigned* flag;
    CK(hipMalloc(&a, nA * 4)); CK(hipMalloc(&s, nS * 4)); CK(hipMalloc(&flag, 256));
    CK(hipMemset(a, 0, nA * 4)); CK(hipMemset(s, 0, nS * 4)); CK(hipMemset(flag, 0, 256));
    hipStream_t mainS, side;
    CK(hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    const int N = 200;
    struct Mode { const char* name; int kind; unsigned flags; };
    std::vector<Mode> modes = {
        {"no hand-over, no side launches", 0, 0},
        {"side launches without dependency", 5, 0},
        {"event: DisableTiming", 1, hipEventDisableTiming},
        {"event: DisableTiming|ReleaseToDevice", 1, hipEventDisableTiming | hipEventReleaseToDevice},
        {"event: DisableTiming|DisableSystemFence", 1, hipEventDisableTiming | hipEventDisableSystemFence},
        {"event: default (timing)", 1, hipEventDefault},
        {"WriteValue32 on main + WaitValue32 on side", 2, 0},
        {"next main kernel writes word + WaitValue32 on side", 3, 0},
        {"hand-over every 4th kernel: event DisableTiming", 4, hipEventDisableTiming},
    };
    unsigned counter = 0;
    for (const Mode& m : modes) {
        std::vector<hipEvent_t> evs(N);
        if (m.kind == 1 || m.kind == 4) for (auto& e : evs) CK(hipEventCreateWithFlags(&e, m.flags));
        hipEvent_t t0, t1;
        CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            auto h0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(t0, mainS));
            for (int i = 0; i < N; ++i) {
                unsigned* f = nullptr; unsigned v = 0;
                if (m.kind == 3 && i > 0) { f = flag; v = counter; }
                hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, mainS, a, nA, f, v);
                if (m.kind == 1 || (m.kind == 4 && i % 4 == 3)) { CK(hipEventRecord(evs[i], mainS)); CK(hipStreamWaitEvent(side, evs[i], 0)); }
                if (m.kind == 2) { ++counter; CK(hipStreamWriteValue32(mainS, flag, counter, 0)); CK(hipStreamWaitValue32(side, flag, counter, hipStreamWaitValueGte, 0xffffffffu)); }
                if (m.kind == 3) { ++counter; CK(hipStreamWaitValue32(side, flag, counter, hipStreamWaitValueGte, 0xffffffffu)); }
                if (m.kind != 0 && (m.kind != 4 || i % 4 == 3)) hipLaunchKernelGGL(k_stream, dim3(128), dim3(256), 0, side, s, nS, (unsigned*)nullptr, 0u);
            }
            if (m.kind == 3) hipLaunchKernelGGL(k_stream, dim3(1), dim3(64), 0, mainS, a, 64l, flag, counter);    // releases the last waiter
            CK(hipEventRecord(t1, mainS));
            auto h1 = std::chrono::steady_clock::now();
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, t0, t1));
            if (rep == 2) printf("%-58s main chain %8.2f us per kernel   host enqueue %6.2f us per iteration\n", m.name, ms * 1e3 / N,
                                 std::chrono::duration<double, std::micro>(h1 - h0).count() / N);
        }
        if (m.kind == 1 || m.kind == 4) for (auto& e : evs) CK(hipEventDestroy(e));
    }
    return 0;
}
