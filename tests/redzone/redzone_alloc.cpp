// Test infrastructure ("race detection / sanitizers"): a pluggable device allocator for PyTorch that puts a poisoned
// 4 KiB red zone in front of and behind EVERY tensor, so that a kernel of libsyncvsr_hip.so that writes outside the buffers it was handed
// (a wrong swizzle, a 32-bit offset that wrapped, an unmasked padding row) is caught by the test that ran it.
//   tests/conftest.py installs it when SVSR_REDZONE=1 (torch.cuda.memory.CUDAPluggableAllocator) and calls rz_check_all() after every
//   test; tests/test_gpu_redzone.py runs the kernel-level test files that way.
// Every allocation is its own hipMalloc of size + 2 x 4 KiB.  The tail zone starts AT the first byte behind the tensor (no alignment
// slack: an off-by-one store is caught).  Freed blocks are kept until the next rz_check_all() (a kernel still in flight on a side stream
// may be writing them, and a late overflow must still be seen), then checked and released.
//   hipcc -shared -fPIC -O2 -o libredzone.so redzone_alloc.cpp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <sys/types.h>

#include <mutex>
#include <unordered_map>
#include <vector>

namespace {
constexpr size_t GUARD = 4096;
constexpr unsigned char POISON = 0xA5;
struct Block { char* base; size_t size; };
std::mutex g_mu;
std::unordered_map<void*, Block> g_live;       // user pointer -> block
std::vector<Block> g_freed;                    // waiting for the next sweep
long g_violations = 0, g_allocs = 0;
unsigned char g_host[2 * GUARD];

long check_block(const Block& b) {
    long bad = 0;
    if (hipMemcpy(g_host, b.base, GUARD, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (hipMemcpy(g_host + GUARD, b.base + GUARD + b.size, GUARD, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    long first = -1;
    for (size_t i = 0; i < 2 * GUARD; ++i)
        if (g_host[i] != POISON) { ++bad; if (first < 0) first = (long)i; }
    if (bad != 0)
        fprintf(stderr, "[redzone] %ld byte(s) of the red zones of a %zu-byte tensor were overwritten (first at %s%ld)\n", bad, b.size,
                first < (long)GUARD ? "head+" : "tail+", first < (long)GUARD ? first : first - (long)GUARD);
    return bad;
}
}  // namespace

extern "C" {

void* rz_malloc(ssize_t size, int device, hipStream_t stream) {
    (void)device;
    if (size < 0) return nullptr;
    char* base = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&base), (size_t)size + 2 * GUARD) != hipSuccess) return nullptr;
    // poison on the allocating stream: whatever is launched on it afterwards (the first possible writer of this tensor) comes later
    (void)hipMemsetAsync(base, POISON, GUARD, stream);
    (void)hipMemsetAsync(base + GUARD + size, POISON, GUARD, stream);
    (void)hipStreamSynchronize(stream);        // (other streams may touch the tensor without ordering behind this one: be done before returning)
    std::lock_guard<std::mutex> lock(g_mu);
    g_live[base + GUARD] = Block{base, (size_t)size};
    ++g_allocs;
    return base + GUARD;
}

void rz_free(void* ptr, ssize_t size, int device, hipStream_t stream) {
    (void)size; (void)device; (void)stream;
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_live.find(ptr);
    if (it == g_live.end()) return;
    g_freed.push_back(it->second);
    g_live.erase(it);
}

/* synchronises the device, checks the red zones of every live and every freed-since-the-last-sweep block, releases the freed ones;
 * returns the total number of violated BYTES seen so far in this process */
long rz_check_all(void) {
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lock(g_mu);
    for (auto& kv : g_live) {
        const long bad = check_block(kv.second);
        if (bad != 0) {      // count once, then re-arm so that the next test starts clean
            g_violations += bad;
            (void)hipMemset(kv.second.base, POISON, GUARD);
            (void)hipMemset(kv.second.base + GUARD + kv.second.size, POISON, GUARD);
        }
    }
    for (const Block& b : g_freed) { g_violations += check_block(b); (void)hipFree(b.base); }
    g_freed.clear();
    return g_violations;
}

long rz_alloc_count(void) { return g_allocs; }

}  // extern "C"
