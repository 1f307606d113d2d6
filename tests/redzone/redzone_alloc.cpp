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
#include <stdlib.h>
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Second mode (SVSR_TAILFLUSH=1): out-of-bounds READS.  Red zones only see stores; a per-lane offset that reads 16 bytes past a tensor, or a
// padding row fetched from behind the last image, changes no byte anywhere.  Here every tensor is its own virtual-memory reservation whose
// LAST mapped byte is the tensor's last byte (rounded up to 16: the kernels' vector width) and whose next page is NOT mapped: the first
// 16-byte access behind a tensor raises a GPU memory-access fault, which aborts the process — the child pytest run dies in the test that
// did it (tests/test_gpu_redzone.py reports the test named last).  hipMemAddressReserve / hipMemCreate / hipMemMap, one allocation
// granule of slack in front of the tensor at most.  Freed tensors are NEVER unmapped: physical pages released with hipMemRelease and handed
// out again by the next hipMemCreate came back with stale cache lines on this stack (measured: a convolution right after a sweep that unmapped
// the previous test's tensors read garbage or zeros) — so the pass keeps everything it ever allocated (a cap of
// 160 GiB makes tf_malloc fail, i.e. torch raise, long before the 288 GB device is full).
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
struct TfBlock { char* va; size_t reserved, mapped; hipMemGenericAllocationHandle_t handle; };
std::unordered_map<void*, TfBlock> g_tf_live;
size_t g_tf_gran = 0, g_tf_bytes = 0;
constexpr size_t TF_CAP = (size_t)160 << 30;
long g_tf_allocs = 0, g_tf_fail = 0;

}  // namespace

extern "C" {

void* tf_malloc(ssize_t size, int device, hipStream_t stream) {
    (void)stream;
    if (size < 0) return nullptr;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_tf_gran == 0) {
        if (hipMemGetAllocationGranularity(&g_tf_gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || g_tf_gran == 0) {
            fprintf(stderr, "[tailflush] hipMemGetAllocationGranularity failed: the virtual-memory API is not available\n");
            ++g_tf_fail;
            return nullptr;
        }
    }
    static size_t align = 0;
    if (align == 0) {        // SVSR_TF_ALIGN: alignment of the tensor's start = granularity of what is caught (default 16: the kernels' vector width)
        const char* e = getenv("SVSR_TF_ALIGN");
        align = e != nullptr && atol(e) >= 16 ? (size_t)atol(e) : 16;
    }
    const size_t used = ((size_t)size + align - 1) / align * align;
    const size_t mapped = used == 0 ? g_tf_gran : (used + g_tf_gran - 1) / g_tf_gran * g_tf_gran;
    if (g_tf_bytes + mapped > TF_CAP) { fprintf(stderr, "[tailflush] more than 160 GiB allocated by this pass (nothing is ever unmapped): refusing\n"); ++g_tf_fail; return nullptr; }
    TfBlock b{};
    b.mapped = mapped;
    b.reserved = mapped + g_tf_gran;                 // the granule behind the tensor stays unmapped
    void* va = nullptr;
    if (hipMemAddressReserve(&va, b.reserved, 0, nullptr, 0) != hipSuccess) { ++g_tf_fail; return nullptr; }
    b.va = static_cast<char*>(va);
    if (hipMemCreate(&b.handle, mapped, &prop, 0) != hipSuccess) { (void)hipMemAddressFree(va, b.reserved); ++g_tf_fail; return nullptr; }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemMap(va, mapped, 0, b.handle, 0) != hipSuccess || hipMemSetAccess(va, mapped, &acc, 1) != hipSuccess) {
        (void)hipMemRelease(b.handle); (void)hipMemAddressFree(va, b.reserved); ++g_tf_fail; return nullptr;
    }
    char* user = b.va + (mapped - used);
    g_tf_live[user] = b;
    g_tf_bytes += mapped;
    ++g_tf_allocs;
    return user;
}

void tf_free(void* ptr, ssize_t size, int device, hipStream_t stream) {
    (void)size; (void)device; (void)stream;
    std::lock_guard<std::mutex> lock(g_mu);
    g_tf_live.erase(ptr);        // (stays mapped: see the header of this section)
}

/* number of allocations that FAILED so far (the virtual-memory API missing, or the cap reached: the run proves nothing then) */
long tf_sweep(void) {
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lock(g_mu);
    return g_tf_fail;
}

size_t tf_bytes(void) { return g_tf_bytes; }

long tf_alloc_count(void) { return g_tf_allocs; }
size_t tf_granule(void) { return g_tf_gran; }

}  // extern "C"
