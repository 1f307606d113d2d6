// Library-wide plumbing (gfx950): the tuning table, and the fixed-order reductions that make a training step
// reproducible.
//
// No kernel of this library accumulates floating-point values with atomics.  Every grid-wide sum (BatchNorm statistics,
// split-K weight gradients, bias / LayerNorm-parameter gradients, losses, the gradient norm) is produced in two steps:
// each workgroup writes its partial result to its OWN row of a caller-owned workspace with plain stores, and a second,
// tiny launch adds the rows in a fixed order (k_colsum below, or the specialised BatchNorm finalisers in norm_act.hip).
// Two identical steps therefore give bit-identical results, which torch's own GPU kernels do not promise for the
// reference path (DistributedDataParallel over cuDNN/MIOpen; reference LRW/video/src/train.py:23-40).
#include <string.h>

#include "common.h"

// ---------------------------------------------------------------------------------------------------------------------
// tuning table: result-preserving knobs only (tile shapes, split counts, variant switches); set by svsr_tune(), never read
// from the environment.
// ---------------------------------------------------------------------------------------------------------------------
static int g_tune[SVSR_TUNE_N] = {
    /* IGEMM_TILE   */ 0,      // 0 auto, 64 / 128: force the M tile of svsr_igemm_fwd
    /* IGEMM_M128   */ 8192,   // rows from which 128-row tiles are used
    /* WG_BLOCKS    */ 0,      // target workgroups of svsr_igemm_wgrad (0: built-in per tile size)
    /* W3_BLOCKS    */ 512,    // target workgroups of svsr_conv3x3_wgrad in its 4-wave form: one round of two per CU; the 8-wave form (W3_WAVES) launches half of it, one per CU (inside the step, 8 waves: 4.874-4.878 / 4.885-4.888 / 4.977-4.986 / 4.943-4.968 ms at 512 / 384 / 640 / 768 in round 5)
    /* LN_RPB       */ 4,      // rows per workgroup of svsr_add_ln_bwd (one per wave: 16 -> 4 measured 6.00 -> 5.97 ms per LRW step)
    /* STEM_LDS_FWD */ 0,      // LDS-tiled stem BN+act+pool forward (measured slower)
    /* STEM_LDS_BWD */ 2,      // stem BN+act+pool backward: 0 plain, 1 LDS-tiled passes, 2 LDS-tiled apply pass + gather-form reduce pass (fastest)
    /* IGEMM_LDS_PAD */ 0,     // extra dynamic LDS bytes per svsr_igemm_fwd workgroup (occupancy experiments: fewer co-resident blocks per CU)
    /* IGEMM_BN64_BELOW */ 300, // multi-tap convolutions with fewer 128x128 tiles than this use 128x64 tiles (three workgroups per CU)
    /* WG_SHORT_K */ 16,       // svsr_igemm_wgrad: contractions of at most this many 64-row chunks use 64-wide tiles and no K split when that fills half the chip
    /* IGEMM_KSPLIT */ 160,    // svsr_igemm_fwd: launches of at most this many 64x64 tiles with >= 12 K steps split K over two wave groups per workgroup (0: never)
    /* EPI_BATCHED */ 1,       // svsr_igemm_fwd epilogue: all rows' staged accumulators / addend pieces requested before the first is used
    /* STEM_WG_PIPE */ 1,      // svsr_stem_conv_wgrad: next tile's operands prefetched into registers during the MFMA block
    /* STEM_FWD_DMA */ 1,      // svsr_stem_conv_fwd: bf16 prep pass + LDS-DMA tile fills (0: direct fp32 -> LDS path)
    /* IGEMM_LIN_BN64 */ 2048, // svsr_igemm_fwd: linears that would get 64x64 tiles use 128x64 tiles from this many rows on (0: never; LRS 768-wide outputs at 2,400 rows: 29.6 -> 29.1 ms per step)
    /* P8 */ 1,                // stride-1 3x3 convolution plans with Co % 128 == 0 and enough tiles use the persistent 8-wave 256x128 kernel (igemm_p8.hip) — also the forward of the stride-2 3x3 convolutions (3: stride 1 only)
    /* P8_GRID */ 0,           // workgroups of that kernel (0: one per CU)
    /* P8_MIN_ITEMS */ 200,    // ... from this many 256x128 tiles on (fewer leave CUs idle for the whole launch)
    /* P8_PH */ 1,             // phases per K tile of the persistent kernel: 1 (16 MFMAs between barriers) or 2 (8)
    /* P8_STAGGER */ 3,        // bit 0: its two wave groups run their phases one barrier apart; bit 1: odd workgroups walk their rounds last to first (their first epilogue falls elsewhere than the even ones')
    /* WG_IMGMAJOR */ 1,       // svsr_igemm_wgrad plans with >= 64 images enumerate rows by (position, block of 64 images): wave-uniform DMA bases (0: row-major)
    /* P8_BN64 */ 1,           // 3x3 plans with too few 256 x 128 items for one per CU use 256 x 64 tiles of the persistent kernel (layer4); 0: the 4-wave kernel
    /* IGEMM_NS64 */ 0,        // ring depth of the 64x64 tiles of svsr_igemm_fwd: 0 auto (3 / 4), or 6 / 8
    /* WG_UNITS */ 1,          // svsr_igemm_wgrad plans of long contractions as balanced unit lists (format 2); 0: (K split, task) grids
    /* WG_UNIT_MAX */ 48,      // ... longest unit in 64-row chunks before the list takes a further round of workgroups
    /* WG_UNIT_MIN */ 8,       // ... shortest unit worth a slab tile of its own
    /* IGEMM_KSPLIT128 */ 12,  // svsr_igemm_fwd: dense layers on <= 288 tiles of 128 x 64 with at least this many 64-deep K steps split K over two wave groups per workgroup (0: never)
    /* WG_XCD */ 1,            // svsr_igemm_wgrad unit lists of multi-tap plans: units dealt to the eight XCDs by the stretch of the contraction they cover (0: plain long-first order); same results
    /* W3_WAVES */ 8,          // waves per workgroup of svsr_conv3x3_wgrad: 8 (one workgroup per CU, the two wave groups split the nine taps: half the slabs, 138 instead of 247 registers per wave) or 4 (two workgroups per CU, nine taps per wave).  Round 5, same box: layer1 launch + reduce 80.1 -> 71.6 us at 928 frames, LRW step 5.03-5.05 -> 4.97-4.99 ms, LRS 23.69 -> 23.58-23.61
    /* REDUCE_CUS */ 256,      // compute units assumed when the split of a reduction is planned (svsr_reduction_cus, common.h): fixed, so the bits of a run do not depend on the partition it runs on; 0: the device's count
    /* W3_DENSE */ 1,          // svsr_conv3x3_wgrad contracts over the REAL pixels (dense dY tile, X rows gathered through a position table) instead of walking the zero-padded grid: 1.40x / 1.19x fewer MFMAs at 11 x 11 / 22 x 22 maps; 0: the padded walk of rounds 2-5
    /* P8_WIDE */ 2,           // epilogue of the persistent 8-wave kernel (256 x 128 tiles): full 128-byte lines per row (16-byte accesses, 8 lanes per row: both fragments of a wave through one [16][64] patch) — 1: the plain epilogue, 2: the BatchNorm-backward epilogue too; 0 = 64 bytes per row (rounds 3-5).  Same outputs bit for bit; stamped epilogue 8,785 -> 7,018 cycles per tile (plain), 28,473 -> 23,176 (BatchNorm backward, whose K loop pays 7 % for 32 more live registers); same-box steps: LRW 4.991 / 4.959 / 4.960 ms, LRS 23.14 / 23.20 / 23.03 at 0 / 1 / 2
};
static const char* const g_tune_names[SVSR_TUNE_N] = {"igemm_tile", "igemm_m128", "wg_blocks", "w3_blocks", "ln_rpb", "stem_lds_fwd", "stem_lds_bwd", "igemm_lds_pad", "igemm_bn64_below", "wg_short_k", "igemm_ksplit", "epi_batched", "stem_wg_pipe", "stem_fwd_dma", "igemm_lin_bn64", "p8", "p8_grid", "p8_min_items", "p8_ph", "p8_stagger", "wg_imgmajor", "p8_bn64", "igemm_ns64", "wg_units", "wg_unit_max", "wg_unit_min", "igemm_ksplit128", "wg_xcd", "w3_waves", "reduce_cus", "w3_dense", "p8_wide"};

int svsr_tune_get(int id) { return (id >= 0 && id < SVSR_TUNE_N) ? g_tune[id] : 0; }

extern "C" int svsr_tune(const char* key, int value) {
    if (key == nullptr) return SVSR_ERR_ARG;
    for (int i = 0; i < SVSR_TUNE_N; ++i)
        if (strcmp(key, g_tune_names[i]) == 0) { g_tune[i] = value; return SVSR_OK; }
    return SVSR_ERR_ARG;
}

extern "C" int svsr_tune_value(const char* key, int* value) {
    if (key == nullptr || value == nullptr) return SVSR_ERR_ARG;
    for (int i = 0; i < SVSR_TUNE_N; ++i)
        if (strcmp(key, g_tune_names[i]) == 0) { *value = g_tune[i]; return SVSR_OK; }
    return SVSR_ERR_ARG;
}

// ---------------------------------------------------------------------------------------------------------------------
// Compute units of the device: every persistent kernel sizes its grid / static tile list / cluster count by it.  (Round 5 also had
// CU-masked streams here — main and side stream on disjoint compute units; slower at every split, DESIGN.md appendix — removed in round 6.)
// ---------------------------------------------------------------------------------------------------------------------
namespace {
int device_cus() {
    static int per_dev[64];
    int d = 0;
    (void)hipGetDevice(&d);
    if (d < 0 || d >= 64) d = 0;
    if (per_dev[d] == 0) {
        int n = 0;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d);
        per_dev[d] = n > 0 ? n : 256;
    }
    return per_dev[d];
}
}  // namespace

int svsr_stream_cus(hipStream_t) { return device_cus(); }
int svsr_reduction_cus() { const int v = svsr_tune_get(SVSR_TUNE_REDUCE_CUS); return v > 0 ? v : device_cus(); }

extern "C" int svsr_device_cus(void) { return device_cus(); }

// ---------------------------------------------------------------------------------------------------------------------
// Test aid: a FOREIGN RESIDENT KERNEL — `workgroups` workgroups of 256 threads that each hold `lds_bytes` of LDS and sleep-poll a word of
// pinned host memory until svsr_debug_occupy_stop() sets it (or ~4 s pass: the kernel can never hang the device).  It stands in for a
// peer-waiting collective kernel of another process / stream that sits on some compute units while a training step runs
// (tests/test_gpu_cotenant.py: the fused encoder must fall back instead of dying, everything else must only get slower).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
unsigned* g_occupy_flag = nullptr;        // pinned host word
unsigned* g_occupy_flag_dev = nullptr;

__global__ __launch_bounds__(256) void k_debug_occupy(const unsigned* stop, unsigned* sink) {
    extern __shared__ unsigned occ_lds[];
    occ_lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
            __builtin_amdgcn_s_sleep(127);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull) break;
        }
    }
    __syncthreads();
    if (occ_lds[(threadIdx.x + 1) & 255] == 0xffffffffu) sink[0] = 1;          // (keeps the LDS allocation alive)
}
}  // namespace

extern "C" int svsr_debug_occupy_start(int workgroups, int lds_bytes, hipStream_t stream) {
    if (workgroups < 1 || workgroups > 1024 || lds_bytes < 1024 || lds_bytes > 160 * 1024) return SVSR_ERR_ARG;
    if (g_occupy_flag == nullptr) {
        void* h = nullptr;
        hipError_t e = hipHostMalloc(&h, 64, hipHostMallocMapped);
        if (e != hipSuccess) return (int)e;
        void* d = nullptr;
        e = hipHostGetDevicePointer(&d, h, 0);
        if (e != hipSuccess) { (void)hipHostFree(h); return (int)e; }
        g_occupy_flag = static_cast<unsigned*>(h);
        g_occupy_flag_dev = static_cast<unsigned*>(d);
    }
    *static_cast<volatile unsigned*>(g_occupy_flag) = 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_debug_occupy), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(k_debug_occupy, dim3(workgroups), dim3(256), lds_bytes, stream, g_occupy_flag_dev, g_occupy_flag_dev + 8);
    return svsr_check_launch();
}

// Effective shader clock: s_memtime counts shader cycles, s_memrealtime the constant 100 MHz reference; their ratio over a fixed block of
// MFMA work (one wave per SIMD) is the clock the compute units ran at while it executed.  bench.py reads it before and after its sustained leg.
namespace {
typedef __attribute__((ext_vector_type(8))) short cp_bf16x8;
typedef __attribute__((ext_vector_type(16))) float cp_f32x16;
__global__ __launch_bounds__(256) void k_clock_probe(long long* out, float* sink, int iters) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    cp_f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    cp_bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (short)(threadIdx.x * 5 + k * 321); b[k] = (short)(threadIdx.x * 3 + k * 77); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = (long long)(c1 - c0); out[blockIdx.x * 2 + 1] = (long long)(w1 - w0); }
}
}  // namespace

/* out: device int64 [blocks][2] = {shader cycles, 100 MHz ticks} of each workgroup's MFMA block (blocks x 256 threads, iters x 4 MFMAs per wave);
 * the effective clock in MHz is 100 * sum(cycles) / sum(ticks).  out needs 8 more bytes behind it (a sink word). */
extern "C" int svsr_clock_probe(int64_t* out, int blocks, int iters, hipStream_t stream) {
    if (out == nullptr || blocks < 1 || iters < 1) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_clock_probe, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<long long*>(out), reinterpret_cast<float*>(out + 2 * (size_t)blocks), iters);
    return svsr_check_launch();
}

extern "C" int svsr_debug_occupy_stop(void) {
    if (g_occupy_flag == nullptr) return SVSR_ERR_ARG;
    *static_cast<volatile unsigned*>(g_occupy_flag) = 1;
    return SVSR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// out[c] (+)= scale * sum_{r < nrows} ws[r * ld + c]   for c < n0 + n1; columns [0, n0) go to out0, [n0, n0 + n1) to out1.
// Block = 256 threads = CL columns x RL row lanes (RL = 256 / CL): lane rl adds rows rl, rl + RL, ... in increasing order,
// then the RL lane sums are added in increasing lane order — a fixed association whatever the launch order.
// ---------------------------------------------------------------------------------------------------------------------
template <int CL>
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ ws, int nrows, long ld, float* __restrict__ out0, long n0,
                                                 float* __restrict__ out1, long n1, int accumulate, float scale) {
    constexpr int RL = 256 / CL;
    __shared__ float sred[256];
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const long c = (long)blockIdx.x * CL + cl;
    const bool live = c < n0 + n1;
    float acc = 0.f;
    if (live) {
        const float* src = ws + c;
        int r = rl;
        for (; r + 7 * RL < nrows; r += 8 * RL) {      // eight independent loads in flight, added in row order
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = src[(long)(r + k * RL) * ld];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += t[k];
        }
        for (; r + 3 * RL < nrows; r += 4 * RL) {      // four
            const float a = src[(long)r * ld], b = src[(long)(r + RL) * ld], d = src[(long)(r + 2 * RL) * ld], e = src[(long)(r + 3 * RL) * ld];
            acc = (((acc + a) + b) + d) + e;
        }
        for (; r < nrows; r += RL) acc += src[(long)r * ld];
    }
    if (RL > 1) {
        sred[threadIdx.x] = acc;
        __syncthreads();
        if (rl != 0) return;
        acc = 0.f;
#pragma unroll 8
        for (int k = 0; k < RL; ++k) acc += sred[k * CL + cl];
    }
    if (!live) return;
    float* dst = c < n0 ? out0 + c : out1 + (c - n0);
    const float v = acc * scale;
    *dst = accumulate ? *dst + v : v;
}

// Up to COLSUM_MULTI_MAX column-sum problems as ONE launch (blockIdx.y = problem): the postponed parameter-gradient reductions of a layer's
// backward (LayerNorm weight / bias, linear biases) are 5-us launches of their own, 128 of them per sentence-level step on the weight-gradient
// stream.  Every problem keeps the column / row-lane shape svsr_colsum_rows would choose for it: the same additions in the same order.
#define COLSUM_MULTI_MAX 16
struct SvsrColsumEntry { const float* ws; float* out0; float* out1; int64_t ld, n0, n1; int nrows, accumulate; float scale; int cl; };      // 64 bytes
struct ColsumBatch { SvsrColsumEntry e[COLSUM_MULTI_MAX]; };
static_assert(sizeof(SvsrColsumEntry) == 64, "entry layout is part of the C ABI (include/syncvsr_hip.h)");

__global__ __launch_bounds__(256) void k_colsum_multi(const ColsumBatch b) {
    __shared__ float sred[256];
    const SvsrColsumEntry& e = b.e[blockIdx.y];
    const int CL = e.cl, RL = 256 / CL;                 // CL a power of two
    const long n = e.n0 + e.n1;
    if ((long)blockIdx.x * CL >= n) return;             // (the grid is as wide as the widest problem)
    const int cl = threadIdx.x & (CL - 1), rl = threadIdx.x / CL;
    const long c = (long)blockIdx.x * CL + cl;
    const bool live = c < n;
    const int nrows = e.nrows;
    const long ld = e.ld;
    float acc = 0.f;
    if (live) {
        const float* src = e.ws + c;
        int r = rl;
        for (; r + 7 * RL < nrows; r += 8 * RL) {
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = src[(long)(r + k * RL) * ld];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += t[k];
        }
        for (; r + 3 * RL < nrows; r += 4 * RL) {
            const float a = src[(long)r * ld], b2 = src[(long)(r + RL) * ld], d = src[(long)(r + 2 * RL) * ld], f = src[(long)(r + 3 * RL) * ld];
            acc = (((acc + a) + b2) + d) + f;
        }
        for (; r < nrows; r += RL) acc += src[(long)r * ld];
    }
    if (RL > 1) {
        sred[threadIdx.x] = acc;
        __syncthreads();
        if (rl != 0) return;
        acc = 0.f;
        for (int k = 0; k < RL; ++k) acc += sred[k * CL + cl];
    }
    if (!live) return;
    float* dst = c < e.n0 ? e.out0 + c : e.out1 + (c - e.n0);
    const float v = acc * e.scale;
    *dst = e.accumulate ? *dst + v : v;
}

static inline int colsum_cl(int nrows, long n) { return (nrows <= 8 || n >= 65536) ? 256 : (nrows <= 64 || n >= 8192) ? 32 : n >= 8 ? 8 : 1; }

/* entries: n records of 64 bytes {const float* ws; float* out0; float* out1; int64 ld, n0, n1; int32 nrows, accumulate; float scale; int32 reserved}
 * in HOST memory, each with svsr_colsum_rows's meaning; outputs of different records must not overlap.  One launch per 16 records. */
extern "C" int svsr_colsum_rows_multi(const void* entries, int n, hipStream_t stream) {
    if (entries == nullptr || n < 1) return SVSR_ERR_ARG;
    const SvsrColsumEntry* src = static_cast<const SvsrColsumEntry*>(entries);
    for (int i0 = 0; i0 < n; i0 += COLSUM_MULTI_MAX) {
        ColsumBatch b;
        const int m = n - i0 < COLSUM_MULTI_MAX ? n - i0 : COLSUM_MULTI_MAX;
        long gx = 1;
        for (int i = 0; i < m; ++i) {
            b.e[i] = src[i0 + i];
            const long cols = (long)b.e[i].n0 + (long)b.e[i].n1;
            if (b.e[i].nrows < 0 || cols <= 0 || b.e[i].ld < cols || (b.e[i].n1 > 0 && b.e[i].out1 == nullptr) || b.e[i].out0 == nullptr || b.e[i].ws == nullptr) return SVSR_ERR_ARG;
            b.e[i].cl = colsum_cl(b.e[i].nrows, cols);
            const long blocks = (cols + b.e[i].cl - 1) / b.e[i].cl;
            if (blocks > gx) gx = blocks;
        }
        for (int i = m; i < COLSUM_MULTI_MAX; ++i) b.e[i] = b.e[0];
        hipLaunchKernelGGL(k_colsum_multi, dim3((unsigned)gx, (unsigned)m), dim3(256), 0, stream, b);
        const int rc = svsr_check_launch();
        if (rc != SVSR_OK) return rc;
    }
    return SVSR_OK;
}

extern "C" int svsr_colsum_rows(const float* ws, int nrows, int64_t ld, float* out0, int64_t n0, float* out1, int64_t n1, int accumulate,
                                float scale, hipStream_t stream) {
    const long n = (long)n0 + (long)n1;
    if (nrows < 0 || n <= 0 || ld < n || (n1 > 0 && out1 == nullptr) || out0 == nullptr) return SVSR_ERR_ARG;
    // many columns, few rows (split-K slabs): one thread per column.  Few columns, many rows (statistics, losses): row lanes.
#define SVSR_COLSUM(CL_) hipLaunchKernelGGL(k_colsum<CL_>, dim3((unsigned)((n + CL_ - 1) / CL_)), dim3(256), 0, stream, ws, nrows, (long)ld, out0, (long)n0, out1, (long)n1, accumulate, scale)
    if (nrows <= 8 || n >= 65536) SVSR_COLSUM(256);
    else if (nrows <= 64 || n >= 8192) SVSR_COLSUM(32);
    else if (n >= 8) SVSR_COLSUM(8);
    else SVSR_COLSUM(1);
#undef SVSR_COLSUM
    return svsr_check_launch();
}
