// Shared device helpers for the SyncVSR gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (8 bf16 = 4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define SVSR_OK 0
#define SVSR_ERR_ARG 1001      // unsupported shape / argument
#define SVSR_ERR_LAUNCH 1002

// Result-preserving tuning knobs (runtime.hip): a process-wide table set through svsr_tune(); the library never reads the
// environment.
enum { SVSR_TUNE_IGEMM_TILE = 0, SVSR_TUNE_IGEMM_M128, SVSR_TUNE_WG_BLOCKS, SVSR_TUNE_W3_BLOCKS, SVSR_TUNE_LN_RPB,
       SVSR_TUNE_STEM_LDS_FWD, SVSR_TUNE_STEM_LDS_BWD, SVSR_TUNE_IGEMM_LDS_PAD, SVSR_TUNE_IGEMM_BN64_BELOW, SVSR_TUNE_WG_SHORT_K, SVSR_TUNE_IGEMM_KSPLIT, SVSR_TUNE_EPI_BATCHED, SVSR_TUNE_STEM_WG_PIPE, SVSR_TUNE_STEM_FWD_DMA, SVSR_TUNE_IGEMM_LIN_BN64, SVSR_TUNE_P8, SVSR_TUNE_P8_GRID, SVSR_TUNE_P8_MIN_ITEMS, SVSR_TUNE_P8_PH, SVSR_TUNE_P8_STAGGER, SVSR_TUNE_WG_IMGMAJOR, SVSR_TUNE_P8_BN64, SVSR_TUNE_IGEMM_NS64, SVSR_TUNE_WG_UNITS, SVSR_TUNE_WG_UNIT_MAX, SVSR_TUNE_WG_UNIT_MIN, SVSR_TUNE_IGEMM_KSPLIT128, SVSR_TUNE_WG_XCD, SVSR_TUNE_W3_WAVES, SVSR_TUNE_REDUCE_CUS, SVSR_TUNE_W3_DENSE, SVSR_TUNE_P8_WIDE, SVSR_TUNE_N };
int svsr_tune_get(int id);
// compute units of the device (runtime.hip).  Persistent kernels size their grids and static tile lists with it.
int svsr_stream_cus(hipStream_t stream);
// compute units ASSUMED where the split of a reduction is planned (weight-gradient unit lists, the layer1 kernel's BatchNorm partial rows): the
// tuning knob reduce_cus (256), not the device's count — the association of every sum, and with it every bit of a training run, then is the
// same on a partition of another size (a checkpoint resumed there continues bit for bit); 0 = follow the device (runtime.hip)
int svsr_reduction_cus();

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((unsigned)u) << 16); }

// float -> bfloat16, round-to-nearest-even (torch's cast): ONE v_cvt_pk_bf16_f32 per pair on gfx950.  (The integer formulation
// u += 0x7fff + lsb with its NaN test cost ~7 VALU operations per element: the output passes of the contraction epilogues and of the
// BatchNorm / stem kernels were VALU-bound on it.  Results are identical for every non-NaN input; NaNs stay NaNs.)
typedef __bf16 svsr_bf16x2 __attribute__((ext_vector_type(2)));
typedef float svsr_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    const svsr_f32x2 v = {lo, hi};
    const svsr_bf16x2 b = __builtin_convertvector(v, svsr_bf16x2);
    return __builtin_bit_cast(unsigned, b);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

// Standard-normal CDF Phi(x) and e = exp(-x^2/2) for the exact (erf) GELU of nn.GELU().
// erfc(z) = t*(a1 + t*(a2 + t*(a3 + t*(a4 + t*a5)))) * exp(-z^2), t = 1/(1 + p*z), z >= 0   (Abramowitz & Stegun 7.1.26,
// |error| <= 1.5e-7 — three decimal orders below bf16 resolution); evaluated on |x| and mirrored, so the negative tail has
// no 1 - erf cancellation.  ~15 VALU ops (v_rcp_f32 + v_exp_f32) instead of libdevice erff's ~45: the stem's
// BatchNorm+GELU+pool passes were VALU-bound on erff, not HBM-bound.
// (every multiply-add spelled out: the same source inlined into two kernels must round the same way — under -ffp-contract=fast the
// compiler otherwise picks its own contractions per call site, and csrc/enc_fused.hip is tested bit for bit against the launch chain)
__device__ __forceinline__ void normal_cdf_exp(float x, float& cdf, float& ex) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    ex = __expf(-(z * z));
    float poly = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    poly = __builtin_fmaf(t, poly, 1.421413741f);
    poly = __builtin_fmaf(t, poly, -0.284496736f);
    poly = __builtin_fmaf(t, poly, 0.254829592f);
    const float hp = 0.5f * (t * poly);
    const float e = hp * ex;
    cdf = x >= 0.f ? __builtin_fmaf(-hp, ex, 1.0f) : e;      // (1 - e as ONE explicit fma: left to the compiler it is contracted in some call sites only)
}
__device__ __forceinline__ float gelu_erf(float x) {
    float cdf, ex;
    normal_cdf_exp(x, cdf, ex);
    return x * cdf;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    // d/dx [x * Phi(x)] = Phi(x) + x * phi(x),  phi(x) = exp(-x^2/2) / sqrt(2 pi)
    float cdf, ex;
    normal_cdf_exp(x, cdf, ex);
    return __builtin_fmaf(x * 0.39894228040143267794f, ex, cdf);
}

// Swish / SiLU (LRS front-end and Conformer convolution module: transformer/convolution.py:78-83)
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float swish(float x) { return x * sigmoid_fast(x); }
__device__ __forceinline__ float swish_grad(float x) {
    const float s = sigmoid_fast(x);
    return s * (1.0f + x * (1.0f - s));
}
// activation codes shared by the BatchNorm/stem passes: 0 none, 1 ReLU (BN passes) or GELU (stem), 2 Swish
#define SVSR_ACT_NONE 0
#define SVSR_ACT_RELU 1
#define SVSR_ACT_GELU 1
#define SVSR_ACT_SWISH 2

// Dropout: counter-based keep decision, reproducible from (seed, site, element index) alone, so the backward regenerates
// the mask instead of storing it and the parity tests can replay it on the host (syncvsr_amd/dropout.py is the numpy twin).
// keep(idx) <=> hash(key, idx) >= thresh, thresh = p * 2^32; kept values are scaled by 1/(1-p).
struct DropArgs { const unsigned* seed; unsigned site; unsigned thresh; float scale; };
__device__ __forceinline__ unsigned svsr_mix(unsigned h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ unsigned drop_key(const DropArgs& d) { return svsr_mix(d.seed[0] * 0x9E3779B9u + d.site * 0x7F4A7C15u + 0x165667B1u); }
__device__ __forceinline__ bool drop_keep(unsigned key, unsigned thresh, unsigned idx) { return svsr_mix(idx * 2654435761u + key) >= thresh; }

// 64-lane all-reduce without LDS-crossbar permutes where the ISA offers a cheaper path (a ds_bpermute butterfly costs ~55 cycles a step in a
// dependent chain: the LayerNorm of eight rows in csrc/enc_fused.hip spent 5,300 cycles in 96 of them): DPP quad permutes (partners
// l^1, l^2), row_half_mirror and row_mirror (the quads / octets already hold equal values, so l <-> 7-l and l <-> 15-l pair the same groups
// as l^4 and l^8), ds_swizzle for l^16 and v_permlane32_swap for l^32.  Every lane ends with the same bits; the association is
// ((((pairs) quads) octets) rows) halves — fixed, whatever the launch.
__device__ __forceinline__ float dpp_xor1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_xor2(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_half_mirror(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_mirror(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_ror8(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true)); }
__device__ __forceinline__ float swz_xor16(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)); }
__device__ __forceinline__ float swz_lane0_of_32(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x0000)); }

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_xor1(v);
    v += dpp_xor2(v);
    v += dpp_half_mirror(v);
    v += dpp_mirror(v);
    v += swz_xor16(v);
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);      // {lower half's value, upper half's value} in every lane
    return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_xor1(v));
    v = fmaxf(v, dpp_xor2(v));
    v = fmaxf(v, dpp_half_mirror(v));
    v = fmaxf(v, dpp_mirror(v));
    v = fmaxf(v, swz_xor16(v));
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    return fmaxf(__int_as_float(r[0]), __int_as_float(r[1]));
}

// 16-byte vector of 8 bf16 as 4 dwords
struct __attribute__((aligned(16))) u32x4 { unsigned x, y, z, w; };

__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
    v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]); v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
    return v;
}

static inline DropArgs svsr_make_drop(const unsigned* seed, unsigned site, float p) {
    DropArgs d;
    const bool on = seed != nullptr && p > 0.f;
    d.seed = on ? seed : nullptr;
    d.site = site;
    const double t = (double)p * 4294967296.0;
    d.thresh = on ? (unsigned)(t > 4294967295.0 ? 4294967295.0 : t) : 0u;
    d.scale = on ? 1.0f / (1.0f - p) : 1.0f;
    return d;
}

// Counted waits of the LDS-DMA pipelines (s_waitcnt vmcnt(n), n > 0: tiles stay in flight across barriers).  A wait that counts one piece too
// few, or a ring slot re-staged one phase too early, reads LDS bytes a DMA has not written yet — and passes every test whenever the DMA happens
// to land first (guide: "place reads by the vmcnt / barrier count, never by clean runs").  The build variant -DSVSR_SYNC_DEBUG
// (python -m syncvsr_amd.build --variant syncdbg -> libsyncvsr_hip_syncdbg.so) turns EVERY counted wait into vmcnt(0): no tile is in flight
// when anything is read.  tests/test_gpu_syncdbg.py runs the benchmark shapes on both libraries and asserts bit-identical outputs — a
// difference is a latent race.
#ifdef SVSR_SYNC_DEBUG
#define SVSR_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define SVSR_WAIT_VM_BARRIER(n) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define SVSR_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define SVSR_WAIT_VM_BARRIER(n) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(n) : "memory")
#endif

static inline int svsr_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SVSR_OK : (int)e;
}
