// Implicit-GEMM contractions on MFMA (gfx950): NHWC bf16 convolution forward / data-gradient, dense linear
// layers (a linear layer is the 1-tap, 1x1-pixel case) and the weight-gradient contraction.
//
// Replaces the ATen/MIOpen kernels the reference reaches through torch.nn:
//   nn.Conv2d 3x3 / 1x1 of the ResNet18 trunk   (reference LRW/video/src/tcn/models/resnet.py:8-16,36,53; timm twin)
//   nn.Linear of the BERT encoder and the heads (reference LRW/video/src/lightning.py:82,92,107,161,168)
// and their autograd backward.
//
// Geometry shared by all three kernels.  The iteration space is M = Nimg*Ha*Wa "positions" (n, a, b):
//   source pixel  = (a*S + dy[t], b*S + dx[t])  in the [Hi x Wi] grid of `in`   (zero outside the grid)
//   target pixel  = (a*OS + oy0,  b*OS + ox0)   in the [Ho x Wo] grid of `out`
// forward conv:  S = stride, dy = kh - pad, OS = 1.      dgrad stride 1: same with transposed weights.
// dgrad stride 2: one launch per output parity class (oy0, ox0) with OS = 2, S = 1 and that class's taps.
// linear:        Hi = Wi = Ha = Wa = 1, one tap.
#include "common.h"

struct IgemmGeom {
    int Nimg, Hi, Wi, Ci, in_pitch;     // source grid; Ci = contraction channels per tap (multiple of 64)
    int Co, Ho, Wo, out_pitch;          // target grid; Co = output channels
    int Ha, Wa, S, OS, oy0, ox0;
    int ntaps, wt_taps;                 // taps iterated / taps physically present in the weight tensor
    int dy[9], dx[9], tw[9];            // tap offsets and the weight-tensor tap index each one uses
    int M;                              // Nimg*Ha*Wa  (< 2^24)
    float inv_hw, inv_w;                // 1/(Ha*Wa), 1/Wa for the float-reciprocal index decode
};

__device__ __forceinline__ void decode_pos(const IgemmGeom& g, int m, int& n, int& a, int& b) {
    // exact for m < 2^24: estimate with a float reciprocal, then correct by one
    const int hw = g.Ha * g.Wa;
    n = (int)((float)m * g.inv_hw);
    int rem = m - n * hw;
    if (rem < 0) { n--; rem += hw; } else if (rem >= hw) { n++; rem -= hw; }
    a = (int)((float)rem * g.inv_w);
    b = rem - a * g.Wa;
    if (b < 0) { a--; b += g.Wa; } else if (b >= g.Wa) { a++; b -= g.Wa; }
}

struct IgemmFwdArgs {
    IgemmGeom g;
    const bf16_t* in;
    const bf16_t* wt;      // [Co][wt_taps][Ci]
    void* out;             // bf16 or f32 pixels
    bf16_t* out_pre;       // optional pre-activation copy (GELU epilogue)
    const float* bias;     // optional [Co]
    const bf16_t* addend;  // optional bf16 pixels with the geometry of `out`, added before the activation
    float* stats;          // optional BatchNorm partials: atomically accumulated slots [SVSR_STAT_SLOTS][2][Co]
    int gelu, out_f32;
};

#define LDS_SWZ(row, chunk) ((row) * 64 + ((((chunk) ^ (((row) >> 1) & 7))) << 3))

template <int BM, int BN>
__global__ __launch_bounds__(256) void k_igemm_fwd(const IgemmFwdArgs p) {
    constexpr int BK = 64;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK;
    constexpr int AR = BM / 32, BR = BN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sA = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* sB = sA + 2 * A_ELEMS;
    long* sRow = reinterpret_cast<long*>(sB + 2 * B_ELEMS);   // [BM] target pixel offsets (elements), -1 = no row

    const IgemmGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int chunk = tid & 7, r0 = tid >> 3;

    long a_base[AR];
    int a_y[AR], a_x[AR];
    bool a_ok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + r0 + 32 * i;
        a_ok[i] = m < g.M;
        int n = 0, a = 0, b = 0;
        if (a_ok[i]) decode_pos(g, m, n, a, b);
        a_base[i] = (long)n * g.Hi * g.Wi;
        a_y[i] = a * g.S;
        a_x[i] = b * g.S;
    }
    for (int r = tid; r < BM; r += 256) {
        const int m = m0 + r;
        long off = -1;
        if (m < g.M) {
            int n, a, b;
            decode_pos(g, m, n, a, b);
            off = (((long)n * g.Ho + (a * g.OS + g.oy0)) * g.Wo + (b * g.OS + g.ox0)) * g.out_pitch;
        }
        sRow[r] = off;
    }
    const bf16_t* b_ptr[BR];
    bool b_ok[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + r0 + 32 * i;
        b_ok[i] = n < g.Co;
        b_ptr[i] = p.wt + (long)(b_ok[i] ? n : 0) * g.wt_taps * g.Ci + chunk * 8;
    }

    const int kc = g.Ci / BK;
    const int KT = g.ntaps * kc;
    u32x4 ra[AR], rb[BR];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    auto load_tiles = [&](int it) {
        const int t = it / kc, c0 = (it - t * kc) * BK;
        const int dy = g.dy[t], dx = g.dx[t], tw = g.tw[t];
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int iy = a_y[i] + dy, ix = a_x[i] + dx;
            const bool ok = a_ok[i] && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
            const bf16_t* src = p.in + (a_base[i] + (long)iy * g.Wi + ix) * g.in_pitch + c0 + chunk * 8;
            ra[i] = ok ? *reinterpret_cast<const u32x4*>(src) : zero4;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
            rb[i] = b_ok[i] ? *reinterpret_cast<const u32x4*>(b_ptr[i] + (long)tw * g.Ci + c0) : zero4;
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int r = r0 + 32 * i;
            *reinterpret_cast<u32x4*>(sA + buf * A_ELEMS + LDS_SWZ(r, chunk)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int r = r0 + 32 * i;
            *reinterpret_cast<u32x4*>(sB + buf * B_ELEMS + LDS_SWZ(r, chunk)) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int it = 0; it < KT; ++it) {
        const int cur = it & 1;
        if (it + 1 < KT) load_tiles(it + 1);
        const bf16_t* cA = sA + cur * A_ELEMS;
        const bf16_t* cB = sB + cur * B_ELEMS;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int ch = ks * 2 + (lane >> 5);
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm0 + i * 32 + (lane & 31);
                fa[i] = *reinterpret_cast<const bf16x8*>(cA + LDS_SWZ(row, ch));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn0 + j * 32 + (lane & 31);
                fb[j] = *reinterpret_cast<const bf16x8*>(cB + LDS_SWZ(row, ch));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (it + 1 < KT) store_tiles(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31] per 32x32 tile ----
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + (lane & 31);
        const bool n_ok = n < g.Co;
        const float bias = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const long off = sRow[row];
                if (off >= 0 && n_ok) {
                    float v = acc[i][j][r] + bias;
                    if (p.addend != nullptr) v += bf2f(p.addend[off + n]);
                    if (p.gelu) {
                        if (p.out_pre) p.out_pre[off + n] = f2bf(v);
                        v = gelu_erf(v);
                    }
                    if (p.out_f32) reinterpret_cast<float*>(p.out)[off + n] = v;
                    else reinterpret_cast<bf16_t*>(p.out)[off + n] = f2bf(v);
                }
            }
        }
    }
    if (p.stats != nullptr) {
        // per-channel sum / sum of squares of this block's rows (rows outside M carry exact zeros)
        float* red = reinterpret_cast<float*>(smem_raw);   // [4 waves][WN][2]; tiles are dead after the last barrier
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; s += v; q += v * v; }
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            if (lane < 32) {
                red[(wave * WN + j * 32 + lane) * 2 + 0] = s;
                red[(wave * WN + j * 32 + lane) * 2 + 1] = q;
            }
        }
        __syncthreads();
        for (int c = tid; c < BN; c += 256) {
            const int wcol = c / WN, cc = c - wcol * WN;      // waves (0,wcol) and (1,wcol) own this column
            const float s = red[((0 * 2 + wcol) * WN + cc) * 2 + 0] + red[((1 * 2 + wcol) * WN + cc) * 2 + 0];
            const float q = red[((0 * 2 + wcol) * WN + cc) * 2 + 1] + red[((1 * 2 + wcol) * WN + cc) * 2 + 1];
            if (n0 + c < g.Co) {
                const int slot = blockIdx.x & (SVSR_STAT_SLOTS - 1);
                atomicAdd(p.stats + ((long)slot * 2 + 0) * g.Co + n0 + c, s);
                atomicAdd(p.stats + ((long)slot * 2 + 1) * g.Co + n0 + c, q);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// weight gradient:  dW[co][tw[t]][ci] += sum_m dY[target(m)][co] * X[source(m, t)][ci]      (fp32 atomics)
// Both operands are stored position-major, so the MFMA fragments (8 consecutive positions per lane) come from
// LDS through the gfx950 transpose read ds_read_b64_tr_b16.
// ------------------------------------------------------------------------------------------------------------
struct IgemmWgradArgs {
    IgemmGeom g;
    const bf16_t* x;       // source pixels (pitch in_pitch)
    const bf16_t* dy;      // target pixels (pitch out_pitch)
    float* dw;             // [Co][wt_taps][Ci] fp32, accumulated
    int chunks_per_block;  // 64-position chunks each block reduces
    int use_tr;            // 1: ds_read_b64_tr_b16 fragments, 0: scalar 16-bit LDS reads (reference path)
};

#define WG_PITCH 80   // bf16 elements per LDS row: 64 channels + 16 pad (160 B) -> conflict-free 4x32B tr-reads

__device__ __forceinline__ bf16x4 lds_tr_read(const bf16_t* addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(addr));
}

// fragment for MFMA 32x32x16: lane l supplies matrix row (l&31) = channel, k = (l>>5)*8 .. +7 = positions.
template <bool USE_TR>
__device__ __forceinline__ bf16x8 load_frag_T(const bf16_t* tile, int ch0, int pos0, int lane) {
    bf16x8 f;
    if (USE_TR) {
        // 16-lane group gq: channels ch0 + (gq&1)*16 .., positions pos0 + (gq>>1)*8 ..; lane s of the group addresses
        // row (s>>2) of a [4 pos][16 ch] block at channel sub-block (s&3)*4 and receives channel column s.
        const int gq = lane >> 4, s = lane & 15;
        const bf16_t* base = tile + (pos0 + (gq >> 1) * 8 + (s >> 2)) * WG_PITCH + ch0 + (gq & 1) * 16 + (s & 3) * 4;
        const bf16x4 lo = lds_tr_read(base);
        const bf16x4 hi = lds_tr_read(base + 4 * WG_PITCH);
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    } else {
        const bf16_t* base = tile + (pos0 + (lane >> 5) * 8) * WG_PITCH + ch0 + (lane & 31);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = (short)base[k * WG_PITCH];
    }
    return f;
}

template <bool USE_TR>
__global__ __launch_bounds__(256) void k_igemm_wgrad(const IgemmWgradArgs p) {
    // block tile: 64 output channels x 64 contraction channels of ONE tap; 4 waves as 2x2 of 32x32
    __shared__ __attribute__((aligned(16))) bf16_t sm[2 * 64 * WG_PITCH];
    bf16_t* sY = sm;
    bf16_t* sX = sm + 64 * WG_PITCH;
    const IgemmGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co_tiles = (g.Co + 63) / 64, ci_tiles = g.Ci / 64;
    int rest = blockIdx.y;
    const int cot = rest % co_tiles; rest /= co_tiles;
    const int cit = rest % ci_tiles; rest /= ci_tiles;
    const int t = rest;
    const int co0 = cot * 64, ci0 = cit * 64;
    const int wco = (wave >> 1) * 32, wci = (wave & 1) * 32;
    const int dyt = g.dy[t], dxt = g.dx[t], tw = g.tw[t];
    const int chunk = tid & 7, r0 = tid >> 3;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int c_begin = blockIdx.x * p.chunks_per_block;
    int c_end = c_begin + p.chunks_per_block;
    const int total_chunks = (g.M + 63) / 64;
    if (c_end > total_chunks) c_end = total_chunks;

    for (int c = c_begin; c < c_end; ++c) {
        u32x4 vy[2], vx[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = c * 64 + r0 + 32 * i;
            vy[i] = zero4; vx[i] = zero4;
            if (m < g.M) {
                int n, a, b;
                decode_pos(g, m, n, a, b);
                const long opix = ((long)n * g.Ho + (a * g.OS + g.oy0)) * g.Wo + (b * g.OS + g.ox0);
                if (co0 + chunk * 8 < g.Co)
                    vy[i] = *reinterpret_cast<const u32x4*>(p.dy + opix * g.out_pitch + co0 + chunk * 8);
                const int iy = a * g.S + dyt, ix = b * g.S + dxt;
                if ((unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi)
                    vx[i] = *reinterpret_cast<const u32x4*>(p.x + (((long)n * g.Hi + iy) * g.Wi + ix) * g.in_pitch + ci0 + chunk * 8);
            }
        }
        __syncthreads();   // previous chunk's fragment reads are done
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4*>(sY + (r0 + 32 * i) * WG_PITCH + chunk * 8) = vy[i];
            *reinterpret_cast<u32x4*>(sX + (r0 + 32 * i) * WG_PITCH + chunk * 8) = vx[i];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 fa = load_frag_T<USE_TR>(sY, wco, ks * 16, lane);
            const bf16x8 fb = load_frag_T<USE_TR>(sX, wci, ks * 16, lane);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        }
    }
    // D[row = co][col = ci]
    const int ci = ci0 + wci + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < g.Co) atomicAdd(p.dw + ((long)co * g.wt_taps + tw) * g.Ci + ci, acc[r]);
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
static int fill_geom(IgemmGeom& g, int Nimg, int Hi, int Wi, int Ci, int in_pitch, int Co, int Ho, int Wo, int out_pitch,
                     int Ha, int Wa, int S, int OS, int oy0, int ox0, int ntaps, int wt_taps,
                     const int* dy, const int* dx, const int* tw) {
    if (ntaps < 1 || ntaps > 9 || Ci % 64 != 0 || Ci <= 0 || Co <= 0) return SVSR_ERR_ARG;
    if (in_pitch % 8 != 0 || Nimg <= 0 || Ha <= 0 || Wa <= 0) return SVSR_ERR_ARG;
    const long M = (long)Nimg * Ha * Wa;
    if (M >= (1L << 24)) return SVSR_ERR_ARG;
    g.Nimg = Nimg; g.Hi = Hi; g.Wi = Wi; g.Ci = Ci; g.in_pitch = in_pitch;
    g.Co = Co; g.Ho = Ho; g.Wo = Wo; g.out_pitch = out_pitch;
    g.Ha = Ha; g.Wa = Wa; g.S = S; g.OS = OS; g.oy0 = oy0; g.ox0 = ox0;
    g.ntaps = ntaps; g.wt_taps = wt_taps;
    for (int i = 0; i < 9; ++i) { g.dy[i] = i < ntaps ? dy[i] : 0; g.dx[i] = i < ntaps ? dx[i] : 0; g.tw[i] = i < ntaps ? tw[i] : 0; }
    g.M = (int)M;
    g.inv_hw = 1.0f / (float)(Ha * Wa);
    g.inv_w = 1.0f / (float)Wa;
    return SVSR_OK;
}

template <int BM, int BN>
static int launch_fwd(const IgemmFwdArgs& a, hipStream_t stream, int* grid_x_out) {
    const int gx = (a.g.M + BM - 1) / BM, gy = (a.g.Co + BN - 1) / BN;
    if (grid_x_out) *grid_x_out = gx;
    const size_t lds = (size_t)2 * (BM + BN) * 64 * sizeof(bf16_t) + (size_t)BM * sizeof(long);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_igemm_fwd<BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_igemm_fwd<BM, BN>), dim3(gx, gy), dim3(256), lds, stream, a);
    return svsr_check_launch();
}

extern "C" {

static int igemm_fwd_tile_m(int M, int Co) {
    if (Co <= 64) return M >= 16384 ? 128 : 64;
    return M >= 8192 ? 128 : 64;
}

int svsr_igemm_fwd(const void* in, const void* wt, void* out, void* out_pre, const float* bias, const void* addend, float* stats,
                   int Nimg, int Hi, int Wi, int Ci, int in_pitch, int Co, int Ho, int Wo, int out_pitch,
                   int Ha, int Wa, int S, int OS, int oy0, int ox0, int ntaps, int wt_taps,
                   const int* dy, const int* dx, const int* tw, int gelu, int out_f32, hipStream_t stream) {
    IgemmFwdArgs a;
    int rc = fill_geom(a.g, Nimg, Hi, Wi, Ci, in_pitch, Co, Ho, Wo, out_pitch, Ha, Wa, S, OS, oy0, ox0, ntaps, wt_taps, dy, dx, tw);
    if (rc != SVSR_OK) return rc;
    a.in = (const bf16_t*)in; a.wt = (const bf16_t*)wt; a.out = out; a.out_pre = (bf16_t*)out_pre;
    a.bias = bias; a.addend = (const bf16_t*)addend; a.stats = stats; a.gelu = gelu; a.out_f32 = out_f32;
    const int bm = igemm_fwd_tile_m(a.g.M, Co);
    if (bm == 128) return Co <= 64 ? launch_fwd<128, 64>(a, stream, nullptr) : launch_fwd<128, 128>(a, stream, nullptr);
    return launch_fwd<64, 64>(a, stream, nullptr);
}

int svsr_igemm_wgrad(const void* x, const void* dyp, float* dw,
                     int Nimg, int Hi, int Wi, int Ci, int in_pitch, int Co, int Ho, int Wo, int out_pitch,
                     int Ha, int Wa, int S, int OS, int oy0, int ox0, int ntaps, int wt_taps,
                     const int* dy, const int* dx, const int* tw, int use_tr, hipStream_t stream) {
    IgemmWgradArgs a;
    int rc = fill_geom(a.g, Nimg, Hi, Wi, Ci, in_pitch, Co, Ho, Wo, out_pitch, Ha, Wa, S, OS, oy0, ox0, ntaps, wt_taps, dy, dx, tw);
    if (rc != SVSR_OK) return rc;
    if (out_pitch % 8 != 0) return SVSR_ERR_ARG;
    a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dyp; a.dw = dw; a.use_tr = use_tr;
    const int tasks = ((Co + 63) / 64) * (Ci / 64) * ntaps;
    const int total_chunks = (a.g.M + 63) / 64;
    int splits = (2048 + tasks - 1) / tasks;            // aim for ~2048 blocks
    if (splits > total_chunks) splits = total_chunks;
    if (splits < 1) splits = 1;
    a.chunks_per_block = (total_chunks + splits - 1) / splits;
    splits = (total_chunks + a.chunks_per_block - 1) / a.chunks_per_block;
    if (use_tr) hipLaunchKernelGGL((k_igemm_wgrad<true>), dim3(splits, tasks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((k_igemm_wgrad<false>), dim3(splits, tasks), dim3(256), 0, stream, a);
    return svsr_check_launch();
}

}  // extern "C"
