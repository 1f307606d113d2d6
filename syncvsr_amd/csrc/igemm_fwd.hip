// Implicit-GEMM forward contraction on MFMA (gfx950): NHWC bf16 convolution forward / data-gradient and dense linear
// layers (a linear layer is the 1-tap, 1x1-pixel case).
//
// Replaces the ATen/MIOpen kernels the reference reaches through torch.nn:
//   nn.Conv2d 3x3 / 1x1 of the ResNet18 trunk   (reference LRW/video/src/tcn/models/resnet.py:8-16,36,53; timm twin)
//   nn.Linear of the BERT encoder and the heads (reference LRW/video/src/lightning.py:82,92,107,161,168)
// and their input-gradients.
//
// Block = 256 threads = 4 waves (2x2), tile BM positions x BN output channels, K step 64 channels of one tap.
// A (gathered activation rows) and B (weight rows) are staged global -> registers -> LDS (XOR-swizzled 16-byte
// chunks, conflict-free ds_read_b128 fragments), double-buffered with one barrier per K step; the next tile's global
// loads are issued before the MFMA block and written to LDS after it.  Epilogue: accumulators -> LDS (fp32) ->
// +bias +addend, exact GELU, 16-byte stores; optional per-channel BatchNorm partial sums.
#include "igemm_common.h"

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
    int* sTap = reinterpret_cast<int*>(sRow + BM);            // [27] dy | dx | tw

    const IgemmGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int chunk = tid & 7, r0 = tid >> 3;

    long a_base[AR];
    int a_y[AR], a_x[AR];
    unsigned row_ok = 0;          // bit i: A row i exists; bit 8+i: B row i exists
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + r0 + 32 * i;
        const bool ok = m < g.M;
        int n, a, b;
        decode_pos(g, ok ? m : 0, n, a, b);
        a_base[i] = (long)n * g.Hi * g.Wi;
        a_y[i] = a * g.S;
        a_x[i] = b * g.S;
        row_ok |= (ok ? 1u : 0u) << i;
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
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) { sTap[i] = g.dy[i]; sTap[9 + i] = g.dx[i]; sTap[18 + i] = g.tw[i]; }
    }
    const bf16_t* b_ptr[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + r0 + 32 * i;
        const bool ok = n < g.Co;
        b_ptr[i] = p.wt + (long)(ok ? n : 0) * g.wt_taps * g.Ci + chunk * 8;
        row_ok |= (ok ? 1u : 0u) << (8 + i);
    }
    __syncthreads();

    const int KT = g.ntaps * (g.Ci / BK);
    u32x4 ra[AR], rb[BR];
    unsigned ld_ok = 0;              // validity of the rows currently held in ra/rb (applied when they are written to LDS)
    int t_next = 0, c_next = 0;      // (tap, channel offset) of the tile the next load_tiles() fetches

    auto load_tiles = [&]() {
        const int dy = sTap[t_next], dx = sTap[9 + t_next], tw = sTap[18 + t_next];
        const int c0 = c_next;
        ld_ok = row_ok & 0xff00u;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int iy = a_y[i] + dy, ix = a_x[i] + dx;
            const bool ok = ((row_ok >> i) & 1u) && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
            const long pix = ok ? a_base[i] + (long)iy * g.Wi + ix : 0;
            ra[i] = *reinterpret_cast<const u32x4*>(p.in + pix * g.in_pitch + c0 + chunk * 8);
            ld_ok |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) rb[i] = *reinterpret_cast<const u32x4*>(b_ptr[i] + (long)tw * g.Ci + c0);
        c_next += BK;
        if (c_next >= g.Ci) { c_next = 0; ++t_next; }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int r = r0 + 32 * i;
            const bool ok = (ld_ok >> i) & 1u;
            u32x4 v = ra[i];
            v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
            *reinterpret_cast<u32x4*>(sA + buf * A_ELEMS + LDS_SWZ(r, chunk)) = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int r = r0 + 32 * i;
            const bool ok = (ld_ok >> (8 + i)) & 1u;
            u32x4 v = rb[i];
            v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
            *reinterpret_cast<u32x4*>(sB + buf * B_ELEMS + LDS_SWZ(r, chunk)) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tiles();
    store_tiles(0);
    __syncthreads();
    for (int it = 0; it < KT; ++it) {
        const int cur = it & 1;
        if (it + 1 < KT) load_tiles();
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

    // ---- per-channel statistics from the fp32 accumulators (rows outside M carry exact zeros) ----------------------
    float st_s[TN], st_q[TN];
    if (p.stats != nullptr) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; s += v; q += v * v; }
            st_s[j] = s + __shfl_xor(s, 32, 64);
            st_q[j] = q + __shfl_xor(q, 32, 64);
        }
    }

    // ---- epilogue: accumulators -> LDS fp32 [BM][BN] (tile buffers are dead after the last barrier) -----------------
    float* sOut = reinterpret_cast<float*>(smem_raw);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                sOut[row * BN + wn0 + j * 32 + (lane & 31)] = acc[i][j][r];
            }
    __syncthreads();
    const bool vec_pitch = (g.out_pitch & 7) == 0;
    constexpr int CV = BN / 8;
    for (int task = tid; task < BM * CV; task += 256) {
        const int r = task / CV, c8 = task - r * CV;
        const long off = sRow[r];
        const int n = n0 + c8 * 8;
        if (off < 0 || n >= g.Co) continue;
        float v[8];
        const f32x4 lo = *reinterpret_cast<const f32x4*>(sOut + r * BN + c8 * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(sOut + r * BN + c8 * 8 + 4);
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        if (vec_pitch && n + 8 <= g.Co) {
            if (p.bias != nullptr) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
                v[0] += b0[0]; v[1] += b0[1]; v[2] += b0[2]; v[3] += b0[3]; v[4] += b1[0]; v[5] += b1[1]; v[6] += b1[2]; v[7] += b1[3];
            }
            if (p.addend != nullptr) {
                float a8[8];
                unpack8(*reinterpret_cast<const u32x4*>(p.addend + off + n), a8);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += a8[k];
            }
            if (p.gelu) {
                if (p.out_pre != nullptr) *reinterpret_cast<u32x4*>(p.out_pre + off + n) = pack8(v);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = gelu_erf(v[k]);
            }
            if (p.out_f32) {
                float* o = reinterpret_cast<float*>(p.out) + off + n;
                *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
            } else {
                *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.out) + off + n) = pack8(v);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (n + k >= g.Co) break;
                float x = v[k];
                if (p.bias != nullptr) x += p.bias[n + k];
                if (p.addend != nullptr) x += bf2f(p.addend[off + n + k]);
                if (p.gelu) {
                    if (p.out_pre != nullptr) p.out_pre[off + n + k] = f2bf(x);
                    x = gelu_erf(x);
                }
                if (p.out_f32) reinterpret_cast<float*>(p.out)[off + n + k] = x;
                else reinterpret_cast<bf16_t*>(p.out)[off + n + k] = f2bf(x);
            }
        }
    }
    if (p.stats != nullptr) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_raw);   // [4 waves][WN][2]
#pragma unroll
        for (int j = 0; j < TN; ++j)
            if (lane < 32) {
                red[(wave * WN + j * 32 + lane) * 2 + 0] = st_s[j];
                red[(wave * WN + j * 32 + lane) * 2 + 1] = st_q[j];
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

template <int BM, int BN>
static int launch_fwd(const IgemmFwdArgs& a, hipStream_t stream) {
    const int gx = (a.g.M + BM - 1) / BM, gy = (a.g.Co + BN - 1) / BN;
    const size_t lds = (size_t)2 * (BM + BN) * 64 * sizeof(bf16_t) + (size_t)BM * sizeof(long) + 128;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_igemm_fwd<BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_igemm_fwd<BM, BN>), dim3(gx, gy), dim3(256), lds, stream, a);
    return svsr_check_launch();
}

static int igemm_fwd_tile_m(int M, int Co) {
    if (Co <= 64) return M >= 16384 ? 128 : 64;
    return M >= 8192 ? 128 : 64;
}

extern "C" int svsr_igemm_fwd(const void* in, const void* wt, void* out, void* out_pre, const float* bias, const void* addend,
                              float* stats, int Nimg, int Hi, int Wi, int Ci, int in_pitch, int Co, int Ho, int Wo, int out_pitch,
                              int Ha, int Wa, int S, int OS, int oy0, int ox0, int ntaps, int wt_taps, const int* dy, const int* dx,
                              const int* tw, int gelu, int out_f32, hipStream_t stream) {
    IgemmFwdArgs a;
    int rc = fill_geom(a.g, Nimg, Hi, Wi, Ci, in_pitch, Co, Ho, Wo, out_pitch, Ha, Wa, S, OS, oy0, ox0, ntaps, wt_taps, dy, dx, tw);
    if (rc != SVSR_OK) return rc;
    a.in = (const bf16_t*)in; a.wt = (const bf16_t*)wt; a.out = out; a.out_pre = (bf16_t*)out_pre;
    a.bias = bias; a.addend = (const bf16_t*)addend; a.stats = stats; a.gelu = gelu; a.out_f32 = out_f32;
    const int bm = igemm_fwd_tile_m(a.g.M, Co);
    if (bm == 128) return Co <= 64 ? launch_fwd<128, 64>(a, stream) : launch_fwd<128, 128>(a, stream);
    return launch_fwd<64, 64>(a, stream);
}
