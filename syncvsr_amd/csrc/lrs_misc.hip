// LRS-only passes that are not contractions (gfx950, wave64):
//   * Conformer convolution module core: GLU -> depthwise Conv1d(k, pad (k-1)/2) with BatchNorm1d partial sums, and its
//     backward (reference LRS/video/espnet/nets/pytorch_backend/transformer/convolution.py:56-75)
//   * CTC loss (torch.nn.CTCLoss(reduction="sum", zero_infinity=True) / batch, blank 0; ctc.py:44-74,83-151) with the
//     gradient with respect to the logits
//   * decoder token embedding * sqrt(d) + sinusoidal table (transformer/embedding.py:78-89) and its backward
//   * ESPnet label-smoothing KL loss + token accuracy (label_smoothing_loss.py:41-63, nets_utils.py:303-323)
//   * y = alpha * x
#include "common.h"

// ---------------------------------------------------------------------------------------------------------------------
// GLU + depthwise conv.  u [B*T][2D] bf16 (value | gate), w [D][K] fp32, bias [D] -> c [B*T][D] bf16, BN partial sums.
// block = (DW_TT-frame tile, 64-channel group, batch item); the gated input (tile + halo) is staged in LDS as fp32.
// ---------------------------------------------------------------------------------------------------------------------
#define DW_TT 32                 // frames per workgroup tile (64 left each thread a serial loop of 16 frames x 31 taps and 576 workgroups)
#define DW_FPT (DW_TT / 4)       // frames per thread: four wave-quarters of a tile
#define DW_MAXK 31

__device__ __forceinline__ void dw_stage_glu(float* sG, const bf16_t* __restrict__ u, int b, int T, int D, int t_lo, int rows, int c0) {
    // sG[r][ch] = a * sigmoid(gate) at frame t_lo + r (zero outside [0,T)), r < rows
    for (int idx = threadIdx.x; idx < rows * 8; idx += 256) {
        const int r = idx >> 3, c8 = idx & 7, t = t_lo + r;
        float g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = 0.f;
        if (t >= 0 && t < T) {
            const bf16_t* p = u + ((long)b * T + t) * (2 * D) + c0 + c8 * 8;
            float av[8], bv[8];
            unpack8(*reinterpret_cast<const u32x4*>(p), av);
            unpack8(*reinterpret_cast<const u32x4*>(p + D), bv);
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = av[k] * sigmoid_fast(bv[k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) sG[r * 64 + c8 * 8 + k] = g[k];
    }
}

// KT = 31 (the shipped cnn_module_kernel): a thread's DW_FPT frames need DW_FPT + 30 rows of its channel — read from LDS ONCE into registers
// (38 reads instead of 248 / 496 one-word reads, the time of these kernels), then the same multiply-adds in the same order; KT = 0: any odd K <= 31
template <int KT>
__global__ __launch_bounds__(256) void k_glu_dwconv_fwd(const bf16_t* __restrict__ u, const float* __restrict__ w, const float* __restrict__ bias,
                                                        bf16_t* __restrict__ c, float* __restrict__ stats, int B, int T, int D, int K) {
    __shared__ float sG[(DW_TT + DW_MAXK - 1) * 64];
    __shared__ float sRed[4][2][64];
    const int pad = (K - 1) / 2;
    const int t0 = blockIdx.x * DW_TT, c0 = blockIdx.y * 64, b = blockIdx.z;
    const int rows = DW_TT + K - 1;
    dw_stage_glu(sG, u, b, T, D, t0 - pad, rows, c0);
    __syncthreads();
    const int ch = threadIdx.x & 63, tq = threadIdx.x >> 6;
    float wk[DW_MAXK];
#pragma unroll
    for (int k = 0; k < DW_MAXK; ++k) wk[k] = k < K ? w[(long)(c0 + ch) * K + k] : 0.f;
    const float bs = bias[c0 + ch];
    float s1 = 0.f, s2 = 0.f;
    if (KT == DW_MAXK) {
        float gr[DW_FPT + DW_MAXK - 1];
#pragma unroll
        for (int j = 0; j < DW_FPT + DW_MAXK - 1; ++j) gr[j] = sG[(tq * DW_FPT + j) * 64 + ch];
#pragma unroll
        for (int i = 0; i < DW_FPT; ++i) {
            const int t = t0 + tq * DW_FPT + i;
            if (t < T) {
                float acc = bs;
#pragma unroll
                for (int k = 0; k < DW_MAXK; ++k) acc += wk[k] * gr[i + k];
                c[((long)b * T + t) * D + c0 + ch] = f2bf(acc);
                s1 += acc; s2 += acc * acc;
            }
        }
    } else {
        for (int tl = tq * DW_FPT; tl < tq * DW_FPT + DW_FPT; ++tl) {
            const int t = t0 + tl;
            if (t >= T) break;
            float acc = bs;
#pragma unroll
            for (int k = 0; k < DW_MAXK; ++k)
                if (k < K) acc += wk[k] * sG[(tl + k) * 64 + ch];
            c[((long)b * T + t) * D + c0 + ch] = f2bf(acc);
            s1 += acc; s2 += acc * acc;
        }
    }
    if (stats != nullptr) {
        sRed[tq][0][ch] = s1; sRed[tq][1][ch] = s2;
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6;
            const float v = ((sRed[0][which][ch] + sRed[1][which][ch]) + sRed[2][which][ch]) + sRed[3][which][ch];
            const long row = (long)blockIdx.z * gridDim.x + blockIdx.x;      // one row of partials per (clip, DW_TT-frame tile)
            stats[(row * 2 + which) * D + c0 + ch] = v;
        }
    }
}

// backward: dc [B*T][D] -> du [B*T][2D]; per-block partial dw/dbias into part[split][D*(K+1)] (reduced by k_dw_reduce)
template <int KT>
__global__ __launch_bounds__(256) void k_glu_dwconv_bwd(const bf16_t* __restrict__ dc, const bf16_t* __restrict__ u, const float* __restrict__ w,
                                                        bf16_t* __restrict__ du, float* __restrict__ part, int B, int T, int D, int K, int ntt) {
    constexpr int STAGE_F = 2 * (DW_TT + DW_MAXK - 1) * 64, RED_F = 4 * (DW_MAXK + 1) * 64;       // staging (g, dc) / final reduction
    __shared__ float sAll[STAGE_F > RED_F ? STAGE_F : RED_F];
    float* sG = sAll;
    float* sDC = sAll + (DW_TT + DW_MAXK - 1) * 64;
    const int pad = (K - 1) / 2;
    const int c0 = blockIdx.x * 64, split = blockIdx.y, nsplit = gridDim.y;
    const int ch = threadIdx.x & 63, tq = threadIdx.x >> 6;
    const int rows = DW_TT + K - 1;
    float wk[DW_MAXK], dwk[DW_MAXK];
#pragma unroll
    for (int k = 0; k < DW_MAXK; ++k) { wk[k] = k < K ? w[(long)(c0 + ch) * K + k] : 0.f; dwk[k] = 0.f; }
    float dbs = 0.f;
    for (int tile = split; tile < B * ntt; tile += nsplit) {
        const int b = tile / ntt, t0 = (tile - b * ntt) * DW_TT;
        __syncthreads();
        dw_stage_glu(sG, u, b, T, D, t0 - pad, rows, c0);
        for (int idx = threadIdx.x; idx < rows * 8; idx += 256) {
            const int r = idx >> 3, c8 = idx & 7, t = t0 - pad + r;
            float g[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = 0.f;
            if (t >= 0 && t < T) unpack8(*reinterpret_cast<const u32x4*>(dc + ((long)b * T + t) * D + c0 + c8 * 8), g);
#pragma unroll
            for (int k = 0; k < 8; ++k) sDC[r * 64 + c8 * 8 + k] = g[k];
        }
        __syncthreads();
        // this thread's DW_FPT (value, gate) pairs of u, requested together ahead of the frame loop: loaded inside it, every frame paid its
        // own dependent global round trip (78 us per layer for 12 MB of traffic)
        unsigned short ua[DW_FPT], ug[DW_FPT];
#pragma unroll
        for (int i = 0; i < DW_FPT; ++i) {
            const int t = t0 + tq * DW_FPT + i;
            const long o = ((long)b * T + (t < T ? t : T - 1)) * (2 * D) + c0 + ch;
            ua[i] = u[o];
            ug[i] = u[o + D];
        }
        float gr[DW_FPT + DW_MAXK - 1], dcr[DW_FPT + DW_MAXK - 1];
        if (KT == DW_MAXK) {
#pragma unroll
            for (int j = 0; j < DW_FPT + DW_MAXK - 1; ++j) {
                gr[j] = sG[(tq * DW_FPT + j) * 64 + ch];
                dcr[j] = sDC[(tq * DW_FPT + j) * 64 + ch];
            }
        }
#pragma unroll
        for (int i = 0; i < DW_FPT; ++i) {
            const int tl = tq * DW_FPT + i;
            const int t = t0 + tl;
            if (t >= T) continue;
            // c[t] = sum_k g[t+k-pad] w[k]  =>  dg[t] = sum_k dc[t-k+pad] w[k];  dw[k] += dc[t] g[t+k-pad]
            float dct, dg = 0.f;
            if (KT == DW_MAXK) {
                dct = dcr[i + (DW_MAXK - 1) / 2];
#pragma unroll
                for (int k = 0; k < DW_MAXK; ++k) {
                    dg += wk[k] * dcr[i + (DW_MAXK - 1) - k];
                    dwk[k] += dct * gr[i + k];
                }
            } else {
                dct = sDC[(tl + pad) * 64 + ch];
#pragma unroll
                for (int k = 0; k < DW_MAXK; ++k)
                    if (k < K) {
                        dg += wk[k] * sDC[(tl + 2 * pad - k) * 64 + ch];
                        dwk[k] += dct * sG[(tl + k) * 64 + ch];
                    }
            }
            dbs += dct;
            const long o = ((long)b * T + t) * (2 * D) + c0 + ch;
            const float av = bf2f(ua[i]), sg = sigmoid_fast(bf2f(ug[i]));
            du[o] = f2bf(dg * sg);
            du[o + D] = f2bf(dg * av * sg * (1.f - sg));
        }
    }
    // reduce the four frame-quarters of the block through LDS, then one plain store per (channel, tap)
    __syncthreads();
    float* sR = sAll;    // [4][K+1][64]
#pragma unroll
    for (int k = 0; k < DW_MAXK; ++k)
        if (k < K) sR[(tq * (K + 1) + k) * 64 + ch] = dwk[k];
    sR[(tq * (K + 1) + K) * 64 + ch] = dbs;
    __syncthreads();
    for (int idx = threadIdx.x; idx < (K + 1) * 64; idx += 256) {
        const int k = idx >> 6, cc = idx & 63;
        const float v = sR[(0 * (K + 1) + k) * 64 + cc] + sR[(1 * (K + 1) + k) * 64 + cc] + sR[(2 * (K + 1) + k) * 64 + cc] + sR[(3 * (K + 1) + k) * 64 + cc];
        float* dst = part + (long)split * D * (K + 1);
        if (k < K) dst[(long)(c0 + cc) * K + k] = v;
        else dst[(long)D * K + c0 + cc] = v;
    }
}

__global__ void k_dw_reduce(const float* __restrict__ part, int nsplit, int D, int K, float* __restrict__ dw, float* __restrict__ dbias) {
    const int n = D * (K + 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float s = 0.f;
        int p = 0;
        for (; p + 8 <= nsplit; p += 8) {            // eight loads in flight, added in split order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = part[(long)(p + k) * n + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; p < nsplit; ++p) s += part[(long)p * n + i];
        if (i < D * K) dw[i] += s;
        else dbias[i - D * K] += s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// row log-sum-exp over V columns of fp32 logits (pitch ld): one wave per row
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_row_lse(const float* __restrict__ z, int ld, int R, int V, float* __restrict__ lse) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
        const float* p = z + (long)row * ld;
        float m = -INFINITY;
        for (int v = lane; v < V; v += 64) m = fmaxf(m, p[v]);
        m = wave_max(m);
        float s = 0.f;
        for (int v = lane; v < V; v += 64) s += __expf(p[v] - m);
        s = wave_sum(s);
        if (lane == 0) lse[row] = m + __logf(s);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// CTC: one block per batch item walks the lattice (alpha forward, beta backward) in log space.
// ext[s] = blank for even s, y[s/2] for odd s;  S = 2*len+1.  ab [B][T][Smax] workspace: alpha, then posterior occupancy.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lae(float a, float b) {       // log(exp(a) + exp(b)) with -inf handling
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    const float m = fmaxf(a, b);
    return m + __logf(__expf(a - m) + __expf(b - m));
}

__global__ __launch_bounds__(256) void k_ctc_lattice(const float* __restrict__ z, int ld, const float* __restrict__ lse,
                                                     const long* __restrict__ labels, int Lmax, const int* __restrict__ ilen,
                                                     int B, int T, int Smax, float* __restrict__ ab, float* __restrict__ nll_out) {
    extern __shared__ float sm[];             // prev[Smax], cur[Smax], ext (int)[Smax]
    float* prev = sm;
    float* cur = sm + Smax;
    int* ext = reinterpret_cast<int*>(sm + 2 * Smax);
    __shared__ int s_len;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int n = 0;
        while (n < Lmax && labels[(long)b * Lmax + n] >= 0) ++n;       // targets are padded with -1 at the tail
        s_len = n;
    }
    __syncthreads();
    const int len = s_len, S = 2 * len + 1, Tb = ilen[b];
    for (int s = tid; s < Smax; s += 256) ext[s] = (s & 1) && s < S ? (int)labels[(long)b * Lmax + (s >> 1)] : 0;
    __syncthreads();
    const float* zb = z + (long)b * T * ld;
    const float* lb = lse + (long)b * T;
    float* abb = ab + (long)b * T * Smax;
    // alpha
    for (int s = tid; s < S; s += 256) {
        float v = -INFINITY;
        if (s == 0) v = zb[0] - lb[0];
        else if (s == 1) v = zb[ext[1]] - lb[0];
        prev[s] = v;
        abb[s] = v;
    }
    __syncthreads();
    for (int t = 1; t < Tb; ++t) {
        for (int s = tid; s < S; s += 256) {
            float v = prev[s];
            if (s >= 1) v = lae(v, prev[s - 1]);
            if (s >= 2 && (s & 1) && ext[s] != ext[s - 2]) v = lae(v, prev[s - 2]);
            v = v == -INFINITY ? v : v + zb[(long)t * ld + ext[s]] - lb[t];
            cur[s] = v;
            abb[(long)t * Smax + s] = v;
        }
        __syncthreads();
        float* tmp = prev; prev = cur; cur = tmp;
    }
    float nll;
    {
        const float a1 = prev[S - 1], a2 = S > 1 ? prev[S - 2] : -INFINITY;
        nll = -lae(a1, a2);
    }
    __syncthreads();
    const bool inf = !(nll < INFINITY);               // zero_infinity=True: infeasible alignments contribute 0 and no gradient
    if (tid == 0) nll_out[b] = inf ? 0.f : nll;          // the loss is the fixed-order sum of nll_out / B (svsr_colsum_rows)
    // beta (including the emission at t), combined into occupancy  occ[t][s] = exp(alpha + beta + nll - lp[t][ext s])
    for (int s = tid; s < S; s += 256) {
        float v = -INFINITY;
        if (s == S - 1 || s == S - 2) v = zb[(long)(Tb - 1) * ld + ext[s]] - lb[Tb - 1];
        prev[s] = v;
    }
    __syncthreads();
    for (int t = Tb - 1; t >= 0; --t) {
        if (t < Tb - 1) {
            for (int s = tid; s < S; s += 256) {
                float v = prev[s];
                if (s + 1 < S) v = lae(v, prev[s + 1]);
                if (s + 2 < S && (s & 1) && ext[s] != ext[s + 2]) v = lae(v, prev[s + 2]);
                v = v == -INFINITY ? v : v + zb[(long)t * ld + ext[s]] - lb[t];
                cur[s] = v;
            }
            __syncthreads();
            float* tmp = prev; prev = cur; cur = tmp;
        }
        for (int s = tid; s < S; s += 256) {
            const float al = abb[(long)t * Smax + s], be = prev[s];
            float occ = 0.f;
            if (!inf && al > -INFINITY && be > -INFINITY) occ = __expf(al + be + nll - (zb[(long)t * ld + ext[s]] - lb[t]));
            abb[(long)t * Smax + s] = occ;
        }
        __syncthreads();
    }
}

// dlogits[b,t,v] = g/B * (softmax[v] - sum_{s: ext s = v} occ[t][s]) for t < ilen[b] (and a feasible target), else 0
__global__ __launch_bounds__(256) void k_ctc_grad(const float* __restrict__ z, int ld, const float* __restrict__ lse,
                                                  const long* __restrict__ labels, int Lmax, const int* __restrict__ ilen,
                                                  const float* __restrict__ ab, const float* __restrict__ nll, int B, int T, int V, int Smax,
                                                  const float* __restrict__ gout, bf16_t* __restrict__ dz, int ldo) {
    extern __shared__ float sAcc[];          // [V] occupancy per class, then [Smax] staged (class, occupancy) pairs
    const int row = blockIdx.x, b = row / T, t = row - b * T;
    bf16_t* o = dz + (long)row * ldo;
    const bool live = t < ilen[b] && nll[b] != 0.f;
    if (!live) {
        for (int v = threadIdx.x; v < ldo; v += 256) o[v] = 0;
        return;
    }
    for (int v = threadIdx.x; v < V; v += 256) sAcc[v] = 0.f;
    __syncthreads();
    int len = 0;
    while (len < Lmax && labels[(long)b * Lmax + len] >= 0) ++len;
    const int S = 2 * len + 1;
    const float* occ = ab + ((long)b * T + t) * Smax;
    // lattice states that emit the same class (every blank; repeated labels) are added in increasing state order by the
    // thread of the FIRST such state — no atomics, so the gradient is reproducible
    int* sExt = reinterpret_cast<int*>(sAcc + V);
    float* sOcc = sAcc + V + Smax;
    for (int s = threadIdx.x; s < S; s += 256) {
        sExt[s] = (s & 1) ? (int)labels[(long)b * Lmax + (s >> 1)] : 0;
        sOcc[s] = occ[s];
    }
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += 256) {
        const int e = sExt[s];
        bool first = true;
        for (int s2 = 0; s2 < s; ++s2) first = first && sExt[s2] != e;
        if (!first) continue;
        float acc = 0.f;
        for (int s2 = s; s2 < S; ++s2) acc += sExt[s2] == e ? sOcc[s2] : 0.f;
        if (e >= 0 && e < V) sAcc[e] = acc;
    }
    __syncthreads();
    const float g = gout[0] / (float)B, l = lse[row];
    const float* zr = z + (long)row * ld;
    for (int v = threadIdx.x; v < ldo; v += 256) o[v] = v < V ? f2bf(g * (__expf(zr[v] - l) - sAcc[v])) : (bf16_t)0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Decoder / CTC targets of a batch from its padded labels — add_sos_eos (reference espnet/nets/pytorch_backend/transformer/add_sos_eos.py:10-31
// as called by e2e_asr_transformer.py:203-215) in ONE launch; as torch index operations it was ~20 small launches per step on the step's
// stream (0.12-0.15 ms of the sentence-level step).  label [B][L] int64; ignore_id entries are DROPPED wherever they sit in a row, as the
// reference's `y[y != ignore_id]` does (a tail is the usual case); with n live tokens in a row:
//   labels[b] = [live.., -1 pad] (CTC);  ys_in[b] = [eos, live.., eos pad];  ys_out[b] = [live.., eos, ignore_id pad]
// A live token outside [1, odim) is what torch's Embedding / CTCLoss would stop on with a device assert; here nothing traps (a trap kills
// the HIP context with an opaque launch failure, on every replay of a recorded step): the token is replaced by eos — every later kernel
// stays inside its tables — and a STICKY error word is set that svsr_lrs_target_errors returns (E2E.check_targets / TrainStep.state raise
// a descriptive error from it).
// ---------------------------------------------------------------------------------------------------------------------
__device__ unsigned g_lrs_target_err = 0;
__global__ __launch_bounds__(256) void k_lrs_targets(const long* __restrict__ label, int L, int odim, long ignore_id, long eos,
                                                     long* __restrict__ labels, long* __restrict__ ys_in, long* __restrict__ ys_out) {
    __shared__ int s_w[4], s_run, s_bad;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_run = 0; s_bad = 0; }
    __syncthreads();
    for (int base = 0; base < L; base += 256) {           // stable compaction: position of a live token = live tokens in front of it
        const int l = base + tid;
        long v = l < L ? label[(long)b * L + l] : ignore_id;
        const bool live = l < L && v != ignore_id;
        if (live && (v < 1 || v >= odim)) { s_bad = 1; v = eos; }
        const unsigned long long m = __ballot(live);
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int pos = s_run + __popcll(m & ((1ull << lane) - 1ull));
        for (int k = 0; k < wave; ++k) pos += s_w[k];
        if (live) {
            labels[(long)b * L + pos] = v;
            ys_in[(long)b * (L + 1) + pos + 1] = v;
            ys_out[(long)b * (L + 1) + pos] = v;
        }
        __syncthreads();
        if (tid == 0) s_run += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    const int n = s_run;
    for (int l = n + tid; l < L; l += 256) {
        labels[(long)b * L + l] = -1;
        ys_in[(long)b * (L + 1) + l + 1] = eos;
        ys_out[(long)b * (L + 1) + l + 1] = ignore_id;
    }
    if (tid == 0) {
        ys_in[(long)b * (L + 1)] = eos;          // sos == eos (e2e_asr_transformer.py:111-112)
        ys_out[(long)b * (L + 1) + n] = eos;
        if (s_bad) __hip_atomic_store(&g_lrs_target_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// decoder input embedding: x[b,l,:] = emb[tok[b,l]] * scale + pe[l]      (bf16 out);  backward scatter-adds into demb
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_embed_pos_fwd(const long* __restrict__ tok, const float* __restrict__ emb, const float* __restrict__ pe,
                                                       bf16_t* __restrict__ x, int R, int L, int D, float scale) {
    const int dv = D >> 3;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < (long)R * dv; idx += (long)gridDim.x * 256) {
        const int r = (int)(idx / dv), c0 = (int)(idx - (long)r * dv) * 8;
        const long tk = tok[r];
        const float* e = emb + tk * D + c0;
        const float* p = pe + (long)(r % L) * D + c0;
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = e[k] * scale + p[k];
        *reinterpret_cast<u32x4*>(x + (long)r * D + c0) = pack8(o);
    }
}

// Rows of a repeated token are added in increasing row order by the workgroup of its FIRST occurrence (no floating-point atomics).
// One workgroup per row: its threads look for an earlier occurrence together, mark the later ones in an LDS bit mask and then walk the
// set bits in increasing order, one column per thread.  (The first version let every (row, column) thread scan all rows by itself:
// R dependent loads per element, 211 us for 1,600 x 768.)
#define EMB_MAX_ROWS 16384
#define EMB_LIST 4096
__global__ __launch_bounds__(256) void k_embed_pos_bwd(const long* __restrict__ tok, const bf16_t* __restrict__ dx, float* __restrict__ demb,
                                                       int R, int D, float scale) {
    __shared__ unsigned s_bits[EMB_MAX_ROWS / 32];
    __shared__ int s_dup;
    const int r = blockIdx.x, tid = threadIdx.x;
    const long tk = tok[r];
    const int nwords = (R + 31) >> 5;
    if (tid == 0) s_dup = 0;
    for (int w = tid; w < nwords; w += 256) s_bits[w] = 0u;
    __syncthreads();
    for (int r2 = tid; r2 < R; r2 += 256) {
        if (tok[r2] != tk) continue;
        if (r2 < r) s_dup = 1;                                   // (every writer writes the same value)
        else atomicOr(&s_bits[r2 >> 5], 1u << (r2 & 31));
    }
    __syncthreads();
    if (s_dup) return;
    // the rows of this token as a list (increasing): a frequent token — the eos padding of the decoder inputs is a third of all rows — made the
    // bit walk below a chain of ~200 dependent loads per thread (285 us for 650 x 768); from the list a thread requests sixteen rows at a time
    // and still adds them in increasing row order
    __shared__ int s_list[EMB_LIST];
    __shared__ int s_n;
    if (tid == 0) {
        int n = 0;
        for (int w = r >> 5; w < nwords && n <= EMB_LIST - 32; ++w) {
            unsigned m = s_bits[w];
            while (m) { s_list[n++] = w * 32 + __builtin_ctz(m); m &= m - 1; }
        }
        int rest = 0;
        for (int w = r >> 5; w < nwords; ++w) rest += __builtin_popcount(s_bits[w]);
        s_n = rest == n ? n : -1;                                 // -1: more rows than the list holds -> the bit walk
    }
    __syncthreads();
    const int n = s_n;
    for (int c = tid; c < D; c += 256) {
        float acc = 0.f;
        if (n >= 0) {
            int i = 0;
            for (; i + 16 <= n; i += 16) {
                float v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = bf2f(dx[(long)s_list[i + k] * D + c]);
#pragma unroll
                for (int k = 0; k < 16; ++k) acc += v[k] * scale;
            }
            for (; i < n; ++i) acc += bf2f(dx[(long)s_list[i] * D + c]) * scale;
        } else {
            for (int w = r >> 5; w < nwords; ++w) {
                unsigned m = s_bits[w];
                while (m) {
                    const int b = __builtin_ctz(m);
                    m &= m - 1;
                    acc += bf2f(dx[(long)(w * 32 + b) * D + c]) * scale;
                }
            }
        }
        demb[tk * D + c] += acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ESPnet label smoothing: true = conf at the target, smoothing/(V-1) elsewhere; loss = sum_rows KL(true || softmax) / denom
// over rows whose target != ignore; counts[0] += correct argmax, counts[1] += live rows.  One wave per row.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ls_loss_fwd(const float* __restrict__ z, int ld, const long* __restrict__ target, int R, int V,
                                                     float smoothing, float inv_denom, float* __restrict__ rows3, float* __restrict__ lse) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float conf = 1.f - smoothing, low = smoothing / (float)(V - 1);
    const float ent = (conf > 0.f ? conf * __logf(conf) : 0.f) + (low > 0.f ? (float)(V - 1) * low * __logf(low) : 0.f);
    for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
        const long t = target[row];
        const float* p = z + (long)row * ld;
        float m = -INFINITY;
        int am = 0;
        for (int v = lane; v < V; v += 64) { const float x = p[v]; if (x > m) { m = x; am = v; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(m, o, 64); const int oa = __shfl_xor(am, o, 64);
            if (om > m || (om == m && oa < am)) { m = om; am = oa; }
        }
        float se = 0.f, sz = 0.f;
        for (int v = lane; v < V; v += 64) { se += __expf(p[v] - m); sz += p[v]; }
        se = wave_sum(se); sz = wave_sum(sz);
        const float l = m + __logf(se);
        if (lane == 0) {
            lse[row] = l;
            float r0 = 0.f, r1 = 0.f, r2 = 0.f;       // (loss, correct, live) of this row; the totals are their fixed-order column sums
            if (t >= 0) {
                const float zt = t < V ? p[t] : __uint_as_float(0x7fc00000u);      // out-of-range target: poison, do not read out of bounds
                // -sum true_v log p_v = conf (lse - z_t) + low ((V-1) lse - (sum z - z_t))
                const float nl = conf * (l - zt) + low * ((float)(V - 1) * l - (sz - zt));
                r0 = (ent + nl) * inv_denom; r1 = am == (int)t ? 1.f : 0.f; r2 = 1.f;
            }
            rows3[row * 3 + 0] = r0; rows3[row * 3 + 1] = r1; rows3[row * 3 + 2] = r2;
        }
    }
}

__global__ __launch_bounds__(256) void k_ls_loss_bwd(const float* __restrict__ z, int ld, const long* __restrict__ target, int R, int V,
                                                     float smoothing, float inv_denom, const float* __restrict__ lse, const float* __restrict__ gout,
                                                     bf16_t* __restrict__ dz, int ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float conf = 1.f - smoothing, low = smoothing / (float)(V - 1);
    const float g = gout[0] * inv_denom;
    for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
        const long t = target[row];
        const float* p = z + (long)row * ld;
        bf16_t* o = dz + (long)row * ldo;
        const float l = lse[row];
        for (int v = lane; v < ldo; v += 64) {
            float d = 0.f;
            if (t >= 0 && v < V) d = g * (__expf(p[v] - l) - (v == t ? conf : low));
            o[v] = f2bf(d);
        }
    }
}

// y = alpha * dropout(x)   (dropout optional; element index = position in the contiguous tensor) ; y may alias x
__global__ __launch_bounds__(256) void k_scale_bf16(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long nvec, float alpha, DropArgs drop) {
    const bool on = drop.seed != nullptr;
    const unsigned key = on ? drop_key(drop) : 0u;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        float f[8];
        unpack8(reinterpret_cast<const u32x4*>(x)[i], f);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (on) f[k] = drop_keep(key, drop.thresh, (unsigned)(i * 8 + k)) ? f[k] * drop.scale : 0.f;
            f[k] *= alpha;
        }
        reinterpret_cast<u32x4*>(y)[i] = pack8(f);
    }
}

static inline int grid1d(long n, int cap = 2048) { long b = (n + 255) / 256; if (b > cap) b = cap; if (b < 1) b = 1; return (int)b; }

// ---------------------------------------------------------------------------------------------------------------------
// CTC prefix scores for beam search (Watanabe et al., "Hybrid CTC/attention architecture for end-to-end speech recognition",
// Algorithm 2; the reference vectorises it in torch, espnet/nets/ctc_prefix_score.py:11-165, one launch of ~10 small kernels per
// frame).  Here one thread owns one (hypothesis, candidate label) pair and walks the T frames of the recursion in registers:
//   r_n[t] = logaddexp(r_n[t-1], phi[t-1]) + logp[t][c]         phi[t] = r_b_prev[t] if c == last label else logaddexp(r_n_prev[t], r_b_prev[t])
//   r_b[t] = logaddexp(r_n[t-1], r_b[t-1]) + logp[t][blank]
//   psi    = logsumexp(r_n[start-1], phi[t-1] + logp[t][c] for t in [start, T)),   start = max(#labels in the prefix, 1)
// eos gets logaddexp(r_n_prev[T-1], r_b_prev[T-1]), blank gets LOGZERO.  r_new [n][S][T][2] is the state of each extension.
// ---------------------------------------------------------------------------------------------------------------------
#define CTC_LOGZERO (-1.0e10f)
__device__ __forceinline__ float logaddexpf_(float a, float b) {
    const float m = fmaxf(a, b);
    return m + __logf(__expf(a - m) + __expf(b - m));
}

__global__ __launch_bounds__(256) void k_ctc_prefix_score(const float* __restrict__ logp, const float* __restrict__ r_prev, const long* __restrict__ last,
                                                          const long* __restrict__ ids, float* __restrict__ r_new, float* __restrict__ psi, int T,
                                                          int V, int ldp, int n, int S, int out_len, int blank, int eos) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * S) return;
    const int h = idx / S, j = idx - h * S;
    const int c = ids != nullptr ? (int)ids[(long)h * S + j] : j;
    float* rn_out = r_new + (long)idx * T * 2;
    const float* rp = r_prev + (long)h * T * 2;
    const bool same = c == (int)last[h];
    const int start = out_len > 1 ? out_len : 1;
    for (int t = 0; t < start - 1 && t < T; ++t) { rn_out[2 * t] = CTC_LOGZERO; rn_out[2 * t + 1] = CTC_LOGZERO; }
    float rn = (out_len == 0) ? logp[c] : CTC_LOGZERO, rb = CTC_LOGZERO;          // r[start-1]
    if (start - 1 < T) { rn_out[2 * (start - 1)] = rn; rn_out[2 * (start - 1) + 1] = rb; }
    float acc = rn;
    for (int t = start; t < T; ++t) {
        const float pn = rp[2 * (t - 1)], pb = rp[2 * (t - 1) + 1];
        const float phi = same ? pb : logaddexpf_(pn, pb);
        const float x = logp[(long)t * ldp + c], xb = logp[(long)t * ldp + blank];
        acc = logaddexpf_(acc, phi + x);
        const float nn = logaddexpf_(rn, phi) + x;
        const float nb = logaddexpf_(rn, rb) + xb;
        rn = nn; rb = nb;
        rn_out[2 * t] = rn; rn_out[2 * t + 1] = rb;
    }
    if (c == eos) acc = logaddexpf_(rp[2 * (T - 1)], rp[2 * (T - 1) + 1]);
    if (c == blank) acc = CTC_LOGZERO;
    psi[idx] = acc;
}

extern "C" {

int svsr_colsum_rows(const float* ws, int nrows, int64_t ld, float* out0, int64_t n0, float* out1, int64_t n1, int accumulate, float scale,
                     hipStream_t stream);

/* rows of [2][D] BatchNorm1d partials svsr_glu_dwconv_fwd writes (one per clip and DW_TT-frame tile) */
int svsr_glu_dwconv_fwd_stat_rows(int B, int T) { return (B < 1 || T < 1) ? 0 : B * ((T + DW_TT - 1) / DW_TT); }

int svsr_glu_dwconv_fwd(const void* u, const float* w, const float* bias, void* c, float* stats, int B, int T, int D, int K, hipStream_t stream) {
    if (D % 64 != 0 || K < 1 || K > DW_MAXK || (K & 1) == 0) return SVSR_ERR_ARG;
    if (K == DW_MAXK)
        hipLaunchKernelGGL(k_glu_dwconv_fwd<DW_MAXK>, dim3((T + DW_TT - 1) / DW_TT, D / 64, B), dim3(256), 0, stream, (const bf16_t*)u, w, bias, (bf16_t*)c,
                           stats, B, T, D, K);
    else
        hipLaunchKernelGGL(k_glu_dwconv_fwd<0>, dim3((T + DW_TT - 1) / DW_TT, D / 64, B), dim3(256), 0, stream, (const bf16_t*)u, w, bias, (bf16_t*)c,
                           stats, B, T, D, K);
    return svsr_check_launch();
}

/* part: fp32 workspace of nsplit * D * (K+1) floats (any contents) */
/* parts: bit 0 = the pass over the activations (du, partial rows of dw / dbias into part), bit 1 = the fixed-order sum of the partial rows into
 * dw / dbias — parameter gradients nothing in the backward chain waits for: the caller may issue bit 1 on another stream behind bit 0 */
int svsr_glu_dwconv_bwd_parts(const void* dc, const void* u, const float* w, void* du, float* dw, float* dbias, float* part, int nsplit,
                              int B, int T, int D, int K, int parts, hipStream_t stream) {
    if (D % 64 != 0 || K < 1 || K > DW_MAXK || (K & 1) == 0 || nsplit < 1 || (parts & 3) == 0 || (parts & ~3) != 0) return SVSR_ERR_ARG;
    const int ntt = (T + DW_TT - 1) / DW_TT;
    if (nsplit > B * ntt) nsplit = B * ntt;
    if (parts & 1) {
        if (K == DW_MAXK)
            hipLaunchKernelGGL(k_glu_dwconv_bwd<DW_MAXK>, dim3(D / 64, nsplit), dim3(256), 0, stream, (const bf16_t*)dc, (const bf16_t*)u, w, (bf16_t*)du, part,
                               B, T, D, K, ntt);
        else
            hipLaunchKernelGGL(k_glu_dwconv_bwd<0>, dim3(D / 64, nsplit), dim3(256), 0, stream, (const bf16_t*)dc, (const bf16_t*)u, w, (bf16_t*)du, part,
                               B, T, D, K, ntt);
    }
    if (parts & 2) hipLaunchKernelGGL(k_dw_reduce, dim3(grid1d((long)D * (K + 1), 256)), dim3(256), 0, stream, part, nsplit, D, K, dw, dbias);
    return svsr_check_launch();
}

int svsr_glu_dwconv_bwd(const void* dc, const void* u, const float* w, void* du, float* dw, float* dbias, float* part, int nsplit,
                        int B, int T, int D, int K, hipStream_t stream) {
    return svsr_glu_dwconv_bwd_parts(dc, u, w, du, dw, dbias, part, nsplit, B, T, D, K, 3, stream);
}

/* logits fp32 [B*T][ld]; labels int64 [B][Lmax] padded with -1; ilen int32 [B]; ab workspace fp32 [B][T][2*Lmax+1];
 * lse [B*T], nll [B] scratch; *loss = sum_b nll_b / B (fixed order).  Then svsr_ctc_grad writes dlogits (bf16, pitch ldo). */
int svsr_ctc_fwd(const float* logits, int ld, const int64_t* labels, int Lmax, const int* ilen, int B, int T, int V, float* lse,
                 float* ab, float* nll, float* loss, hipStream_t stream) {
    if (Lmax < 1 || T < 1 || V < 2) return SVSR_ERR_ARG;
    const int Smax = 2 * Lmax + 1;
    hipLaunchKernelGGL(k_row_lse, dim3(grid1d((long)B * T * 64)), dim3(256), 0, stream, logits, ld, B * T, V, lse);
    hipLaunchKernelGGL(k_ctc_lattice, dim3(B), dim3(256), (size_t)3 * Smax * sizeof(float), stream, logits, ld, lse, (const long*)labels, Lmax, ilen,
                       B, T, Smax, ab, nll);
    const int rc = svsr_check_launch();
    if (rc != SVSR_OK) return rc;
    return svsr_colsum_rows(nll, B, 1, loss, 1, nullptr, 0, 0, 1.0f / (float)B, stream);
}

int svsr_ctc_grad(const float* logits, int ld, const int64_t* labels, int Lmax, const int* ilen, int B, int T, int V, const float* lse,
                  const float* ab, const float* nll, const float* gout, void* dlogits, int ldo, hipStream_t stream) {
    const size_t lds = ((size_t)V + 2 * (2 * Lmax + 1)) * sizeof(float);
    if (Lmax < 1 || lds > 60 * 1024) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_ctc_grad, dim3(B * T), dim3(256), lds, stream, logits, ld, lse, (const long*)labels, Lmax, ilen, ab,
                       nll, B, T, V, 2 * Lmax + 1, gout, (bf16_t*)dlogits, ldo);
    return svsr_check_launch();
}

int svsr_lrs_targets(const int64_t* label, int B, int L, int odim, int64_t ignore_id, int64_t eos, int64_t* labels, int64_t* ys_in,
                     int64_t* ys_out, hipStream_t stream) {
    if (B < 1 || L < 1 || odim < 2 || label == nullptr || labels == nullptr || ys_in == nullptr || ys_out == nullptr) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_lrs_targets, dim3(B), dim3(256), 0, stream, (const long*)label, L, odim, (long)ignore_id, (long)eos, (long*)labels,
                       (long*)ys_in, (long*)ys_out);
    return svsr_check_launch();
}

/* 1 if a svsr_lrs_targets launch since the last reset met a label outside [1, odim) (it was replaced by eos), else 0; -1 if the word cannot be
 * read.  Synchronises the device. */
int svsr_lrs_target_errors(int reset) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_lrs_target_err), sizeof v) != hipSuccess) return -1;
    if (reset && v != 0) { const unsigned z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lrs_target_err), &z, sizeof z); }
    return v != 0 ? 1 : 0;
}

int svsr_embed_pos_fwd(const int64_t* tok, const float* emb, const float* pe, void* x, int R, int L, int D, float scale, hipStream_t stream) {
    if (D % 8 != 0) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_embed_pos_fwd, dim3(grid1d((long)R * (D / 8))), dim3(256), 0, stream, (const long*)tok, emb, pe, (bf16_t*)x, R, L, D, scale);
    return svsr_check_launch();
}

int svsr_embed_pos_bwd(const int64_t* tok, const void* dx, float* demb, int R, int D, float scale, hipStream_t stream) {
    if (R < 1 || R > EMB_MAX_ROWS || D < 1) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_embed_pos_bwd, dim3(R), dim3(256), 0, stream, (const long*)tok, (const bf16_t*)dx, demb, R, D, scale);
    return svsr_check_launch();
}

int svsr_ls_loss_fwd(const float* logits, int ld, const int64_t* target, int R, int V, float smoothing, float inv_denom, float* loss,
                     float* lse, float* counts, float* rows3, hipStream_t stream) {
    if (V < 2 || R < 1 || rows3 == nullptr) return SVSR_ERR_ARG;          // rows3: [R][3] floats
    hipLaunchKernelGGL(k_ls_loss_fwd, dim3(grid1d((long)R * 64)), dim3(256), 0, stream, logits, ld, (const long*)target, R, V, smoothing, inv_denom,
                       rows3, lse);
    const int rc = svsr_check_launch();
    if (rc != SVSR_OK) return rc;
    return svsr_colsum_rows(rows3, R, 3, loss, 1, counts, 2, 0, 1.0f, stream);
}

int svsr_ls_loss_bwd(const float* logits, int ld, const int64_t* target, int R, int V, float smoothing, float inv_denom, const float* lse,
                     const float* gout, void* dlogits, int ldo, hipStream_t stream) {
    hipLaunchKernelGGL(k_ls_loss_bwd, dim3(grid1d((long)R * 64)), dim3(256), 0, stream, logits, ld, (const long*)target, R, V, smoothing, inv_denom,
                       lse, gout, (bf16_t*)dlogits, ldo);
    return svsr_check_launch();
}

int svsr_scale_bf16(const void* x, void* y, int64_t n, float alpha, const unsigned* drop_seed, unsigned drop_site, float drop_p,
                    hipStream_t stream) {
    if (n % 8 != 0) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_scale_bf16, dim3(grid1d(n / 8)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, (long)(n / 8), alpha,
                       svsr_make_drop(drop_seed, drop_site, drop_p));
    return svsr_check_launch();
}

int svsr_ctc_prefix_score(const float* logp, int ldp, const float* r_prev, const int64_t* last, const int64_t* ids, float* r_new, float* psi, int T,
                          int V, int n, int S, int out_len, int blank, int eos, hipStream_t stream) {
    if (T < 1 || V < 2 || n < 1 || S < 1 || (ids == nullptr && S != V) || ldp < V || out_len < 0 || blank < 0 || blank >= V || eos < 0 || eos >= V)
        return SVSR_ERR_ARG;
    // one thread per (hypothesis, candidate), no grid-stride loop in the kernel: the grid covers every pair (full-vocabulary scoring at a
    // wide beam exceeds the 2048-block cap of grid1d)
    if (((long)n * S + 255) / 256 > 0x7fffffffL) return SVSR_ERR_ARG;
    hipLaunchKernelGGL(k_ctc_prefix_score, dim3((unsigned)(((long)n * S + 255) / 256)), dim3(256), 0, stream, logp, r_prev, (const long*)last, (const long*)ids, r_new,
                       psi, T, V, ldp, n, S, out_len, blank, eos);
    return svsr_check_launch();
}

}  // extern "C"
