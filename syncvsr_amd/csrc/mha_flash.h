// Attention for the sentence-level model without the probability matrix in HBM on the way forward (included from mha.hip).
//
// Same function as k_mha_fwd4 / k_mha_bwd_q4 (reference LRS/video/espnet/nets/pytorch_backend/transformer/attention.py:38-108 plain and
// :191-278 relative-position attention, rel_shift :216-236, mask semantics :71-78), restructured around what bounded those kernels
// (LRS kernels): they ran one 32-query tile per workgroup as a serial chain global load -> LDS -> barrier -> ONE MFMA
// per 32 keys, gathered every k-major fragment with eight 2-byte LDS reads, and wrote / re-read P.  Here
//   * a workgroup owns up to eight query tiles of one (clip, head) — one per wave — and streams the keys in blocks of 32: K, V of a
//     block are staged ONCE per workgroup (V transposed on the way in), double-buffered, one barrier per block;
//   * everything is computed TRANSPOSED, S^T = K (Q+u)^T, so a lane owns one query: softmax statistics are lane-local (one exchange with
//     lane ^ 32 per block), the online-softmax rescale is a per-lane scalar, and P^T leaves the accumulators as the B operand of
//     ctx^T = V^T P^T after one bf16 pack and a v_permlane32_swap — no LDS round trip, no P in memory;
//   * the relative-position term (q+v) pe[Lq-1+j-i] is one MFMA block per key block (the 64-row window of the position table a query
//     tile needs moves by 32 rows per key block: one new block per step) turned through a wave-private fp32 LDS patch for the skew;
//   * the forward keeps only the row's log-sum-exp; the backward's query pass recomputes P from it.
// Backward, query pass (k_mhaf_bwd_q): recomputes S^T and P^T, dP^T = V dctx^T, dS = P o (dP' - D) * scale with D_i = dctx_i . ctx_i
// (the flash-attention identity: sum_j P'_ij dP_ij = dctx_i . (P' V)_i), dq = dS K (+ dS_skewed PE from a pre-transposed position table,
// aligned 16-byte loads), and writes P and dS (bf16) for the key pass (k_mha_bwd_kv4) and the position-table pass (k_mha_bwd_pe4) of
// mha_coop.h, which are unchanged.
#pragma once

#define MF_VP 40           // pitch (bf16) of transposed [64 d][32 j] LDS tiles: 80-byte rows, 16-byte aligned fragments
#define MF_BDP 66          // pitch (floats) of the wave-private skew patch [32 queries][64 table rows]
#define MF_SP 40           // pitch (bf16) of the wave-private dS strip [32 queries][32 keys]
#define MF_SWZ(row, chunk) ((row) * 64 + (((chunk) ^ (((row) >> 1) & 7)) << 3))      // [32][64] bf16 tile, 16-byte chunks XOR-swizzled

struct MhafArgs {
    MhaArgs m;                 // the tensors and sizes of the unfused kernels (probs / ds: written by the backward's query pass)
    float* lse;                // [B*H][Lq] log-sum-exp of the scaled, masked scores of a query (+inf: every key masked)
    const bf16_t* ctx; int ctx_in_pitch;     // backward: the forward's output
    const bf16_t* pet; int pet_pitch, pet_lm;   // backward, rel-pos: transposed position table [H*64][pet_pitch], column = row index + pet_lm
};

__device__ __forceinline__ int mf_row(int e, int half) { return (e & 3) + 8 * (e >> 2) + 4 * half; }

// 16 fp32 of the S^T layout (lane: query il, keys mf_row(e, half)) -> the two B fragments (keys 0..15, 16..31) of the next contraction
__device__ __forceinline__ void mf_pack_swap(const float (&p)[16], bf16x8 (&frag)[2]) {
    unsigned pk[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) { pk[g][0] = pack2bf(p[4 * g + 0], p[4 * g + 1]); pk[g][1] = pack2bf(p[4 * g + 2], p[4 * g + 3]); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        // lanes 0-31 keep their group 2ks (keys 16ks + 0..3) and receive the upper half's group 2ks (keys 4..7); lanes 32-63 receive the lower
        // half's group 2ks+1 (keys 8..11) and keep their own (keys 12..15)
        const auto r0 = __builtin_amdgcn_permlane32_swap((int)pk[2 * ks][0], (int)pk[2 * ks + 1][0], false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap((int)pk[2 * ks][1], (int)pk[2 * ks + 1][1], false, false);
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
        const u4 v = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
        frag[ks] = __builtin_bit_cast(bf16x8, v);
    }
}

// stage key block kb of (b, h): K rows (and V rows) -> LDS.  Threads 0..255: row tid >> 3, 16-byte chunk tid & 7.
struct MfStage { u32x4 k, v; };
__device__ __forceinline__ void mf_load_kv(const MhaArgs& a, int b, int h, int kb, int tid, MfStage& s) {
    s.k = u32x4{0u, 0u, 0u, 0u}; s.v = s.k;
    const int r = tid >> 3, c8 = tid & 7, j = kb * 32 + r;
    if (tid < 256 && j < a.Lk) {
        const long o = ((long)b * a.Lk + j) * a.kv_pitch + h * MHA_DH + c8 * 8;
        s.k = *reinterpret_cast<const u32x4*>(a.k + o);
        s.v = *reinterpret_cast<const u32x4*>(a.v + o);
    }
}
__device__ __forceinline__ void mf_store_rows(bf16_t* dst, const u32x4& v, int tid) {          // [32][64] swizzled
    if (tid < 256) *reinterpret_cast<u32x4*>(dst + MF_SWZ(tid >> 3, tid & 7)) = v;
}
__device__ __forceinline__ void mf_store_transposed(bf16_t* dst, const u32x4& v, int tid) {     // [64][MF_VP]: element (d, j) at d * MF_VP + j
    if (tid < 256) {
        const int r = tid >> 3, c8 = tid & 7;
        const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dst[(c8 * 8 + 2 * e) * MF_VP + r] = (bf16_t)(wv[e] & 0xffffu);
            dst[(c8 * 8 + 2 * e + 1) * MF_VP + r] = (bf16_t)(wv[e] >> 16);
        }
    }
}

// one 32-row block of the position table against (q + v)^T -> the wave's skew patch, columns colbase + table row (mod 64).  The four
// fragments of a block are requested one key block ahead (mf_pe_load).
__device__ __forceinline__ void mf_pe_load(const MhaArgs& a, int h, int rfirst, int il, int half, bf16x8 (&f)[4]) {
    int pr = rfirst + il;
    pr = pr < 0 ? 0 : (pr > 2 * a.Lq - 2 ? 2 * a.Lq - 2 : pr);
    const bf16_t* prow = a.pe + (long)pr * a.pe_pitch + h * MHA_DH + half * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f[kk] = ld_frag(prow + kk * 16);
}
__device__ __forceinline__ void mf_bd_block(const bf16x8 (&pef)[4], const bf16x8 (&qv)[4], float* myBD, int colbase, int il, int half) {
    f32x16 accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pef[kk], qv[kk], accb, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) myBD[il * MF_BDP + ((colbase + mf_row(e, half)) & 63)] = accb[e];
}

// scaled, masked scores of key block kb for this wave's query tile, S^T layout; REL: the skew patch holds table block "A" of this key
// block in ring half (kb & 1); block "B" is computed here into the other half (it is block A of the next key block)
template <bool REL>
__device__ __forceinline__ void mf_scores(const MhaArgs& a, const bf16_t* sKb, const bf16x8 (&qu)[4], const bf16x8 (&qv)[4], const bf16x8 (&pef)[4], float* myBD,
                                          int kb, int i0, int klen, int il, int half, float (&s)[16]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(sKb + MF_SWZ(il, kk * 2 + half)), qu[kk], acc, 0, 0, 0);
    const int j0 = kb * 32;
    if (REL) {
        mf_bd_block(pef, qv, myBD, ((kb + 1) & 1) * 32, il, half);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the patch is private to this wave: its LDS operations complete in order
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] += myBD[il * MF_BDP + (((kb & 1) * 32 + 31 + mf_row(e, half) - il) & 63)];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (j0 + 32 <= klen && i0 + 32 <= a.Lq && (!a.causal || j0 + 31 <= i0)) {        // (wave-uniform) nothing of this block is masked
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = acc[e] * a.scale;
        return;
    }
    const int i = i0 + il;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int j = j0 + mf_row(e, half);
        const bool ok = j < klen && i < a.Lq && (!a.causal || j <= i);
        s[e] = ok ? acc[e] * a.scale : -INFINITY;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// forward: grid (query-tile groups, B*H), blockDim = 64 x waves (4..8); dynamic LDS = mhaf_fwd_lds(waves)
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool REL>
__global__ __launch_bounds__(512) void k_mhaf_fwd(const MhafArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const MhaArgs& a = p.m;
    const int W = (int)blockDim.x >> 6;
    bf16_t* sK = reinterpret_cast<bf16_t*>(smem_raw);                    // [2][32 * 64]
    bf16_t* sVt = sK + 2 * 2048;                                          // [2][64 * MF_VP]
    float* sBD = reinterpret_cast<float*>(sVt + 2 * 64 * MF_VP);          // [W][32 * MF_BDP]
    const int tid = threadIdx.x, lane = tid & 63, il = lane & 31, half = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int nq = (a.Lq + 31) >> 5, qt = blockIdx.x * W + w;
    const bool active = qt < nq;
    const int i0 = qt * 32;
    const int klen = a.klen != nullptr ? min(a.klen[b], a.Lk) : a.Lk;
    int kend_wg = klen, kend = klen;
    if (a.causal) { kend_wg = min(kend_wg, min(a.Lq, (int)(blockIdx.x + 1) * W * 32)); kend = min(kend, i0 + 32); }
    const int nkb_wg = (kend_wg + 31) >> 5, nkb = active ? (kend + 31) >> 5 : 0;
    float* myBD = sBD + w * 32 * MF_BDP;

    bf16x8 qu[4], qv[4];
    {
        const int qi = min(active ? i0 + il : 0, a.Lq - 1);
        const bf16_t* qrow = a.q + ((long)b * a.Lq + qi) * a.q_pitch + h * MHA_DH + half * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 f = ld_frag(qrow + kk * 16);
            if (REL) {
                qu[kk] = add_bias_frag(f, a.bias_u + h * MHA_DH + kk * 16 + half * 8);
                qv[kk] = add_bias_frag(f, a.bias_v + h * MHA_DH + kk * 16 + half * 8);
            } else {
                qu[kk] = f; qv[kk] = f;
            }
        }
    }
    MfStage st;
    if (nkb_wg > 0) {
        mf_load_kv(a, b, h, 0, tid, st);
        mf_store_rows(sK, st.k, tid);
        mf_store_transposed(sVt, st.v, tid);
    }
    bf16x8 pef[4], pen[4];
    if (REL && nkb > 0) {
        mf_pe_load(a, h, (a.Lq - 1) - i0 - 31, il, half, pen);
        mf_pe_load(a, h, (a.Lq - 1) - i0 + 1, il, half, pef);         // block B of key block 0
        mf_bd_block(pen, qv, myBD, 0, il, half);                       // table block A of key block 0
    }
    __syncthreads();

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    const bool drop_on = a.drop.seed != nullptr;
    const unsigned dkey = drop_on ? drop_key(a.drop) : 0u;

    for (int kb = 0; kb < nkb_wg; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb_wg) mf_load_kv(a, b, h, kb + 1, tid, st);
        if (kb < nkb) {
            float s[16];
            if (REL && kb + 1 < nkb) mf_pe_load(a, h, (a.Lq - 1) + (kb + 1) * 32 - i0 + 1, il, half, pen);     // block B of the next key block
            mf_scores<REL>(a, sK + buf * 2048, qu, qv, pef, myBD, kb, i0, klen, il, half, s);
            if (REL) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) pef[kk] = pen[kk];
            }
            float bm = s[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) bm = fmaxf(bm, s[e]);
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
            const float m_new = fmaxf(m_run, bm);
            float alpha = 1.f, pr[16];
            if (m_new == -INFINITY) {
#pragma unroll
                for (int e = 0; e < 16; ++e) pr[e] = 0.f;
            } else {
                alpha = __expf(m_run - m_new);
                float sum = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) { pr[e] = __expf(s[e] - m_new); sum += pr[e]; }
                l_run = l_run * alpha + sum;
                m_run = m_new;
            }
            if (!__all(alpha == 1.f)) {
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
            }
            if (drop_on) {
                const long rowbase = ((long)bh * a.Lq + min(i0 + il, a.Lq - 1)) * a.ldp + kb * 32;
#pragma unroll
                for (int e = 0; e < 16; ++e) pr[e] = drop_keep(dkey, a.drop.thresh, (unsigned)(rowbase + mf_row(e, half))) ? pr[e] * a.drop.scale : 0.f;
            }
            bf16x8 pf[2];
            mf_pack_swap(pr, pf);
            const bf16_t* vt = sVt + buf * 64 * MF_VP;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(vt + (db * 32 + il) * MF_VP + ks * 16 + half * 8), pf[ks], o[db], 0, 0, 0);
        }
        if (kb + 1 < nkb_wg) {
            mf_store_rows(sK + (buf ^ 1) * 2048, st.k, tid);
            mf_store_transposed(sVt + (buf ^ 1) * 64 * MF_VP, st.v, tid);
        }
        __syncthreads();
    }
    if (!active) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    const int i = i0 + il;
    if (i < a.Lq) {
        if (half == 0) p.lse[(long)bh * a.Lq + i] = l_tot > 0.f ? m_run + __logf(l_tot) : INFINITY;
        bf16_t* crow = a.ctx + ((long)b * a.Lq + i) * a.ctx_pitch + h * MHA_DH;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = o[db][e] * inv;
            bf16x8 f[2];
            mf_pack_swap(v, f);            // rows of o^T are head columns: lane < 32 now holds columns 16m .. +7, lane >= 32 columns 16m + 8 .. +15
#pragma unroll
            for (int m = 0; m < 2; ++m) *reinterpret_cast<bf16x8*>(crow + db * 32 + m * 16 + half * 8) = f[m];
        }
    }       // (the two lanes of a swap pair own the same query: they take this branch together)
}

static inline size_t mhaf_fwd_lds(int waves) { return (size_t)(2 * 2048 + 2 * 64 * MF_VP) * 2 + (size_t)waves * 32 * MF_BDP * 4; }

// ---------------------------------------------------------------------------------------------------------------------------------
// position table transposed: pet[c][LM + r] = pe[r][c] for 0 <= r < 2 Lq - 1, zeros elsewhere; grid (ceil(Rp / 32), D / 32), 256 threads
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mhaf_pe_transpose(const bf16_t* __restrict__ pe, int pe_pitch, int nrows, bf16_t* __restrict__ pet, int Rp, int LM) {
    __shared__ bf16_t tile[32][34];
    const int c0 = blockIdx.x * 32, d0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = c0 + ty + 8 * k - LM;
        tile[ty + 8 * k][tx] = (r >= 0 && r < nrows) ? pe[(long)r * pe_pitch + d0 + tx] : (bf16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + tx;
        if (c < Rp) pet[(long)(d0 + ty + 8 * k) * Rp + c] = tile[tx][ty + 8 * k];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward, query pass: grid (query-tile groups, B*H), blockDim = 64 x waves; dynamic LDS = mhaf_bwd_lds(waves)
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool REL>
__global__ __launch_bounds__(512) void k_mhaf_bwd_q(const MhafArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const MhaArgs& a = p.m;
    const int W = (int)blockDim.x >> 6;
    bf16_t* sK = reinterpret_cast<bf16_t*>(smem_raw);                    // [2][32 * 64]
    bf16_t* sV = sK + 2 * 2048;                                           // [2][32 * 64]
    bf16_t* sKt = sV + 2 * 2048;                                          // [2][64 * MF_VP]
    float* sBD = reinterpret_cast<float*>(sKt + 2 * 64 * MF_VP);          // [W][32 * MF_BDP]
    bf16_t* sDS = reinterpret_cast<bf16_t*>(sBD + W * 32 * MF_BDP);       // [W][32 * MF_SP]
    const int tid = threadIdx.x, lane = tid & 63, il = lane & 31, half = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int nq = (a.Lq + 31) >> 5, qt = blockIdx.x * W + w;
    const bool active = qt < nq;
    const int i0 = qt * 32;
    const int klen = a.klen != nullptr ? min(a.klen[b], a.Lk) : a.Lk;
    int kend_wg = klen, kend = klen;
    if (a.causal) { kend_wg = min(kend_wg, min(a.Lq, (int)(blockIdx.x + 1) * W * 32)); kend = min(kend, i0 + 32); }
    const int nkb_wg = (kend_wg + 31) >> 5, nkb = active ? (kend + 31) >> 5 : 0;
    const int nkb_all = (a.Lk + 31) >> 5;
    float* myBD = sBD + w * 32 * MF_BDP;
    bf16_t* myDS = sDS + w * 32 * MF_SP;
    const int i = i0 + il, qi = min(active ? i : 0, a.Lq - 1);
    const bool live = active && i < a.Lq;

    bf16x8 qu[4], qv[4], dct[4];
    float dsum = 0.f, lse = INFINITY;
    {
        const bf16_t* qrow = a.q + ((long)b * a.Lq + qi) * a.q_pitch + h * MHA_DH + half * 8;
        const bf16_t* drow = a.dctx + ((long)b * a.Lq + qi) * a.dctx_pitch + h * MHA_DH + half * 8;
        const bf16_t* crow = p.ctx + ((long)b * a.Lq + qi) * p.ctx_in_pitch + h * MHA_DH + half * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 f = ld_frag(qrow + kk * 16);
            if (REL) {
                qu[kk] = add_bias_frag(f, a.bias_u + h * MHA_DH + kk * 16 + half * 8);
                qv[kk] = add_bias_frag(f, a.bias_v + h * MHA_DH + kk * 16 + half * 8);
            } else {
                qu[kk] = f; qv[kk] = f;
            }
            dct[kk] = ld_frag(drow + kk * 16);
            const bf16x8 c = ld_frag(crow + kk * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum = __builtin_fmaf(bf2f((bf16_t)dct[kk][e]), bf2f((bf16_t)c[e]), dsum);
        }
        dsum += __shfl_xor(dsum, 32, 64);
        if (live) lse = p.lse[(long)bh * a.Lq + i];
    }
    MfStage st;
    if (nkb_wg > 0) {
        mf_load_kv(a, b, h, 0, tid, st);
        mf_store_rows(sK, st.k, tid);
        mf_store_rows(sV, st.v, tid);
        mf_store_transposed(sKt, st.k, tid);
    }
    bf16x8 pef[4], pen[4];
    if (REL && nkb > 0) {
        mf_pe_load(a, h, (a.Lq - 1) - i0 - 31, il, half, pen);
        mf_pe_load(a, h, (a.Lq - 1) - i0 + 1, il, half, pef);
        mf_bd_block(pen, qv, myBD, 0, il, half);
    }
    __syncthreads();

    f32x16 oac[2], obd[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { oac[db][r] = 0.f; obd[db][r] = 0.f; }
    const bool drop_on = a.drop.seed != nullptr;
    const unsigned dkey = drop_on ? drop_key(a.drop) : 0u;
    bf16_t* prow_out = a.probs + ((long)bh * a.Lq + qi) * a.ldp;
    bf16_t* dsrow_out = a.ds + ((long)bh * a.Lq + qi) * a.ldp;

    for (int kb = 0; kb < nkb_all; ++kb) {
        const int buf = kb & 1;
        const bool wg_has = kb < nkb_wg;
        if (wg_has && kb + 1 < nkb_wg) mf_load_kv(a, b, h, kb + 1, tid, st);
        float pv[16], dsv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { pv[e] = 0.f; dsv[e] = 0.f; }
        if (kb < nkb) {
            float s[16];
            if (REL && kb + 1 < nkb) mf_pe_load(a, h, (a.Lq - 1) + (kb + 1) * 32 - i0 + 1, il, half, pen);
            mf_scores<REL>(a, sK + buf * 2048, qu, qv, pef, myBD, kb, i0, klen, il, half, s);
            if (REL) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) pef[kk] = pen[kk];
            }
            f32x16 dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(sV + buf * 2048 + MF_SWZ(il, kk * 2 + half)), dct[kk], dp, 0, 0, 0);
            const long rowbase = ((long)bh * a.Lq + qi) * a.ldp + kb * 32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pe_ = (s[e] == -INFINITY || lse == INFINITY) ? 0.f : __expf(s[e] - lse);
                float d = dp[e];
                if (drop_on) d = drop_keep(dkey, a.drop.thresh, (unsigned)(rowbase + mf_row(e, half))) ? d * a.drop.scale : 0.f;
                pv[e] = pe_;
                dsv[e] = pe_ * (d - dsum) * a.scale;
            }
        }
        // P and dS of this (query tile, key block) for the key / position-table passes: 16-byte stores, zeros where nothing was computed
        bf16x8 pf[2], df[2];
        mf_pack_swap(pv, pf);
        mf_pack_swap(dsv, df);
        if (live) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int j = kb * 32 + m * 16 + half * 8;
                if (j < a.ldp) {
                    *reinterpret_cast<bf16x8*>(prow_out + j) = pf[m];
                    *reinterpret_cast<bf16x8*>(dsrow_out + j) = df[m];
                }
            }
        }
        if (kb < nkb) {
            const bf16_t* kt = sKt + buf * 64 * MF_VP;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    oac[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(kt + (db * 32 + il) * MF_VP + ks * 16 + half * 8), df[ks], oac[db], 0, 0, 0);
            if (REL) {
                // dq_bd^T[d][i] += sum_r pe^T[d][r] dS[i][r - (Lq - 1) + i]: the strip holds this block's dS by (query, key); table rows
                // rb + 16 m + 8 half .. + 7 pair with keys 16 m + 8 half + il - 31 .. + 7 of the block (zero outside it)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 v;
                    v.x = pack2bf(dsv[4 * g + 0], dsv[4 * g + 1]); v.y = pack2bf(dsv[4 * g + 2], dsv[4 * g + 3]);
                    *reinterpret_cast<uint2*>(myDS + il * MF_SP + 8 * g + 4 * half) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // (requesting these eight table fragments a block ahead needs ~270 registers: with four-wave workgroups instead of eight the
                // launch was SLOWER, 56 vs 37 us at 16 x 160 frames — the pass is bound by its instruction count per wave, not by this latency)
                const int rb = (a.Lq - 1) + kb * 32 - i0 - 31;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int c0 = 16 * m + 8 * half + il - 31;
                    bf16x8 fs;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int c = c0 + e;
                        fs[e] = (c >= 0 && c < 32) ? (short)myDS[il * MF_SP + c] : (short)0;
                    }
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const bf16_t* trow = p.pet + (long)(h * MHA_DH + db * 32 + il) * p.pet_pitch + (rb + 16 * m + 8 * half + p.pet_lm);
                        obd[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(trow), fs, obd[db], 0, 0, 0);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        if (wg_has) {
            if (kb + 1 < nkb_wg) {
                mf_store_rows(sK + (buf ^ 1) * 2048, st.k, tid);
                mf_store_rows(sV + (buf ^ 1) * 2048, st.v, tid);
                mf_store_transposed(sKt + (buf ^ 1) * 64 * MF_VP, st.k, tid);
            }
            __syncthreads();
        }
    }
    // dq (and its two summands for the pos_bias_u / pos_bias_v gradients)
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        float vs[16], va[16], vb[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { va[e] = oac[db][e]; vb[e] = obd[db][e]; vs[e] = va[e] + vb[e]; }
        bf16x8 fs[2], fa[2], fb[2];
        mf_pack_swap(vs, fs);
        if (REL) { mf_pack_swap(va, fa); mf_pack_swap(vb, fb); }
        if (live) {
            const long rr = (long)b * a.Lq + i;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int col = h * MHA_DH + db * 32 + m * 16 + half * 8;
                *reinterpret_cast<bf16x8*>(a.dq + rr * a.dq_pitch + col) = fs[m];
                if (REL) {
                    *reinterpret_cast<bf16x8*>(a.dq_ac + rr * a.aux_pitch + col) = fa[m];
                    *reinterpret_cast<bf16x8*>(a.dq_bd + rr * a.aux_pitch + col) = fb[m];
                }
            }
        }
    }
}

static inline size_t mhaf_bwd_lds(int waves) { return (size_t)(4 * 2048 + 2 * 64 * MF_VP) * 2 + (size_t)waves * (32 * MF_BDP * 4 + 32 * MF_SP * 2); }
