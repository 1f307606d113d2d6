// The attention kernels of mha.hip (included from there): four waves cooperate on one 32-row tile.
//
// A first version ran one wave per tile: ~4 waves per CU at LRS sizes (B*H*ceil(T/32) ~ 1,000 tiles), every wave walking a
// serial chain of global loads -> MFMA -> LDS round trips, so a launch took as long as one wave's latency chain (rel-pos
// backward 244 us vs 97 us now).  Here the four waves of a workgroup share one 32-row tile: score / dP blocks are dealt out
// over the waves, the row passes (softmax, dS) take eight rows each, staging of the k-major operand is one 16-byte load per thread, and in the contractions with a
// staged operand wave w takes output half (w & 1) and k-slice (w >> 1) of every staged block; the two k-slice partials are
// summed through LDS at the end.

// stage 32 rows x 64 columns with all 256 threads: one 16-byte load each
__device__ __forceinline__ void stage_rows64_256(bf16_t* dst, const bf16_t* src, long row0, long row_end, int pitch, int col0, int tid) {
    const int r = tid >> 3, c8 = tid & 7;
    u32x4 v{0u, 0u, 0u, 0u};
    if (row0 + r < row_end && row0 + r >= 0) v = *reinterpret_cast<const u32x4*>(src + (row0 + r) * (long)pitch + col0 + c8 * 8);
    unsigned* d = reinterpret_cast<unsigned*>(dst + r * MHA_VP + c8 * 8);
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
}

// sum the partial accumulators of the two k-slice waves (kk = 1 -> kk = 0) of each output half; contains a barrier
__device__ __forceinline__ void reduce_kslices(f32x16& o, float* sRed, int nb, int kk, int lane) {
    if (kk == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sRed[nb * 1024 + acc_row(r, lane) * 32 + (lane & 31)] = o[r];
    }
    __syncthreads();
    if (kk == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] += sRed[nb * 1024 + acc_row(r, lane) * 32 + (lane & 31)];
    }
}

template <bool REL>
__global__ __launch_bounds__(256) void k_mha_fwd4(const MhaArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int LkS = ((a.Lk + 31) & ~31) + 4;
    float* sS = reinterpret_cast<float*>(smem_raw);                  // [32][LkS]
    float* sBD = sS + 32 * LkS;                                      // [4 waves][32][64]; later [2][32][32] reduction scratch
    float* sStat = sBD + 4 * 32 * 64;                                // [32][2]
    bf16_t* sV = reinterpret_cast<bf16_t*>(sStat + 64);              // [32][MHA_VP]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, row = lane & 31, half = lane >> 5;
    const int i0 = blockIdx.x * 32, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int klen = a.klen != nullptr ? a.klen[b] : a.Lk;
    const int qi = min(i0 + row, a.Lq - 1);
    const bf16_t* qrow = a.q + ((long)b * a.Lq + qi) * a.q_pitch + h * MHA_DH + half * 8;
    bf16x8 qu[4], qv[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 f = ld_frag(qrow + kk * 16);
        if (REL) {
            qu[kk] = add_bias_frag(f, a.bias_u + h * MHA_DH + kk * 16 + half * 8);
            qv[kk] = add_bias_frag(f, a.bias_v + h * MHA_DH + kk * 16 + half * 8);
        } else {
            qu[kk] = f;
        }
    }
    float* myBD = sBD + w * 32 * 64;
    for (int j0 = w * 32; j0 < a.Lk; j0 += 128) {
        const int kj = min(j0 + row, a.Lk - 1);
        const bf16_t* krow = a.k + ((long)b * a.Lk + kj) * a.kv_pitch + h * MHA_DH + half * 8;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qu[kk], ld_frag(krow + kk * 16), acc, 0, 0, 0);
        if (REL) {
            const int rbase = (a.Lq - 1) + j0 - i0 - 31;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                int pr = rbase + nb * 32 + row;
                pr = pr < 0 ? 0 : (pr > 2 * a.Lq - 2 ? 2 * a.Lq - 2 : pr);
                const bf16_t* prow = a.pe + (long)pr * a.pe_pitch + h * MHA_DH + half * 8;
                f32x16 accb;
#pragma unroll
                for (int r = 0; r < 16; ++r) accb[r] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qv[kk], ld_frag(prow + kk * 16), accb, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) myBD[acc_row(r, lane) * 64 + nb * 32 + row] = accb[r];
            }
            // the scratch is private to this wave: LDS operations of one wave complete in order, only the compiler must not reorder
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = acc_row(r, lane);
                acc[r] += myBD[il * 64 + row - il + 31];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int il = acc_row(r, lane), i = i0 + il, j = j0 + row;
            const bool ok = j < klen && j < a.Lk && (!a.causal || j <= i);
            sS[il * LkS + j0 + row] = ok ? acc[r] * a.scale : -INFINITY;
        }
    }
    __syncthreads();
    {   // row statistics: 8 threads per row
        const int r = tid >> 3, part = tid & 7;
        float m = -INFINITY;
        for (int j = part; j < a.Lk; j += 8) m = fmaxf(m, sS[r * LkS + j]);
        m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 4, 64));
        float s = 0.f;
        if (m > -INFINITY)
            for (int j = part; j < a.Lk; j += 8) s += __expf(sS[r * LkS + j] - m);
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (part == 0) { sStat[r * 2] = m; sStat[r * 2 + 1] = s > 0.f ? 1.f / s : 0.f; }
    }
    __syncthreads();
    const int LkR = (a.Lk + 31) & ~31;
    const bool drop_on = a.drop.seed != nullptr;
    const unsigned dkey = drop_on ? drop_key(a.drop) : 0u;
    for (int il = w * 8; il < w * 8 + 8; ++il) {
        const float m = sStat[il * 2], inv = sStat[il * 2 + 1];
        const int i = i0 + il;
        for (int j = lane; j < LkR; j += 64) {
            float p = 0.f;
            if (j < a.Lk && inv > 0.f) p = __expf(sS[il * LkS + j] - m) * inv;
            if (i < a.Lq && j < a.ldp) a.probs[((long)bh * a.Lq + i) * a.ldp + j] = f2bf(p);
            if (drop_on) p = drop_keep(dkey, a.drop.thresh, (unsigned)(((long)bh * a.Lq + i) * a.ldp + j)) ? p * a.drop.scale : 0.f;
            sS[il * LkS + j] = p;
        }
    }
    // ctx = P·V: wave w -> output half nb = w & 1, key slice kk = w >> 1 of every staged 32-key block
    const int nb = w & 1, kk = w >> 1;
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    for (int j0 = 0; j0 < a.Lk; j0 += 32) {
        __syncthreads();
        stage_rows64_256(sV, a.v, (long)b * a.Lk + j0, (long)b * a.Lk + a.Lk, a.kv_pitch, h * MHA_DH, tid);
        __syncthreads();
        const bf16x8 fp = f32row_frag(sS + row * LkS + j0 + kk * 16 + half * 8);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp, gather_frag(sV, MHA_VP, kk * 16, nb * 32, lane), o, 0, 0, 0);
    }
    __syncthreads();
    reduce_kslices(o, sBD, nb, kk, lane);
    if (kk == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + acc_row(r, lane);
            if (i < a.Lq) a.ctx[((long)b * a.Lq + i) * a.ctx_pitch + h * MHA_DH + nb * 32 + row] = f2bf(o[r]);
        }
    }
}

template <bool REL>
__global__ __launch_bounds__(256) void k_mha_bwd_q4(const MhaArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int LkS = ((a.Lk + 31) & ~31) + 4;
    float* sS = reinterpret_cast<float*>(smem_raw);                  // [32][LkS]: dP, then dS
    float* sRed = sS + 32 * LkS;                                     // [2][32][32]
    bf16_t* sV = reinterpret_cast<bf16_t*>(sRed + 2 * 1024);         // [32][MHA_VP]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, row = lane & 31, half = lane >> 5;
    const int i0 = blockIdx.x * 32, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int qi = min(i0 + row, a.Lq - 1);
    {
        const bf16_t* drow = a.dctx + ((long)b * a.Lq + qi) * a.dctx_pitch + h * MHA_DH + half * 8;
        bf16x8 fd[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fd[kk] = ld_frag(drow + kk * 16);
        for (int j0 = w * 32; j0 < a.Lk; j0 += 128) {
            const int kj = min(j0 + row, a.Lk - 1);
            const bf16_t* vrow = a.v + ((long)b * a.Lk + kj) * a.kv_pitch + h * MHA_DH + half * 8;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd[kk], ld_frag(vrow + kk * 16), acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) sS[acc_row(r, lane) * LkS + j0 + row] = acc[r];
        }
    }
    __syncthreads();
    const int LkR = (a.Lk + 31) & ~31;
    const bool drop_on = a.drop.seed != nullptr;
    const unsigned dkey = drop_on ? drop_key(a.drop) : 0u;
    for (int il = w * 8; il < w * 8 + 8; ++il) {
        const int i = i0 + il;
        const bool live = i < a.Lq;
        const bf16_t* prow = a.probs + ((long)bh * a.Lq + (live ? i : 0)) * a.ldp;
        float part = 0.f;
        if (live)
            for (int j = lane; j < a.Lk; j += 64) {
                float dp = sS[il * LkS + j];
                if (drop_on) {
                    dp = drop_keep(dkey, a.drop.thresh, (unsigned)(((long)bh * a.Lq + i) * a.ldp + j)) ? dp * a.drop.scale : 0.f;
                    sS[il * LkS + j] = dp;
                }
                part += bf2f(prow[j]) * dp;
            }
        const float dsum = wave_sum(part);
        for (int j = lane; j < LkR; j += 64) {
            float d = 0.f;
            if (live && j < a.Lk) d = bf2f(prow[j]) * (sS[il * LkS + j] - dsum) * a.scale;
            sS[il * LkS + j] = d;
            if (live && j < a.ldp) a.ds[((long)bh * a.Lq + i) * a.ldp + j] = f2bf(d);
        }
    }
    const int nb = w & 1, kk = w >> 1;
    f32x16 oac, obd;
#pragma unroll
    for (int r = 0; r < 16; ++r) { oac[r] = 0.f; obd[r] = 0.f; }
    for (int j0 = 0; j0 < a.Lk; j0 += 32) {
        __syncthreads();
        stage_rows64_256(sV, a.k, (long)b * a.Lk + j0, (long)b * a.Lk + a.Lk, a.kv_pitch, h * MHA_DH, tid);
        __syncthreads();
        const bf16x8 fs = f32row_frag(sS + row * LkS + j0 + kk * 16 + half * 8);
        oac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fs, gather_frag(sV, MHA_VP, kk * 16, nb * 32, lane), oac, 0, 0, 0);
    }
    if (REL) {
        int r_lo = (a.Lq - 1) - (i0 + 31), r_hi = (a.Lq - 1) + (a.Lk - 1) - i0;
        if (r_lo < 0) r_lo = 0;
        if (r_hi > 2 * a.Lq - 2) r_hi = 2 * a.Lq - 2;
        for (int r0 = r_lo & ~31; r0 <= r_hi; r0 += 32) {
            __syncthreads();
            stage_rows64_256(sV, a.pe, r0, 2 * a.Lq - 1, a.pe_pitch, h * MHA_DH, tid);
            __syncthreads();
            bf16x8 fs;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = r0 + kk * 16 + half * 8 + e - (a.Lq - 1) + i0 + row;
                fs[e] = (short)f2bf((j >= 0 && j < a.Lk) ? sS[row * LkS + j] : 0.f);
            }
            obd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fs, gather_frag(sV, MHA_VP, kk * 16, nb * 32, lane), obd, 0, 0, 0);
        }
    }
    __syncthreads();
    reduce_kslices(oac, sRed, nb, kk, lane);
    if (REL) {
        __syncthreads();
        reduce_kslices(obd, sRed, nb, kk, lane);
    }
    if (kk == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + acc_row(r, lane);
            if (i >= a.Lq) continue;
            const int col = h * MHA_DH + nb * 32 + row;
            const long rr = (long)b * a.Lq + i;
            a.dq[rr * a.dq_pitch + col] = f2bf(oac[r] + obd[r]);
            if (REL) {
                a.dq_ac[rr * a.aux_pitch + col] = f2bf(oac[r]);
                a.dq_bd[rr * a.aux_pitch + col] = f2bf(obd[r]);
            }
        }
    }
}

template <bool REL>
__global__ __launch_bounds__(256) void k_mha_bwd_kv4(const MhaArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t sP[32 * MHA_TP], sD[32 * MHA_TP], sDC[32 * MHA_VP], sQ[32 * MHA_VP];
    __shared__ float sRed[2 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, row = lane & 31;
    const int nb = w & 1, kk = w >> 1;
    const int j0 = blockIdx.x * 32, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const bool drop_on = a.drop.seed != nullptr;
    const unsigned dkey = drop_on ? drop_key(a.drop) : 0u;
    f32x16 odv, odk;
#pragma unroll
    for (int r = 0; r < 16; ++r) { odv[r] = 0.f; odk[r] = 0.f; }
    const bf16_t* pb = a.probs + (long)bh * a.Lq * a.ldp;
    const bf16_t* db = a.ds + (long)bh * a.Lq * a.ldp;
    for (int i0 = 0; i0 < a.Lq; i0 += 32) {
        __syncthreads();
        for (int idx = tid; idx < 32 * 32; idx += 256) {
            const int r = idx >> 5, c = idx & 31, i = i0 + r, j = j0 + c;
            const bool ok = i < a.Lq && j < a.Lk;
            bf16_t pv = ok ? pb[(long)i * a.ldp + j] : (bf16_t)0;
            if (drop_on && ok) pv = drop_keep(dkey, a.drop.thresh, (unsigned)(((long)bh * a.Lq + i) * a.ldp + j)) ? f2bf(bf2f(pv) * a.drop.scale) : (bf16_t)0;
            sP[r * MHA_TP + c] = pv;
            sD[r * MHA_TP + c] = ok ? db[(long)i * a.ldp + j] : (bf16_t)0;
        }
        stage_rows64_256(sDC, a.dctx, (long)b * a.Lq + i0, (long)b * a.Lq + a.Lq, a.dctx_pitch, h * MHA_DH, tid);
        {   // q rows (+ u bias, rounded to bf16 as in the forward)
            const int r = tid >> 3, c8 = tid & 7;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = 0.f;
            if (i0 + r < a.Lq) {
                unpack8(*reinterpret_cast<const u32x4*>(a.q + ((long)b * a.Lq + i0 + r) * a.q_pitch + h * MHA_DH + c8 * 8), f);
                if (REL) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += a.bias_u[h * MHA_DH + c8 * 8 + e];
                }
            }
            const u32x4 v = pack8(f);
            unsigned* d = reinterpret_cast<unsigned*>(sQ + r * MHA_VP + c8 * 8);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        const bf16x8 fp = gather_frag(sP, MHA_TP, kk * 16, 0, lane);
        const bf16x8 fs = gather_frag(sD, MHA_TP, kk * 16, 0, lane);
        odv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp, gather_frag(sDC, MHA_VP, kk * 16, nb * 32, lane), odv, 0, 0, 0);
        odk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fs, gather_frag(sQ, MHA_VP, kk * 16, nb * 32, lane), odk, 0, 0, 0);
    }
    __syncthreads();
    reduce_kslices(odv, sRed, nb, kk, lane);
    __syncthreads();
    reduce_kslices(odk, sRed, nb, kk, lane);
    if (kk == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + acc_row(r, lane);
            if (j >= a.Lk) continue;
            const long o = ((long)b * a.Lk + j) * a.dkv_pitch + h * MHA_DH + nb * 32 + row;
            a.dk[o] = f2bf(odk[r]);
            a.dv[o] = f2bf(odv[r]);
        }
    }
}

__global__ __launch_bounds__(256) void k_mha_bwd_pe4(const MhaArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t sA[32 * MHA_TP], sQ[32 * MHA_VP];
    __shared__ float sRed[2 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, row = lane & 31;
    const int nb = w & 1, kk = w >> 1;
    const int r0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    const bf16_t* dsb = a.ds + ((long)b * a.H + h) * a.Lq * a.ldp;
    for (int i0 = 0; i0 < a.Lq; i0 += 32) {
        const int jmin = r0 - (a.Lq - 1) + i0, jmax = r0 + 31 - (a.Lq - 1) + i0 + 31;
        if (jmax < 0 || jmin >= a.Lk) continue;            // uniform across the workgroup
        __syncthreads();
        for (int idx = tid; idx < 32 * 32; idx += 256) {
            const int il = idx >> 5, rl = idx & 31;
            const int i = i0 + il, j = r0 + rl - (a.Lq - 1) + i;
            sA[il * MHA_TP + rl] = (i < a.Lq && j >= 0 && j < a.Lk) ? dsb[(long)i * a.ldp + j] : (bf16_t)0;
        }
        {
            const int r = tid >> 3, c8 = tid & 7;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = 0.f;
            if (i0 + r < a.Lq) {
                unpack8(*reinterpret_cast<const u32x4*>(a.q + ((long)b * a.Lq + i0 + r) * a.q_pitch + h * MHA_DH + c8 * 8), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += a.bias_v[h * MHA_DH + c8 * 8 + e];
            }
            const u32x4 v = pack8(f);
            unsigned* d = reinterpret_cast<unsigned*>(sQ + r * MHA_VP + c8 * 8);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gather_frag(sA, MHA_TP, kk * 16, 0, lane), gather_frag(sQ, MHA_VP, kk * 16, nb * 32, lane), o, 0, 0, 0);
    }
    __syncthreads();
    reduce_kslices(o, sRed, nb, kk, lane);
    if (kk == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = r0 + acc_row(r, lane);
            if (rr < 2 * a.Lq - 1) a.pe_part[((long)b * (2 * a.Lq - 1) + rr) * a.dpe_pitch + h * MHA_DH + nb * 32 + row] = o[r];
        }
    }
}
