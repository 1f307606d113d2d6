// Geometry shared by the implicit-GEMM kernels (forward / data-gradient / linear, and weight-gradient).
//
// The iteration space is M = Nimg*Ha*Wa "positions" (n, a, b):
//   source pixel  = (a*S + dy[t], b*S + dx[t])  in the [Hi x Wi] grid of `in`   (zero outside the grid)
//   target pixel  = (a*OS + oy0,  b*OS + ox0)   in the [Ho x Wo] grid of `out`
// forward conv:  S = stride, dy = kh - pad, OS = 1.      dgrad stride 1: same with transposed weights.
// dgrad stride 2: one launch per output parity class (oy0, ox0) with OS = 2, S = 1 and that class's taps.
// linear:        Hi = Wi = Ha = Wa = 1, one tap.
#pragma once
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

// NOTE on two hipcc behaviours these kernels are written around (both measured: 5-10x on the first version):
//  * a kernel-argument array indexed with a run-time value is copied to scratch memory -> taps are copied to LDS
//    (or selected with compile-time indices) at kernel start;
//  * a global load under a lane-dependent branch gets its own `s_waitcnt vmcnt(0)` -> every staging load is
//    unconditional from a clamped address, and the "out of grid -> zero" select is applied when the registers are
//    written to LDS, i.e. after the MFMA block the loads are meant to overlap with.

static inline int fill_geom(IgemmGeom& g, int Nimg, int Hi, int Wi, int Ci, int in_pitch, int Co, int Ho, int Wo, int out_pitch,
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
