// Shared declarations of the implicit-GEMM forward contraction kernels (igemm_fwd.hip: 4-wave kernels; igemm_p8.hip: the persistent
// 8-wave kernel).
#pragma once
#include "common.h"

struct IgemmFwdArgs {
    const bf16_t* in;
    const bf16_t* wt;      // [Co][wt_taps][Ci]
    void* out;             // bf16 or f32 pixels
    bf16_t* out_pre;       // optional pre-activation copy (GELU epilogue)
    const float* bias;     // optional [Co]
    const bf16_t* addend;  // optional bf16 pixels with the geometry of `out`, added before the activation
    float* stats;          // optional BatchNorm partials [gridDim.x][2][Co]: row blockIdx.x = this M tile's column sums / sums of squares
    const int* plan;       // device copy of the plan words
    int Nimg, in_pix, Ci, in_pitch;      // images, pixels per source image, contraction channels per tap (multiple of 64), source pitch
    int Co, out_pix, out_pitch, wt_taps; // output channels, pixels per target image, target pitch, taps physically present in wt
    int act, out_f32;          // act: 0 none, 1 GELU(erf) (pre-activation kept in out_pre), 2 ReLU
    float alpha;               // out = alpha * dropout(act(acc + bias)) + addend
    DropArgs drop;             // drop.seed == nullptr: no dropout
    int epi_batched;           // epilogue: request all rows' operands before using the first (tuning knob "epi_batched", default on)
    // BatchNorm-backward fusion (data-gradient launches whose result is the gradient of a BatchNorm+ReLU output y = relu(bn(x) [+ res])):
    // out = g = (y > 0 ? result : 0) and stats rows = this tile's column sums of {g, g * (x - mean) * rstd}   (bnb_x == nullptr: off)
    const bf16_t* bnb_y;
    const bf16_t* bnb_x;
    const float* bnb_mean;
    const float* bnb_rstd;
    const float* bnb_gamma;    // with bnb_y == nullptr (no residual branch): the mask is recomputed as bn(x) > 0 with the forward's own
    const float* bnb_beta;     // expression (norm_act.hip k_bn_act_fwd) instead of being read from y
    int bnb_act;               // 1 ReLU (bnb_y = the output y, or null), 2 Swish: g = result * swish'(bn(x) + r), bnb_y = the residual input r or null
};


#define LDS_SWZ(row, chunk) ((row) * 64 + ((((chunk) ^ (((row) >> 1) & 7))) << 3))

// ---- persistent 8-wave kernel (igemm_p8.hip) -------------------------------------------------------------------------------------
// Plan words of the 256 x 128 persistent kernel ("p8" format):
//   w[0] = P8_MAGIC, w[1] = M tiles, w[2] = N tiles (gy), w[3] = word offset of the tile descriptors, w[4] = word offset of the row
//   tables, w[5] = work items incl. holes = ceil(M tiles / 8) * 8 * gy
//   descriptor of M tile m (64 words): [0] taps, [1] valid rows, [2..10] delta[9], [11..19] tw[9]
//   row table of M tile m ([256][2] words): (source pixel, target pixel) as GLOBAL pixel indices (image * pixels + position); past the end: (row 0's source, -1)
#define P8_MAGIC 0x50380001
#define P8_HDR_WORDS 8
#define P8_DESC_WORDS 64
#define P8_BM 256
#define P8_BN 128
int igemm_p8_launch(const IgemmFwdArgs& a, const int* meta, hipStream_t stream);
