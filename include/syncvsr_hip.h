/* libsyncvsr_hip.so — C ABI of the MI355X (gfx950) SyncVSR training hot path.
 *
 * The reference (KAIST-AILab/SyncVSR) has no native/FFI layer: its hot path is reached through torch.nn modules
 * (SURVEY.md §8b).  Each entry point below therefore names the reference *module call* it replaces.
 *
 * Conventions (all entry points):
 *   - arguments are raw device pointers + explicit sizes; `void*` activations are bf16 unless stated; parameters,
 *     statistics and gradients are fp32.  The caller owns every buffer, including workspaces.
 *   - asynchronous: work is enqueued on `stream`; the call returns 0, SVSR_ERR_ARG (1001) for an unsupported
 *     shape/argument, or a hipError_t value if the launch failed.  Nothing throws; the only global mutable state is the
 *     table of result-preserving tuning knobs behind svsr_tune() (the library never reads the environment).
 *   - activations are NHWC ("pixels x channels"); a frame index is just the leading pixel index.
 *   - REPRODUCIBLE: no kernel accumulates floating-point values with atomics.  Grid-wide sums (BatchNorm statistics, split-K
 *     weight gradients, bias / LayerNorm gradients, losses, the gradient norm) are written as one partial row per workgroup
 *     into a caller-owned workspace and added in a fixed order by a second small launch (svsr_colsum_rows or a specialised
 *     finaliser), so two identical calls give bit-identical results.  Entry points named *_rows / *_plan are HOST-side
 *     queries (no launch) that tell the caller how large such a workspace must be for a shape.
 *
 * The declarations are kept one per statement in a regular form because syncvsr_amd/_lib.py derives its ctypes
 * signatures from this file.
 */
#ifndef SYNCVSR_HIP_H
#define SYNCVSR_HIP_H

#include <stdint.h>

#ifndef __HIP_PLATFORM_AMD__
typedef void* hipStream_t;
#else
#include <hip/hip_runtime_api.h>
#endif

#define SVSR_OK 0
#define SVSR_ERR_ARG 1001

/* one problem of svsr_igemm_wgrad_group: the arguments of a svsr_igemm_wgrad call (device pointers; meta = the HOST meta[8] of the plan) */
typedef struct svsr_wgrad_problem {
    const void* x; const void* dy; float* dw; float* dbias; const int* plan_dev; const int* meta;
    int Nimg; int in_pix; int Ci; int in_pitch; int Co; int out_pix; int out_pitch; int wt_taps;
} svsr_wgrad_problem;

/* one encoder layer of svsr_enc_fwd: DEVICE pointers.  Weights: the bf16 shadows [out][in] of query|key|value (adjacent: [1536][512]),
 * attention.output.dense [512][512], intermediate.dense [2048][512], output.dense [512][2048]; biases and LayerNorm parameters fp32.
 * Written by the launch (what the backward reads): qkv bf16 [R][1536], probs bf16 [B*8][S][ldp = ceil8(S)], ctx / ao / x1 / f / xout bf16
 * [R][512], z / hg bf16 [R][2048], LayerNorm statistics m1 r1 m2 r2 fp32 [R].  site_*: dropout sites of the attention probabilities,
 * BertSelfOutput's and BertOutput's dropout. */
typedef struct svsr_enc_layer {
    const void* wqkv; const void* wo; const void* w1; const void* w2;
    const float* bqkv; const float* bo; const float* b1; const float* b2; const float* g1; const float* be1; const float* g2; const float* be2;
    void* qkv; void* probs; void* ctx; void* ao; void* x1; void* z; void* hg; void* f; void* xout;
    float* m1; float* r1; float* m2; float* r2;
    unsigned site_probs; unsigned site_ao; unsigned site_fo; unsigned pad_;
} svsr_enc_layer;

/* one encoder layer of svsr_enc_bwd (the backward of svsr_enc_fwd: autograd of HF BertLayer, reference LRW/video/src/lightning.py:92,152-156):
 * DEVICE pointers.  Weights: the TRANSPOSED bf16 shadows [in][out] the data-gradient launches read — output.dense [2048][512],
 * intermediate.dense [512][2048], attention.output.dense [512][512], query|key|value [512][1536]; g1 / g2: LayerNorm weights.  Read: what
 * the forward kept (f, x1, ao, the layer input xin, z, qkv, probs, LayerNorm statistics).  Written: see svsr_enc_bwd. */
typedef struct svsr_enc_bwd_layer {
    const void* w2t; const void* w1t; const void* wot; const void* wqkvt;
    const float* g1; const float* g2;
    const void* f; const void* x1; const void* ao; const void* xin; const void* z; const void* qkv; const void* probs;
    const float* m1; const float* r1; const float* m2; const float* r2;
    void* ds2; void* df; void* dz; void* dx1; void* ds1; void* dao; void* dqkv; void* dx;
    float* part1; float* part2;
    unsigned site_probs; unsigned site_ao; unsigned site_fo; unsigned pad_;
} svsr_enc_bwd_layer;

#ifdef __cplusplus
extern "C" {
#endif

/* ---- plumbing (runtime.hip) ------------------------------------------------------------------------------------
 * svsr_tune: sets a result-preserving tuning knob ("igemm_tile", "igemm_m128", "wg_blocks", "w3_blocks", "ln_rpb", "stem_lds_fwd",
 * "stem_lds_bwd", "igemm_lds_pad", "igemm_bn64_below", "wg_short_k", "igemm_ksplit", "epi_batched", "stem_wg_pipe",
 * "stem_fwd_dma", "igemm_lin_bn64", "p8", "p8_grid", "p8_min_items", "p8_ph", "p8_stagger", "wg_imgmajor", "p8_bn64", "igemm_ns64", "wg_units", "wg_unit_max", "wg_unit_min", "igemm_ksplit128", "wg_xcd", "w3_waves", "reduce_cus", "w3_dense", "p8_wide": tile shapes, split counts, kernel-variant switches — documented at the table in
 * runtime.hip; never read from the environment); unknown key -> SVSR_ERR_ARG.  svsr_tune_value reads a knob back.  The knobs that decide
 * the ORDER in which a weight gradient's or a BatchNorm statistic's partial sums are added (reduce_cus — the compute-unit count the splits
 * are planned for: a fixed 256, not the device's —, wg_blocks, wg_units, wg_unit_max, wg_unit_min, wg_short_k, w3_blocks, w3_waves, w3_dense) are what a
 * bit-identical resume depends on: engine.TrainStep.state_dict() records them, load_state_dict() warns when they differ.
 * svsr_colsum_rows: out[c] (+)= scale * sum_{r<nrows} ws[r*ld + c], rows added in a fixed order; columns [0,n0) go to out0,
 * [n0,n0+n1) to out1 (may be null when n1 = 0); accumulate != 0 adds to the existing values. */
int svsr_tune(const char* key, int value);
int svsr_tune_value(const char* key, int* value);
int svsr_colsum_rows(const float* ws, int nrows, int64_t ld, float* out0, int64_t n0, float* out1, int64_t n1, int accumulate, float scale, hipStream_t stream);
/* up to any number of svsr_colsum_rows problems, 16 per launch (the postponed parameter-gradient reductions of a layer's backward: LayerNorm
 * weight / bias, linear biases — autograd's accumulation into .grad of lightning.py's modules).  entries: n records of 64 bytes in HOST memory,
 * {const float* ws; float* out0; float* out1; int64_t ld, n0, n1; int32_t nrows, accumulate; float scale; int32_t reserved}; the outputs of
 * different records must not overlap.  Same additions in the same order as n separate calls. */
int svsr_colsum_rows_multi(const void* entries, int n, hipStream_t stream);

/* ---- implicit-GEMM contractions (igemm_fwd.hip) -------------------------------------------------------------------
 * svsr_igemm_fwd replaces: nn.Conv2d forward of resnet.layer{1..4} (reference LRW/video/src/tcn/models/resnet.py:8-16,
 *   59-72 / timm resnet18 via lightning.py:55,114-117), their input-gradient (autograd), and every nn.Linear forward /
 *   input-gradient of the BERT encoder and heads (lightning.py:92,107,82,161,168).
 * The rows of the contraction come from a PLAN built on the host once per shape (svsr_conv_plan / svsr_rows_plan) and kept
 *   on the device: output positions are grouped into classes that share the same set of in-grid taps, so the kernel never
 *   multiplies by a convolution's zero padding and a stride-2 input-gradient (four output-parity classes) is one launch.
 *   svsr_*_plan(words = null) returns the number of int32 words; a second call fills `words` (host) and meta[8] =
 *   {bm, bn, ns, tiles, grid_y, classes, max taps, rows}: the kernel instantiation k_igemm_fwd_glds<bm,bn,ns> the launch will
 *   use and `tiles` = rows of BatchNorm partials it writes.  Negative return = -SVSR_ERR_ARG.
 *   svsr_conv_plan: k x k / stride / pad convolution over Nimg images whose FORWARD input is H x W; mode 0 forward
 *   (in = x [H][W] -> out = y [Ho][Wo], wt [Co][k*k][Ci]), mode 1 input-gradient (in = dy [Ho][Wo] -> out = dx [H][W], wt =
 *   transposed weights [Ci][k*k][Co]; pixels no tap reaches are written as 0 (+ bias / addend)), mode 2 the same for an
 *   in-place accumulation (addend aliases out): pixels no tap reaches are left alone.
 *   svsr_rows_plan: dense layer over Nimg sequences, row (n, j < P) reads source row n*in_pix + src0 + j and writes target row
 *   n*out_pix + dst0 + j (plain linear: P = 1, in_pix = out_pix = 1).
 * svsr_igemm_fwd(plan_dev = device copy of the words, meta = the host meta): in [Nimg][in_pix] pixels of pitch in_pitch with
 *   Ci channels per tap, out [Nimg][out_pix] pixels of pitch out_pitch with Co channels; wt bf16 [Co][wt_taps][Ci].
 *   Optional epilogues: +bias, +addend (bf16 pixels laid out like `out`: residual-gradient merge; may alias `out`),
 *   out = alpha * dropout(act(acc + bias)) + addend with act 0 none / 1 exact GELU (pre-activation saved to out_pre) / 2 ReLU
 *   (LRS PositionwiseFeedForward, transformer/positionwise_feed_forward.py:28-30; alpha carries the Conformer's 0.5 macaron
 *   scale and the sqrt(d) embedding scale, encoder_layer.py:97,131, embedding.py:208); dropout(p) with the keep decision
 *   hash(*drop_seed, drop_site, output element index) (drop_seed null or p = 0: off; see svsr_scale_bf16), fp32 output,
 *   per-channel BatchNorm partial sums stats[tiles][2][Co] (plain stores; reduced by svsr_bn_finalize). */
int svsr_conv_plan(int mode, int Nimg, int H, int W, int Co_out, int k, int stride, int pad, int* words, int cap_words, int* meta);
int svsr_rows_plan(int Nimg, int P, int src0, int dst0, int Co_out, int* words, int cap_words, int* meta);
int svsr_igemm_fwd_kgroups(const int* meta, int Ci, int Co, int bn_epilogue);   /* host query: 2 when the launch splits K over two wave groups (k_igemm_fwd_glds<64,64,4,2>) */
int svsr_igemm_fwd(const void* in, const void* wt, void* out, void* out_pre, const float* bias, const void* addend, float* stats, const int* plan_dev, const int* meta, int Nimg, int in_pix, int Ci, int in_pitch, int Co, int out_pix, int out_pitch, int wt_taps, int act, int out_f32, float alpha, const unsigned* drop_seed, unsigned drop_site, float drop_p, hipStream_t stream);

/* svsr_igemm_wgrad replaces: the weight-gradient of the same Conv2d / Linear layers (torch autograd).
 * dw fp32 [Co][wt_taps][Ci] is ACCUMULATED (dw += ...).  x = forward input pixels [Nimg][in_pix] (pitch in_pitch, Ci channels),
 * dyp = output-gradient pixels [Nimg][out_pix] (pitch out_pitch, Co channels).  dbias (optional, fp32 [Co]) += column sums of
 * dyp = the bias gradient of an nn.Linear, from the tiles the kernel stages anyway (one MFMA against a fragment of ones).
 * Rows come from a host-built plan, per tap the (source pixel, target pixel) pairs whose source lies inside the grid (no products
 * with the zero padding): svsr_wgrad_plan for a k x k / stride / pad convolution of Nimg images [H][W], svsr_wgrad_rows_plan for
 * a dense layer over Nimg sequences (row (n, j < P): source row n*in_pix + src0 + j, target row n*out_pix + dst0 + j).  Both
 * return the number of int32 words (words = null: count only; negative = error), meta[8] = {tile edge, ring depth, K splits,
 * chunks per split, tasks, taps, max positions per tap} and *part_floats = the workspace svsr_igemm_wgrad needs for its split-K
 * slabs (any contents; added into dw / dbias in a fixed order; 0 when a single split writes dw directly). */
int svsr_wgrad_plan(int Nimg, int H, int W, int Ci, int Co, int k, int stride, int pad, int* words, int cap_words, int* meta, int64_t* part_floats);
int svsr_wgrad_rows_plan(int Nimg, int P, int src0, int dst0, int Ci, int Co, int has_bias, int* words, int cap_words, int* meta, int64_t* part_floats);
int svsr_igemm_wgrad(const void* x, const void* dyp, float* dw, float* dbias, const int* plan_dev, const int* meta, int Nimg, int in_pix, int Ci, int in_pitch, int Co, int out_pix, int out_pitch, int wt_taps, float* part, int64_t part_floats, hipStream_t stream);
/* svsr_igemm_wgrad_group: n independent svsr_igemm_wgrad problems in ONE launch — the weight gradients of the encoder's and the heads'
 * nn.Linear layers (reference lightning.py:82,92,107: 26 short contractions of a backward pass; as launches of their own each is 7-17 us
 * of latency for ~1 us of work).  Every problem must carry a plan without K split on 64-wide tiles (meta[0..2] = {64, 3, 1}); anything
 * else returns SVSR_ERR_ARG.  table_dev: caller-owned device buffer of >= svsr_igemm_wgrad_group_bytes(n) bytes.  Results are identical
 * to n separate svsr_igemm_wgrad calls (same workgroup code, one writer per element). */
int64_t svsr_igemm_wgrad_group_bytes(int n);
int svsr_igemm_wgrad_group(const svsr_wgrad_problem* problems, int n, void* table_dev, int64_t table_bytes, hipStream_t stream);

/* svsr_conv3x3_c64: conv3x3(64, 64), stride 1, pad 1 (layer1 of the trunk, resnet.py:8-10,36,53) forward and, with the
 * transposed weights and mirrored taps, its input-gradient; persistent workgroups, weights resident in LDS
 * (conv3x3_c64.hip).  in/out/addend bf16 [Nimg][H][W][64]; wt bf16 [64][9][64]; tap t reads pixel (y+dy[t], x+dx[t]) with
 * weight tap tw[t] (HOST arrays of 9 ints); out = conv (+ addend); stats [rows][2][64] with rows =
 * svsr_conv3x3_c64_stat_rows(Nimg, H, W) (one per persistent workgroup).  Requires W <= 29. */
int svsr_conv3x3_c64_stat_rows(int Nimg, int H, int W);
/* pixtab (both launches below): device copy of svsr_conv3x3_c64_pixtab's table for (Nimg, H, W) — pixel index or -1 per padded
 * coordinate — or null (the kernel then derives the source pixel of every staged row by arithmetic: ~500 instructions per chunk and thread).
 * svsr_conv3x3_c64_pixtab(..., out = null) returns the entry count. */
int64_t svsr_conv3x3_c64_pixtab(int Nimg, int H, int W, int* out, int64_t cap);
int svsr_conv3x3_c64(const void* in, const void* wt, void* out, const void* addend, float* stats, int Nimg, int H, int W, const int* dy, const int* dx, const int* tw, const int* pixtab, hipStream_t stream);

/* Data-gradient launches with the FIRST PASS OF THE BATCHNORM BACKWARD in their epilogue (replaces the separate reduce pass of
 * svsr_bn_act_bwd for the ReLU trunk; reference backward of tcn/models/resnet.py:59-72 = autograd of bn -> relu).  The launch computes
 * dL/dy (+ addend, which may alias out) for a tensor y = relu(bn(x) [+ residual]) that has the geometry of `out`, and while the tile
 * is in registers stores g = (y > 0 ? dL/dy : 0) INSTEAD of dL/dy and writes, per row tile / persistent workgroup, the column sums
 * of g and of g * (x - mean) * rstd into stats[rows][2][Co] (rows = meta[3] of the plan, resp. svsr_conv3x3_c64_stat_rows).
 * y == NULL (allowed only when y has no residual branch): the mask is recomputed from x as bn(x) > 0 with gamma / beta in the forward
 * pass's own arithmetic (svsr_bn_act_fwd) and y is not read; otherwise gamma / beta may be NULL.
 * act = 1: ReLU as described.  act = 2: Swish (LRS trunk, Conformer convolution module): g = dL/dy * swish'(bn(x) + r), where `y` is the
 * residual INPUT r of the output (NULL: none) and gamma / beta are required.
 * svsr_igemm_dgrad_bn needs a plan that visits every target pixel once (svsr_conv_plan mode 1), Co % 8 == 0, out_pitch % 8 == 0.
 * svsr_bn_bwd_from_stats then adds the rows in a fixed order (dgamma +=, dbeta +=, coef[3][C] scratch) and writes
 * dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)). */
int svsr_igemm_dgrad_bn(const void* in, const void* wt, void* out, const void* addend, float* stats, const int* plan_dev, const int* meta, int Nimg, int in_pix, int Ci, int in_pitch, int Co, int out_pix, int out_pitch, int wt_taps, const void* y, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int act, hipStream_t stream);
/* The same epilogue for a linear layer / convolution whose INPUT was y = dropout(relu(z)) (positionwise_feed_forward.py:28-30): stores
 * dz = (y > 0 ? gscale * dL/dy : 0) (gscale = 1 / (1 - p): y is zero where ReLU or the dropout mask cut) and writes the column sums of dz per row tile
 * into the first half of stats[meta[3]][2][Co] — the partial rows of the bias gradient of the layer in front (svsr_colsum_rows adds them).
 * zeros / ones: device vectors of Co floats holding 0 / 1.  4-wave plans only (not the persistent 3x3 kernel). */
int svsr_igemm_dgrad_relu(const void* in, const void* wt, void* out, float* stats, const int* plan_dev, const int* meta, int Nimg, int in_pix, int Ci, int in_pitch, int Co, int out_pix, int out_pitch, int wt_taps, const void* y, const float* zeros, const float* ones, float gscale, hipStream_t stream);
int svsr_conv3x3_c64_dgrad_bn(const void* in, const void* wt, void* out, const void* addend, float* stats, int Nimg, int H, int W, const int* dy, const int* dx, const int* tw, const void* y, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int act, const int* pixtab, hipStream_t stream);
int svsr_bn_bwd_from_stats(const void* g, const void* x, const float* mean, const float* rstd, const float* gamma, const float* stats, int nrows, float* coef, float* dgamma, float* dbeta, void* dx, int64_t npix, int C, hipStream_t stream);


/* svsr_conv3x3_wgrad: weight gradient of a 3x3 / stride-1 / pad-1 Conv2d (resnet.py:8-10) with all nine taps sharing one
 * pass over x [Nimg][H][W][Ci] and dy [Nimg][H][W][Co] (zero-padded coordinates, wgrad3x3.hip).  dw fp32 [Co][9][Ci] is
 * ACCUMULATED; split-K slabs go through `part` (size from svsr_conv3x3_wgrad_plan) and are added in a fixed order.
 * Requires Ci, Co multiples of 64 and W <= 29; other shapes go through svsr_igemm_wgrad. */
int svsr_conv3x3_wgrad_plan(int Nimg, int H, int W, int Ci, int Co, int* splits, int64_t* part_floats);
int svsr_conv3x3_wgrad(const void* x, const void* dy, float* dw, int Nimg, int H, int W, int Ci, int Co, float* part, int64_t part_floats, hipStream_t stream);

/* ---- 3-D stem (stem.hip) --------------------------------------------------------------------------------------
 * svsr_stem_conv_fwd replaces stem3d[0] = nn.Conv3d(1,64,(5,7,7),(1,2,2),(2,3,3),bias=False) (lightning.py:50).
 * vid fp32 [B][1][T][H][W]; w fp32 [64][1][5][7][7]; out bf16 [B*T][H/2][W/2][64]; stats [rows][2][64] with rows =
 * svsr_stem_conv_fwd_stat_rows(B, T, H, W). */
int svsr_stem_conv_fwd_stat_rows(int B, int T, int H, int W);
int64_t svsr_stem_conv_fwd_ws_bytes(int B, int T, int H, int W);
int svsr_stem_conv_fwd(const float* vid, const float* w, void* out, float* stats, int B, int T, int H, int W, void* ws, int64_t ws_bytes, hipStream_t stream);

/* weight gradient of the stem conv (autograd of lightning.py:50); dw fp32 [64][245] accumulated (per-workgroup slabs in
 * `part`, size from svsr_stem_conv_wgrad_plan, added in a fixed order). */
int svsr_stem_conv_wgrad_plan(int B, int T, int H, int W, int* splits, int64_t* part_floats);
int svsr_stem_conv_wgrad(const float* vid, const void* dy, float* dw, int B, int T, int H, int W, int use_tr, float* part, int64_t part_floats, hipStream_t stream);

/* The apply pass of svsr_stem_bn_act_pool_bwd and svsr_stem_conv_wgrad as ONE pass (autograd of lightning.py:49-54's stem3d[1..3] into
 * stem3d[0].weight): the gradient of the convolution output is made tile by tile in LDS and contracted at once, never written.  Call
 * svsr_stem_bn_act_pool_bwd first with dx = NULL and xwin / gpool given (it then runs its reduce pass and the finalisation only and
 * leaves gpool and coef); x = the convolution output, amax / mean / rstd as there.  dw bit-identical to the two-launch form.
 * svsr_stem_bwd_wgrad_ok: 1 when the shape is covered (else use the two launches). */
int svsr_stem_bwd_wgrad_ok(int B, int T, int H, int W);
int svsr_stem_bwd_wgrad(const float* vid, const void* gpool, const void* amax, const void* x, const float* mean, const float* rstd, const float* coef, float* dw, int B, int T, int H, int W, float* part, int64_t part_floats, hipStream_t stream);

/* ---- BatchNorm / activation / pooling passes (norm_act.hip) ---------------------------------------------------
 * svsr_bn_finalize: train-mode statistics of nn.BatchNorm2d/3d (lightning.py:51; resnet.py:37,54,14): adds the nrows
 * partial rows part[nrows][2][C] written by the producing convolution in a fixed order (double accumulation), writes
 * mean/rstd, updates running_mean/var (momentum, unbiased var) and num_batches_tracked. */
int svsr_bn_finalize(const float* part, int nrows, int C, float count, float eps, float momentum, float* mean, float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, hipStream_t stream);

/* eval-mode statistics: mean = running_mean, rstd = rsqrt(running_var + eps). */
int svsr_bn_eval_prepare(const float* running_mean, const float* running_var, int C, float eps, float* mean, float* rstd, hipStream_t stream);

/* y = act(gamma*(x-mean)*rstd + beta [+ res]); act 0 none, 1 ReLU, 2 Swish (LRS backbones/modules/resnet.py:90-107,
 * transformer/convolution.py:69); any C % 8 == 0 up to 2048.  Replaces bn1/relu1, bn2/+residual/relu2 and the
 * downsample BatchNorm of BasicBlock.forward (resnet.py:59-72). */
int svsr_bn_act_fwd(const void* x, const void* res, void* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int64_t npix, int C, int act, hipStream_t stream);

/* backward of the above: dgamma/dbeta accumulated; dx (grad of the conv output) and optional dres (= masked dy).
 * slots: workspace of svsr_bn_act_bwd_rows(npix, C) rows of [2][C] floats (any contents); coef [3][C] scratch.  Swish recomputes its
 * pre-activation and therefore needs beta and the forward's residual input `res` (null if there was none). */
int svsr_bn_act_bwd_rows(int64_t npix, int C);
int svsr_bn_act_bwd(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, const float* gamma, float* slots, float* coef, float* dgamma, float* dbeta, void* dx, void* dres, int64_t npix, int C, int act, const float* beta, const void* res, hipStream_t stream);

/* stem3d[1..3]: BatchNorm3d -> activation -> MaxPool3d((1,3,3),(1,2,2),(0,1,1)) fused; act 1 = exact nn.GELU()
 * (LRW lightning.py:51-53), act 2 = Swish (LRS backbones/conv3d_extractor.py:30-36).
 * x [N][Hc][Wc][C] -> y [N][Hp][Wp][C], amax uint8 [N][Hp][Wp][C] = window-local argmax (first max wins). */
int svsr_stem_bn_act_pool_fwd(const void* x, void* y, void* amax, const float* mean, const float* rstd, const float* gamma, const float* beta, int N, int Hc, int Wc, int Hp, int Wp, int C, int act, void* xwin, hipStream_t stream);

/* backward of the fused stem pass: dx = gradient of the stem conv output; slots: workspace of
 * svsr_stem_bn_act_pool_bwd_rows(N, Hc, Wc, C) rows of [2][C] floats. */
int svsr_stem_bn_act_pool_bwd_rows(int N, int Hc, int Wc, int C);
/* xwin (forward: optional output, backward: optional input), gpool (backward: optional workspace), both bf16 [N][Hp][Wp][C] like y:
 * the convolution output at every window's arg-max, kept by the forward so that the backward's reduce pass streams pooled-size tensors
 * (one activation derivative per pooled output, g = dpool * act' written to gpool) instead of gathering from the 4x larger
 * convolution output, and its apply pass evaluates no activation derivative at all.  Null: the gather form. */
int svsr_stem_bn_act_pool_bwd(const void* dpool, const void* amax, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, float* slots, float* coef, float* dgamma, float* dbeta, void* dx, int N, int Hc, int Wc, int Hp, int Wp, int C, int act, const void* xwin, void* gpool, hipStream_t stream);

/* hidden.mean((2,3)) (lightning.py:118) and its backward: [N][HW][C] <-> [N][C]. */
int svsr_avgpool_fwd(const void* x, void* y, int64_t N, int HW, int C, hipStream_t stream);
int svsr_avgpool_bwd(const void* dy, void* dx, int64_t N, int HW, int C, hipStream_t stream);

/* ---- transformer encoder passes (bert.hip) --------------------------------------------------------------------
 * y = LayerNorm(a + r) (BertSelfOutput / BertOutput, reached from lightning.py:152-156; LRS transformer/layer_norm.py);
 * r may be null; any D % 8 == 0 up to 2048.  The backward returns ds = dLN/d(a+r) (+ addend: the skip-path gradient of a
 * pre-LN residual block, encoder_layer.py:93-137); dgamma/dbeta are accumulated from per-workgroup partial rows in `part`
 * (svsr_add_ln_bwd_rows(R) rows of [2][D] floats). */
int svsr_add_ln_fwd(const void* a, const void* r, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int R, int D, float eps, hipStream_t stream);
int svsr_add_ln_bwd_rows(int R);
int svsr_add_ln_bwd(const void* dy, const void* a, const void* r, const float* gamma, const float* mean, const float* rstd, void* ds, float* dgamma, float* dbeta, int R, int D, const void* addend, float* part, hipStream_t stream);
/* svsr_add_ln_bwd / svsr_bias_act_bwd WITHOUT their fixed-order reduction: the partial rows stay in `part` (a buffer of the caller's that must live until
 * it has added them with svsr_colsum_rows on a stream of its choice — parameter-gradient sums the backward chain never waits for) */
int svsr_add_ln_bwd_partials(const void* dy, const void* a, const void* r, const float* gamma, const float* mean, const float* rstd, void* ds, int R, int D, const void* addend, float* part, hipStream_t stream);
/* svsr_add_ln_bwd_partials with a second output for the residual branch that ends in this sum (x' = x + alpha * dropout(branch), the LRS
 * encoder / decoder layers, encoder_layer.py:93-150): ds2 = alpha2 * mask / (1 - p) * ds, computed from the bf16-rounded ds with the element
 * indices svsr_scale_bf16 uses, i.e. bit for bit what svsr_scale_bf16(ds) would write (ds2 null: exactly svsr_add_ln_bwd_partials) */
int svsr_add_ln_bwd_branch(const void* dy, const void* a, const void* r, const float* gamma, const float* mean, const float* rstd, void* ds, int R, int D, const void* addend, float* part, void* ds2, float alpha2, const unsigned* drop_seed, unsigned drop_site, float drop_p, hipStream_t stream);
int svsr_bias_act_bwd_partials(const void* dy, const void* z, void* dz, float* db, int R, int N, int n_valid, int ld, int act, float gscale, float* part, hipStream_t stream);

/* BertEmbeddings on inputs_embeds = emb_dropout(cat(cls_token, feats)) (lightning.py:149-156):
 * y = dropout_out(LN(dropout_in(e) + pos[s] + type[0])); dropout_in = the module's emb_dropout (site_in, p_in), dropout_out =
 * BertEmbeddings' hidden dropout (site_out, p_out); both off when drop_seed is null or p = 0 (mask semantics: svsr_scale_bf16,
 * element index = position in the [B*S][D] tensor).  feats bf16 [B][S-1][D]; sum_out bf16 [B*S][D] keeps the pre-norm sum for the
 * backward.  svsr_embed_bwd_scatter takes ds = gradient of that sum and regenerates the dropout_in mask for dfeats / dcls
 * (part: [S][D] float workspace). */
int svsr_embed_ln_fwd(const void* feats, const float* cls, const float* pos, const float* type0, const float* gamma, const float* beta, void* sum_out, void* y, float* mean, float* rstd, int B, int S, int D, float eps, const unsigned* drop_seed, unsigned site_in, float p_in, unsigned site_out, float p_out, hipStream_t stream);
int svsr_embed_bwd_scatter(const void* ds, void* dfeats, float* dcls, float* dpos, float* dtype0, int B, int S, int D, float* part, const unsigned* drop_seed, unsigned site_in, float p_in, hipStream_t stream);

/* ---- `model.bert.type: x-transformers` encoder passes (csrc/xt.hip) ----------------------------------------------------
 * The reference instantiates x_transformers.Encoder(dim, depth, heads, attn_dropout, layer_dropout, ff_dropout, use_rmsnorm,
 * ff_glu, rotary_pos_emb) (lightning.py:93-105) — a third-party package absent from the reference tree, so these restate its
 * published algorithm (parity unpinned).  Activations are bf16 [R][ld] with ld = D rounded up to 64 and pad columns zero.
 *
 * RMSNorm: y = x / max(||x||_2 * D^-0.5, eps) * g; inv [R] keeps 1 / max(...) for the backward.  The backward returns
 * dx (+ addend, the gradient that bypasses the block through the residual connection) and accumulates dg from
 * svsr_rmsnorm_bwd_rows(R) partial rows of [ld] floats in `part`. */
int svsr_rmsnorm_fwd(const void* x, const float* g, void* y, float* inv, int R, int D, int ld, float eps, hipStream_t stream);
int svsr_rmsnorm_bwd_rows(int R);
int svsr_rmsnorm_bwd(const void* dy, const void* x, const float* g, const float* inv, const void* addend, void* dx, float* dg, float* part, int R, int D, int ld, hipStream_t stream);

/* Rotary embedding, in place, on the first 32 dims of `heads_total` consecutive 64-wide heads per row (q | k | v of the fused
 * projection); position = row % S; tab fp32 [S][32] = 16 cosines then 16 sines of position * 10000^(-j/16); sign -1 = backward. */
int svsr_rotary(void* qkv, const float* tab, int R, int S, int heads_total, int ld, int sign, hipStream_t stream);

/* GEGLU: u = [value | gate] bf16 [R][ldu] (halves I wide, I % 4 == 0); y [R][ldy] = dropout(value * gelu(gate)), zero for columns
 * >= I.  Backward: du [R][ldu] from dy [R][ldy] (mask regenerated), zero for columns >= 2I.  Dropout index = r * ldy + column. */
int svsr_geglu_fwd(const void* u, void* y, int R, int I, int ldu, int ldy, const unsigned* drop_seed, unsigned drop_site, float drop_p, hipStream_t stream);
int svsr_geglu_bwd(const void* dy, const void* u, void* du, int R, int I, int ldu, int ldy, const unsigned* drop_seed, unsigned drop_site, float drop_p, hipStream_t stream);

/* x0 = emb_dropout(cat(cls_token, cat(feats, word_mask[..., None], -1), 1)) (lightning.py:145-150): feats bf16 [B][S-1][F],
 * wmask fp32 [B][S-1] (D = F + 1) or null (D = F), cls fp32 [>= D], x0 bf16 [B*S][ld].  The backward writes dfeats bf16
 * [B][S-1][F] and adds the batch sum of row 0 into dcls (batch order fixed).  Dropout index = row * ld + column. */
int svsr_xt_embed_fwd(const void* feats, const float* wmask, const float* cls, void* x0, int B, int S, int F, int D, int ld, const unsigned* drop_seed, unsigned drop_site, float drop_p, hipStream_t stream);
int svsr_xt_embed_bwd(const void* dx0, void* dfeats, float* dcls, int B, int S, int F, int D, int ld, const unsigned* drop_seed, unsigned drop_site, float drop_p, hipStream_t stream);

/* dz = dy * act'(z) when z != null: act 1 = GELU from the saved pre-activation (BertIntermediate), act 2 = ReLU from the
 * saved output (PositionwiseFeedForward; gscale = 1/(1-p) when that output went through dropout — dropped elements are
 * exactly the zeros of the saved output); db[n] += column sums (bias gradient of any nn.Linear; db may be null), via
 * svsr_bias_act_bwd_rows(R, N) partial rows of [N] floats in `part` (0 rows: a single row slab adds into db directly). */
int svsr_bias_act_bwd_rows(int R, int N);
int svsr_bias_act_bwd(const void* dy, const void* z, void* dz, float* db, int R, int N, int n_valid, int ld, int act, float gscale, float* part, hipStream_t stream);

/* ---- losses / metric / optimiser (loss_optim.hip) --------------------------------------------------------------
 * F.cross_entropy(logits.float(), target, label_smoothing) mean over R rows (lightning.py:163-165,171): exactly one of
 * target_idx (int64 [R]) / target_prob (fp32 [R][V], pitch ldt) is non-null.  *loss = mean loss (row losses land in
 * row_loss [R] and are added in a fixed order).  A target index outside [0, V) yields NaN (torch raises a device assert). */
int svsr_ce_fwd(const void* logits, int logits_f32, int ld, const int64_t* target_idx, const float* target_prob, int ldt, int R, int V, float smoothing, float* loss, float* lse, float* row_loss, hipStream_t stream);
/* svsr_ce_bwd: dlogits bf16 [R][ldo] = gout / R * d loss / d logits; columns V .. ldo - 1 of every row are written as zeros. */
int svsr_ce_bwd(const void* logits, int logits_f32, int ld, const int64_t* target_idx, const float* target_prob, int ldt, int R, int V, float smoothing, const float* lse, const float* gout, void* dlogits, int ldo, hipStream_t stream);

/* ---- the SyncVSR audio-token head (audio_head.hip) ----------------------------------------------------------------------
 * svsr_linear_ce_fwd: loss = F.cross_entropy((h W^T + bias).reshape(-1, V), tok) of lightning.py:168-171 as ONE contraction whose
 * accumulators never leave the registers: h bf16 rows of pitch K (row r = hidden row (r / seq_T) * seq_S + seq_s0 + r % seq_T, or row r
 * when seq_T = 0), W bf16 [G*V][K] (nn.Linear layout), bias fp32 [G*V] or null, tok int64 [R*G] (row r, group g -> tok[r*G + g]: the
 * reference's logits.reshape(-1, V) order).  Writes lse [R*G], row_loss [R*G] (workspace) and *loss = their mean (fixed order).  A target
 * outside [0, V) yields NaN.  svsr_linear_ce_ok: 1 when the shape is taken (V a multiple of 320 — both codecs —, K a multiple of 64 up to 576).
 * svsr_linear_ce_bwd: the contraction again, then dlogits [R][G*V] bf16 = gout[0] / (R*G) * (softmax - onehot) for the projection's
 * data- and weight-gradient launches (svsr_igemm_fwd on the transposed weight, svsr_igemm_wgrad): no logits tensor is ever stored. */
int svsr_linear_ce_ok(int R, int K, int G, int V);
int svsr_linear_ce_fwd(const void* h, const void* w, const float* bias, const int64_t* tok, int R, int K, int G, int V, int seq_S, int seq_s0, int seq_T, float* loss, float* lse, float* row_loss, hipStream_t stream);
int svsr_linear_ce_bwd(const void* h, const void* w, const float* bias, const int64_t* tok, int R, int K, int G, int V, int seq_S, int seq_s0, int seq_T, const float* lse, const float* gout, void* dlogits, hipStream_t stream);

/* top-1 / top-5 accuracy (lightning.py:177-183); out2 = {top1, top5}; rows2: [B][2] float workspace. */
int svsr_topk_acc(const float* logits, const int64_t* labels, const float* soft_labels, int B, int C, float* out2, float* rows2, hipStream_t stream);

/* clip_grad_norm_ + AdamW + HF cosine-with-warm-up (lightning.py:216-223; Lightning gradient_clip_val).
 * opt_state: device struct {int step; int skipped; float lr_last; float gnorm_last; float part[1024]} (4112 bytes),
 * zero-initialised; svsr_grad_sumsq writes the 1024 partial sums of squares, svsr_adamw_step adds them in a fixed order.
 * A step whose gradient norm is NOT FINITE is skipped: parameters, moments and `step` stay as they are, `skipped` counts it and
 * gnorm_last shows the value (the reference would train on with NaN parameters; a poisoned fused-encoder launch — svsr_enc_gave_up —
 * is the case this exists for). */
int svsr_grad_sumsq(const float* g, int64_t n, void* opt_state, hipStream_t stream);
/* one range of the gradient into partial sums [part0, part0 + nparts): a step's ranges cover the buffer and the 1024 partials once each
 * (the clip of lightning's `gradient_clip_val` needs the whole norm; most of it can be summed before the last weight gradient is done) */
int svsr_grad_sumsq_parts(const float* g, int64_t n, void* opt_state, int part0, int nparts, hipStream_t stream);
int svsr_adamw_step(float* p, const float* g, float* m, float* v, void* shadow, int64_t n, int64_t decay_end, float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm, int warmup, int total_steps, void* opt_state, hipStream_t stream);
/* the same over one RANGE of the flat buffers (pointers offset by the caller, decay_end relative to the range); the step counter advances only
 * with advance != 0 — the last range of a step.  Lets a step update what the next forward needs first and the rest on another stream. */
int svsr_adamw_range(float* p, const float* g, float* m, float* v, void* shadow, int64_t n, int64_t decay_end, float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm, int warmup, int total_steps, void* opt_state, int advance, hipStream_t stream);

/* bf16 shadows of fp32 parameters: plain cast, and a table-driven [A][T][B] -> [B][T][Apad] transpose-cast.
 * table: device array of {int64 src_off, dst_off; int32 A, T, Bd, Apad} (elements). */
int svsr_cast_bf16(const float* src, void* dst, int64_t n, hipStream_t stream);
int svsr_transpose_cast_multi(const float* src, void* dst, const void* table, int n_entries, hipStream_t stream);
/* the same from the bf16 shadow (src16: bf16 copy of the parameter buffer, same offsets): half the bytes read, identical result */
int svsr_transpose_bf16_multi(const void* src16, void* dst, const void* table, int n_entries, hipStream_t stream);
int svsr_fill_f32(float* p, int64_t n, float v, hipStream_t stream);
/* scalars on the device, so that a step never depends on a host value: *word += delta (the dropout seed word, advanced once per
 * training forward: lightning.py:150 draws fresh masks every step);  *out = *a + wb * *b  (loss_total = loss_category +
 * lambda_audio * loss_audio, lightning.py:187). */
/* ---- fused encoder forward (enc_fused.hip) -----------------------------------------------------------------------
 * svsr_enc_fwd replaces the forward of `n_layers` (<= 8 per call) consecutive HF BertLayers of the word-level model's encoder
 * (reference LRW/video/src/lightning.py:92,152-156: BertSelfAttention, BertSelfOutput, BertIntermediate, BertOutput) for width 512,
 * 8 heads of 64, FFN 2048 and sequences of S <= 32 rows: ONE launch per 32 sequences instead of seven per layer.  Eight workgroups
 * per sequence (one per head / column eighth) exchange ctx, ao, hg and f through memory between arrival counters (write-through
 * stores, relaxed agent-scope counter, L1-bypassing loads; all eight must be resident together, which the 32-sequence launches
 * guarantee on 256 CUs).  x0 bf16 [B*S][512]; `layers`: HOST array of records (copied by value into the launch); every tensor a
 * layer writes has the contents the unfused launches (svsr_igemm_fwd, svsr_mha_fwd, svsr_add_ln_fwd) give it, dropout masks included.
 * ws: svsr_enc_fwd_ws_bytes(B) bytes of device workspace (arrival counters, zeroed here; word B = error flag, non-zero if a bounded
 * wait gave up; the layer records). */
/* A bounded cluster wait that gives up is LOUD: the launch's error word (word B of ws) is set, the giving-up workgroup overwrites its slice
 * of the launch's final output with NaN (the step's loss / gradient norm turn NaN without a host synchronisation), and a sticky
 * process-wide flag is set that svsr_enc_gave_up(reset) returns (1 / 0; it synchronises: call it where the host waits anyway;
 * -1 if the flag cannot be read).  svsr_enc_gave_up_peek() reads a copy of the flag that the giving-up workgroup stores into pinned host
 * memory: no synchronisation, so engine.TrainStep looks at it before every step and, when set, re-routes the encoder to the per-layer launch
 * chain for the rest of the run (the poisoned step itself is skipped by svsr_adamw_step's non-finite guard).  A launch holds at most
 * svsr_device_cus() / 8 sequences. */
int svsr_enc_gave_up(int reset);
int svsr_enc_gave_up_peek(void);
int svsr_debug_enc_spin_limit(unsigned limit);     /* test aid: polls before a cluster wait gives up (0 = default 2^20; 1 provokes the give-up path) */
int64_t svsr_enc_fwd_ws_bytes(int B);
int svsr_debug_enc_trace(int64_t* out, int n);     /* debug: out == null arms s_memtime stamps of workgroup 0 at the phase boundaries of the next launches; else copies n stamps out */
int svsr_enc_fwd(const void* x0, const svsr_enc_layer* layers, int n_layers, int B, int S, float ln_eps, const unsigned* drop_seed, float p_hidden, float p_attn, void* ws, int64_t ws_bytes, hipStream_t stream);

/* Backward of the same layers in one launch per 32 sequences (enc_fused.hip, k_enc_bwd): replaces, per layer, two svsr_add_ln_bwd, four
 * svsr_igemm_fwd data-gradient launches, svsr_bias_act_bwd, svsr_mha_bwd and the dropout re-scalings between them.  dy bf16 [B*S][512] =
 * gradient of the last layer's output; layers = HOST array in forward order.  Per layer it writes ds2 / df / dx1 / ds1 / dao / dx bf16
 * [R][512], dz bf16 [R][2048], dqkv bf16 [R][1536] — (hg, df), (x1, dz), (ctx, dao), (xin, dqkv) are the operand pairs of the layer's four weight
 * gradients; df may alias ds2 and dao ds1 without hidden dropout — and part1 / part2 fp32 [B][2][512], one row of {sum dy*xhat | sum dy} per
 * sequence and LayerNorm (svsr_colsum_rows over B rows of 1024 gives gamma | beta gradients).  layers[0].dx = gradient of the first input. */
int64_t svsr_enc_bwd_ws_bytes(int B);
int svsr_enc_bwd(const void* dy, const svsr_enc_bwd_layer* layers, int n_layers, int B, int S, const unsigned* drop_seed, float p_hidden, float p_attn, void* ws, int64_t ws_bytes, hipStream_t stream);

int svsr_word_add(int* word, int delta, hipStream_t stream);
int svsr_lincomb2(const float* a, const float* b, float wb, float* out, hipStream_t stream);
/* out0 = (wa*a + wb*b) + wc*c with one rounding per product and per sum (what `mtlalpha * loss_ctc + (1 - mtlalpha) * loss_att + w * loss_audio` gives
 * in torch, reference LRS e2e_asr_transformer.py:217-224), and out1 = num / den (the token accuracy) when out1 is not null: 0-d device values */
int svsr_lincomb3_ratio(const float* a, float wa, const float* b, float wb, const float* c, float wc, float* out0, const float* num, const float* den, float* out1, hipStream_t stream);

/* Device-side input pipeline (reference LRW/video/src/data.py:150,157-171: x/255 -> RandomHorizontalFlip ->
 * RandomResizedCrop | CenterCrop -> Normalize(0.421, 0.165)): stored uint8 clips [B][T][Hs][Ws] -> fp32 model input
 * [B][1][T][H][W].  params: device int32 [B][5] = {top, left, h, w, flip} chosen by the host per clip; the window is resized to
 * H x W with bilinear sampling (align_corners = False, no antialias; a window of exactly H x W is a plain crop). */
int svsr_clip_prep(const void* src_u8, const int* params, float* dst, int B, int T, int Hs, int Ws, int H, int W, float mean, float std, hipStream_t stream);

/* ---- LRS (E2E: Conformer encoder + CTC / attention decoder) -----------------------------------------------------
 * Multi-head attention with 64-wide heads (mha.hip).  q rows [B*Lq] with pitch q_pitch (head h at column h*64), k/v rows
 * [B*Lk] with pitch kv_pitch; klen[b] = number of valid keys (null: all), causal != 0 masks j > i.  With pe != null the
 * Conformer's relative-position scores are used: ((q+u)·k_j + (q+v)·pe[Lq-1+j-i]) * scale, pe = linear_pos(pos_emb)
 * [2*Lq-1][pe_pitch] (reference LRS/video/espnet/nets/pytorch_backend/transformer/attention.py:191-278 incl. rel_shift
 * :216-236; plain MHA :38-108; mask semantics :71-78).  probs [B*H][Lq][ldp] bf16 (before dropout) is kept for the backward;
 * attention dropout (attention.py:80) uses the keep decision hash(*drop_seed, drop_site, index into probs). */
int svsr_mha_fwd(const void* q, int q_pitch, const void* k, const void* v, int kv_pitch, const void* pe, int pe_pitch, const float* bias_u, const float* bias_v, const int* klen, int causal, int B, int H, int dh, int Lq, int Lk, int ldp, float scale, void* ctx, int ctx_pitch, void* probs, const unsigned* drop_seed, unsigned drop_site, float drop_p, hipStream_t stream);

/* backward of svsr_mha_fwd: ds [B*H][Lq][ldp] workspace (score gradients); dq/dk/dv written (not accumulated); for the
 * relative-position form also dq_ac / dq_bd (the two summands of dq, whose column sums are the pos_bias_u / pos_bias_v
 * gradients) and dpe [2*Lq-1][dpe_pitch] (gradient of the projected position table, feeds linear_pos's weight gradient);
 * pe_part: fp32 workspace [B][2*Lq-1][dpe_pitch] (per-batch-item partials of dpe; dpe_pitch must equal H*64). */
int svsr_mha_bwd(const void* dctx, int dctx_pitch, const void* q, int q_pitch, const void* k, const void* v, int kv_pitch, const void* pe, int pe_pitch, const float* bias_u, const float* bias_v, const void* probs, void* ds, int B, int H, int dh, int Lq, int Lk, int ldp, float scale, void* dq, int dq_pitch, void* dq_ac, void* dq_bd, int aux_pitch, void* dk, void* dv, int dkv_pitch, void* dpe, int dpe_pitch, float* pe_part, const unsigned* drop_seed, unsigned drop_site, float drop_p, hipStream_t stream);

/* The same attention without the probability matrix on the way forward (mha_flash.h; attention.py:38-108,191-278 as above): keys streamed in
 * blocks of 32 with an online softmax, one wave per 32-query tile, up to eight tiles of one (clip, head) per workgroup.  The forward keeps
 * lse [B*H][Lq] fp32 (log-sum-exp of a query's scaled, masked scores; +inf when every key is masked) instead of probs; dropout decisions are
 * the same hash(*drop_seed, drop_site, (bh * Lq + i) * ldp + j) as svsr_mha_fwd's.  The backward recomputes P from lse and takes
 * rowsum(P o dP) as dctx_i . ctx_i (ctx: the forward's output); its query pass writes probs / ds [B*H][Lq][ldp] bf16 (ldp % 8 == 0) as
 * WORKSPACE for the key and position-table passes, which are svsr_mha_bwd's.  ws: svsr_mha_flash_ws_bytes(H, Lq) bytes (the transposed
 * position table; rel-pos only).  klen / causal must be the forward's. */
int64_t svsr_mha_flash_ws_bytes(int H, int Lq);
int svsr_mha_flash_fwd(const void* q, int q_pitch, const void* k, const void* v, int kv_pitch, const void* pe, int pe_pitch, const float* bias_u, const float* bias_v, const int* klen, int causal, int B, int H, int dh, int Lq, int Lk, int ldp, float scale, void* ctx, int ctx_pitch, float* lse, const unsigned* drop_seed, unsigned drop_site, float drop_p, hipStream_t stream);
int svsr_mha_flash_bwd(const void* dctx, int dctx_pitch, const void* ctx, int ctx_pitch, const float* lse, const void* q, int q_pitch, const void* k, const void* v, int kv_pitch, const void* pe, int pe_pitch, const float* bias_u, const float* bias_v, const int* klen, int causal, void* probs, void* ds, int B, int H, int dh, int Lq, int Lk, int ldp, float scale, void* dq, int dq_pitch, void* dq_ac, void* dq_bd, int aux_pitch, void* dk, void* dv, int dkv_pitch, void* dpe, int dpe_pitch, float* pe_part, void* ws, int64_t ws_bytes, const unsigned* drop_seed, unsigned drop_site, float drop_p, hipStream_t stream);
/* the same in two parts (parts bit 0: query + key passes — dq, dq_ac, dq_bd, dk, dv; bit 1: the position-table pass — dpe, which only the weight
 * gradient of linear_pos reads, so it may be issued on another stream behind bit 0's launches; 3 = svsr_mha_flash_bwd; bit 2, with bit 0: ws already
 * holds the transposed position table, made by svsr_mha_pe_transpose — it depends on pe alone, so the forward or another stream can make it) */
int svsr_mha_pe_transpose(const void* pe, int pe_pitch, int H, int Lq, void* ws, int64_t ws_bytes, hipStream_t stream);
int svsr_mha_flash_bwd_parts(const void* dctx, int dctx_pitch, const void* ctx, int ctx_pitch, const float* lse, const void* q, int q_pitch, const void* k, const void* v, int kv_pitch, const void* pe, int pe_pitch, const float* bias_u, const float* bias_v, const int* klen, int causal, void* probs, void* ds, int B, int H, int dh, int Lq, int Lk, int ldp, float scale, void* dq, int dq_pitch, void* dq_ac, void* dq_bd, int aux_pitch, void* dk, void* dv, int dkv_pitch, void* dpe, int dpe_pitch, float* pe_part, void* ws, int64_t ws_bytes, const unsigned* drop_seed, unsigned drop_site, float drop_p, int parts, hipStream_t stream);

/* Conformer convolution module core (transformer/convolution.py:56-75): u [B*T][2D] = pointwise_cov1 output ->
 * GLU -> depthwise Conv1d(K odd <= 31, pad (K-1)/2, weight [D][K], bias) -> c [B*T][D] bf16 + BatchNorm1d partial sums
 * stats[rows][2][D], rows = svsr_glu_dwconv_fwd_stat_rows(B, T) (finalised by svsr_bn_finalize; BN+Swish itself is svsr_bn_act_fwd act 2). */
int svsr_glu_dwconv_fwd_stat_rows(int B, int T);
int svsr_glu_dwconv_fwd(const void* u, const float* w, const float* bias, void* c, float* stats, int B, int T, int D, int K, hipStream_t stream);

/* backward: dc = gradient of c -> du [B*T][2D]; dw [D][K] and dbias [D] accumulated.  part: fp32 workspace of
 * nsplit * D * (K+1) floats. */
int svsr_glu_dwconv_bwd(const void* dc, const void* u, const float* w, void* du, float* dw, float* dbias, float* part, int nsplit, int B, int T, int D, int K, hipStream_t stream);
/* the same in two parts (bit 0: du and the partial rows; bit 1: the fixed-order sum of the rows into dw / dbias, which nothing in the backward
 * chain waits for — another stream may take it; part must then be a buffer of its own until that launch is done) */
int svsr_glu_dwconv_bwd_parts(const void* dc, const void* u, const float* w, void* du, float* dw, float* dbias, float* part, int nsplit, int B, int T, int D, int K, int parts, hipStream_t stream);

/* CTC loss as the reference calls it (ctc.py:44-74,83-151): log_softmax over V, torch.nn.CTCLoss(reduction="sum",
 * zero_infinity=True), blank 0, divided by the batch size.  logits fp32 [B*T][ld]; labels int64 [B][Lmax] padded with -1
 * at the tail; ilen int32 [B].  Workspaces: lse [B*T], ab [B][T][2*Lmax+1], nll [B].  *loss = sum_b nll_b / B. */
int svsr_ctc_fwd(const float* logits, int ld, const int64_t* labels, int Lmax, const int* ilen, int B, int T, int V, float* lse, float* ab, float* nll, float* loss, hipStream_t stream);

/* dlogits [B*T][ldo] bf16 = gout/B * (softmax - label occupancy) for t < ilen[b], 0 elsewhere (and for infeasible targets). */
int svsr_ctc_grad(const float* logits, int ld, const int64_t* labels, int Lmax, const int* ilen, int B, int T, int V, const float* lse, const float* ab, const float* nll, const float* gout, void* dlogits, int ldo, hipStream_t stream);

/* CTC prefix scores of beam-search extensions (espnet/nets/ctc_prefix_score.py:11-165 `CTCPrefixScoreTH.__call__`, driven by
 * scorers/ctc.py:87-127 from LRS/video/lightning.py:237-279).  logp fp32 [T][ldp] = log-softmax of ctc_lo over the clip;
 * r_prev fp32 [n][T][2] = (non-blank, blank) forward log-probabilities of each hypothesis' prefix; last int64 [n] = last label
 * of each prefix; ids int64 [n][S] = candidate labels per hypothesis (null: all V labels, S = V); out_len = labels in the
 * prefixes without <sos>.  Writes r_new fp32 [n][S][T][2] (state of every extension) and psi fp32 [n][S] (log prefix
 * probability; eos -> total probability of the prefix, blank -> -1e10). */
int svsr_ctc_prefix_score(const float* logp, int ldp, const float* r_prev, const int64_t* last, const int64_t* ids, float* r_new, float* psi, int T, int V, int n, int S, int out_len, int blank, int eos, hipStream_t stream);

/* Decoder input: x[r] = emb[tok[r]] * scale + pe[r % L]  (torch.nn.Embedding + PositionalEncoding, decoder.py:80-84,
 * embedding.py:78-89); backward scatter-adds scale * dx into demb. */
/* add_sos_eos (reference add_sos_eos.py:10-31, e2e_asr_transformer.py:203-215) + the CTC label form, one launch: label [B][L] int64 with
 * ignore_id padding (dropped wherever it sits in a row, as the reference's `y[y != ignore_id]`) -> labels [B][L] (live tokens, -1 padded),
 * ys_in / ys_out [B][L+1] (sos = eos; ys_out padded with ignore_id).  A live token outside [1, odim) — torch's Embedding / CTCLoss stop on it
 * with a device-side assert — is replaced by eos and reported through svsr_lrs_target_errors (nothing traps). */
int svsr_lrs_targets(const int64_t* label, int B, int L, int odim, int64_t ignore_id, int64_t eos, int64_t* labels, int64_t* ys_in, int64_t* ys_out, hipStream_t stream);
/* sticky error word of svsr_lrs_targets: 1 if a label outside [1, odim) was met since the last reset (it was replaced by eos so that no
 * later kernel leaves its tables; ignore_id entries anywhere in a row are dropped as the reference's add_sos_eos drops them), else 0; -1 if
 * the word cannot be read.  Synchronises. */
int svsr_lrs_target_errors(int reset);
int svsr_embed_pos_fwd(const int64_t* tok, const float* emb, const float* pe, void* x, int R, int L, int D, float scale, hipStream_t stream);
int svsr_embed_pos_bwd(const int64_t* tok, const void* dx, float* demb, int R, int D, float scale, hipStream_t stream);

/* ESPnet LabelSmoothingLoss (label_smoothing_loss.py:41-63): KL(true || softmax(logits)) summed over rows whose target
 * is not -1, times inv_denom (1/batch, or 1/#tokens with length normalisation); true = 1-smoothing at the target and
 * smoothing/(V-1) elsewhere.  counts[0] = rows whose argmax equals the target, counts[1] = live rows (th_accuracy,
 * nets_utils.py:303-323).  logits fp32 [R][ld]; lse [R] saved for the backward; rows3: [R][3] float workspace. */
int svsr_ls_loss_fwd(const float* logits, int ld, const int64_t* target, int R, int V, float smoothing, float inv_denom, float* loss, float* lse, float* counts, float* rows3, hipStream_t stream);
int svsr_ls_loss_bwd(const float* logits, int ld, const int64_t* target, int R, int V, float smoothing, float inv_denom, const float* lse, const float* gout, void* dlogits, int ldo, hipStream_t stream);

/* y = alpha * dropout_p(x) over n (multiple of 8) contiguous bf16 elements (y may alias x).  Dropout everywhere in this
 * library is counter based: element i of a tensor is kept iff mix(i * 2654435761 + key) >= p * 2^32 with
 * key = mix(*drop_seed * 0x9E3779B9 + drop_site * 0x7F4A7C15 + 0x165667B1) and mix = the murmur3 finaliser; kept values are
 * scaled by 1/(1-p).  The backward passes regenerate the mask from (seed, site); drop_seed is a device word the caller
 * advances once per step.  drop_seed == null or p == 0 disables it. */
int svsr_scale_bf16(const void* x, void* y, int64_t n, float alpha, const unsigned* drop_seed, unsigned drop_site, float drop_p, hipStream_t stream);

/* ---- native step enqueuer (steplist.hip; host code, launches nothing of its own) -----------------------------------
 * Stands where the reference's per-step host loop stands (pl.Trainer.fit -> training_step, LRW/video/src/train.py:23-45,
 * lightning.py:194-202): ONE host call per optimisation step instead of one per launch.  A list records, once, the launch
 * sequence of a training step: CALL = one stream-taking entry point of this header with its arguments frozen (arguments are
 * passed as 64-bit slots: integers sign-extended, floats as the bits of a double, pointers as they are), WAIT = `waiter`
 * waits for everything enqueued so far on `signaller`, MEMSET = hipMemsetAsync, BREAK = segment boundary (the host may issue
 * a collective between two segments).  svsr_steplist_run(list, k, &failed) re-issues segment k (k < 0: every segment) on the
 * recorded streams and returns 0 or the first failing call's code with its op index in *failed.  Every buffer a recorded call
 * points to (device and host) must stay alive and in place while the list exists.  svsr_steplist_knows(name): 1 if the
 * entry point can be recorded.  svsr_stream_wait / svsr_memset_async are the eager twins of WAIT / MEMSET. */
void* svsr_steplist_create(void);
int svsr_steplist_destroy(void* list);
int svsr_steplist_knows(const char* name);
int svsr_steplist_push_call(void* list, const char* name, const int64_t* slots, int nslots);
int svsr_steplist_push_wait(void* list, hipStream_t waiter, hipStream_t signaller);
int svsr_steplist_push_memset(void* list, void* ptr, int value, int64_t bytes, hipStream_t stream);
int svsr_steplist_push_break(void* list);
int svsr_steplist_segments(void* list);
int64_t svsr_steplist_size(void* list);
int svsr_steplist_run(void* list, int segment, int* failed);
int svsr_stream_wait(hipStream_t waiter, hipStream_t signaller);
int svsr_memset_async(void* ptr, int value, int64_t bytes, hipStream_t stream);

/* Compute units of the current device (persistent kernels size their grids, static tile lists and cluster counts by it). */
int svsr_device_cus(void);

/* Measurement aid: the effective shader clock.  `blocks` workgroups run iters x 4 MFMAs per wave and write {shader cycles (s_memtime),
 * 100 MHz ticks (s_memrealtime)} of that block to out[b][2] (device int64, blocks * 2 + 1 words): MHz = 100 * sum(cycles) / sum(ticks).
 * bench.py reads it before and after its sustained leg (the `sustained` object of its line). */
int svsr_clock_probe(int64_t* out, int blocks, int iters, hipStream_t stream);

/* Test aid (csrc/runtime.hip): a foreign resident kernel — `workgroups` workgroups x 256 threads, each holding lds_bytes of LDS, that sleep
 * until svsr_debug_occupy_stop() (or ~4 s).  Stands in for a peer-waiting collective kernel beside a training step of the reference's
 * strategy="ddp" loop (LRW/video/src/train.py:28): tests/test_gpu_cotenant.py runs steps beside it. */
int svsr_debug_occupy_start(int workgroups, int lds_bytes, hipStream_t stream);
int svsr_debug_occupy_stop(void);

#ifdef __cplusplus
}
#endif
#endif /* SYNCVSR_HIP_H */
