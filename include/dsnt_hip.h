/*
 * dsnt_hip.h — C ABI of libdsnt_hip.so: the MI355X (gfx950) device path of the
 * dsnt-pose2d hot path.
 *
 * The reference (anibali/dsnt-pose2d) has NO FFI layer: its device work is whatever
 * PyTorch 0.3.1 + cuDNN 7 execute underneath pure-Python modules.
 * Each entry point below therefore cites the reference *call site* whose device
 * work it replaces (paths relative to /root/reference/src/dsnt/).
 *
 * Conventions
 *  - plain pointers + sizes; every pointer is DEVICE memory owned by the caller
 *    (PyTorch's allocator); the library allocates nothing and the product entry points
 *    declared here keep no state except a thread-local error string (the calibration /
 *    timeline diagnostics live in dsnt_hip_debug.h, are not part of this ABI and are the
 *    only code with process-wide switches);
 *  - all arithmetic is fp32; activations are NHWC ([N][H][W][C], C innermost);
 *    conv weights are OHWI ([Cout][R][S][Cin]); heat-maps for the DSNT head are
 *    planar rows ([rows = N*J][H*W]);
 *  - kernels are enqueued on `stream` (a hipStream_t passed as void*) and never
 *    synchronise, so calls can be captured into a hipGraph;
 *  - return value 0 = OK, otherwise a dsnt_status; dsnt_last_error() describes it.
 *    No C++ exception crosses this boundary.
 */
#ifndef DSNT_HIP_H
#define DSNT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    DSNT_OK = 0,
    DSNT_ERR_SHAPE = 1,       /* unsupported / inconsistent shape */
    DSNT_ERR_ALIGN = 2,       /* pointer or channel count not 16-byte compatible */
    DSNT_ERR_ARG = 3,         /* null pointer / bad enum */
    DSNT_ERR_HIP = 4          /* hipGetLastError() after launch */
} dsnt_status;

/* What a launch leaves behind for the CONSUMERS of its output besides the output itself: fp16x3 operand bounds.
 * amax (may be NULL): a 64-float bound as dsnt_amax leaves it; the launch RAISES it to the max |value| it writes to its
 *         output (atomic max on the bit patterns of non-negative floats: the caller zeroes the 64 slots once per step) — the
 *         fp16x3 operand bound of a consumer that reads the output without a BatchNorm in between (skip projections, `lin`
 *         convolutions: hourglass.py:45-48,120-135), or of a data gradient written by a convolution epilogue.  With a
 *         bn_bwd_epilogue: max |dz| (what dsnt_bn_bwd_finalize_bound turns into the bound of a BatchNorm backward folded
 *         into dsnt_conv1x1_bwd_f16x3).
 * amax_bn (may be NULL): the same for max |relu?(value * amax_scale[c] + amax_shift[c])| — the operand a consumer with
 *         an EVAL-mode BatchNorm(+ReLU) prologue will form from this output (inference.py:38-48: the vectors come from
 *         running statistics, so they exist before the producer runs); amax_scale / amax_shift: [C], 16-byte aligned.
 * (Until round 3 this struct — then dsnt_out_bounds — also described a BatchNorm finalisation run by the launch's last
 * workgroup; slower than the separate dsnt_bn_finalize launches in every measurement of three rounds, removed in round 4.) */
typedef struct {
    float* amax;
    float* amax_bn; const float* amax_scale; const float* amax_shift; int amax_relu, reserved;
} dsnt_out_bounds;

/* BatchNorm finalisation folded into the prologue of the launch that CONSUMES the BatchNorm (the low-resolution
 * hourglass levels, /root/reference/src/dsnt/hourglass.py:33-43 between two 10-us convolutions: a separate
 * dsnt_bn_finalize launch costs the dependency chain ~8 us there, ~120 of them 1.0 ms of an hg2 step).  Every
 * workgroup sums partial[tiles][2][C] itself (fp64, fixed order), writes mean / invstd / scale / shift — which the
 * launch then reads as its in_scale / in_shift and backward reads later — and workgroup 0 moves the running
 * statistics.  Limits: C <= 256, tiles * C <= 16384 (dsnt_conv_fwd_pro_ok). */
typedef struct dsnt_bn_prologue {
    const float* partial; int tiles, C;
    int64_t M;                                   /* rows the sums run over */
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;     /* may be NULL */
    float momentum, eps;
    float* mean; float* invstd; float* scale; float* shift;
} dsnt_bn_prologue;

int dsnt_version(void);
const char* dsnt_last_error(void);

/* ------------------------------------------------------------------ DSNT head
 * rows = product of leading dims (N*J); each row is one H x W map, contiguous. */

/* model.py:24-45 `_hm_preact`: mode 0 softmax (F.softmax over H*W), 1 thresholded
 * softmax (nn.py:119-139; threshold, eps), 2 abs, 3 relu, 4 sigmoid (each / (sum+eps)). */
int dsnt_preact_fwd(const float* x, float* y, int64_t rows, int hw, int mode,
                    float threshold, float eps, void* stream);
/* backward of the same (nn.py:131-139 for the softmax forms: y*(g - sum(y*g))). */
int dsnt_preact_bwd(const float* x, const float* y, const float* gy, float* gx,
                    int64_t rows, int hw, int mode, float threshold, float eps, void* stream);

/* nn.py:25-78 `generate_xy` + `expectation_2d` + `dsnt`: coords[row] = (E[x], E[y]). */
int dsnt_expect_fwd(const float* hm, float* coords, int64_t rows, int h, int w, void* stream);
/* its backward: ghm[row][i] = gcoords.x * X_i + gcoords.y * Y_i. */
int dsnt_expect_bwd(const float* gcoords, float* ghm, int64_t rows, int h, int w, void* stream);

/* nn.py:168-205 `make_gauss`: out[row] = normalised Gaussian at coords[row], sigma. */
int dsnt_make_gauss(const float* coords, float* out, int64_t rows, int h, int w, float sigma,
                    void* stream);
/* its backward (autograd through nn.py:180-203; the reference's make_gauss is differentiable in `coords`):
 * g_coords[row] = sum_i g_out[row][i] * d out[row][i] / d coords[row]  (closed form, Gaussian re-evaluated). */
int dsnt_make_gauss_bwd(const float* coords, const float* g_out, float* g_coords, int64_t rows, int h, int w,
                        float sigma, void* stream);

/* 'fc' output strategy: out_fc = nn.Linear(H*W, 2) on the flattened heat-maps (model.py:222-223, 293-303, 196-198).
 * out[rows][2] = hm[rows][hw] . W[2][hw]^T + b[2];  backward: ghm (may be NULL), gW[2][hw], gb[2] (may be NULL),
 * overwritten, rows summed in order (deterministic). */
int dsnt_fc2_fwd(const float* hm, const float* w, const float* b, float* out, int64_t rows, int hw, void* stream);
int dsnt_fc2_bwd(const float* g, const float* hm, const float* w, float* ghm, float* gw, float* gb, int64_t rows,
                 int hw, void* stream);

/* nn.py:208-298 regularisers, per-row value before masked_average.
 * kind: 0 js, 1 kl, 2 mse, 3 var.  target = mu_t [rows][2]. */
int dsnt_reg_fwd(const float* hm, const float* target, float* per_row, int64_t rows, int h, int w,
                 float sigma, int kind, void* stream);
/* ghm[row] = g_row[row] * d(per_row)/d(hm). */
int dsnt_reg_bwd(const float* hm, const float* target, const float* g_row, float* ghm,
                 int64_t rows, int h, int w, float sigma, int kind, void* stream);
/* gmu[row][2] = g_row[row] * d(per_row)/d(target[row]) for kind 0..2: the reference's target Gaussian is
 * make_gauss(mu_t, ...) inside autograd (/root/reference/src/dsnt/nn.py:219-271), so its regularisers are differentiable
 * in the target means; the derivative through the target pixel is composed with make_gauss's backward in registers. */
int dsnt_reg_bwd_mu(const float* hm, const float* target, const float* g_row, float* gmu,
                    int64_t rows, int h, int w, float sigma, int kind, void* stream);

/* nn.py:97-116 `euclidean_loss` (per-point part): dist[i] = ||a_i - t_i||_2, d dims. */
int dsnt_euclid_fwd(const float* actual, const float* target, float* dist, int64_t n, int d,
                    void* stream);
/* ga = g_dist * (a - t) / dist  (un-guarded at dist == 0, like the reference). */
int dsnt_euclid_bwd(const float* actual, const float* target, const float* dist,
                    const float* g_dist, float* g_actual, int64_t n, int d, void* stream);

/* nn.py:81-94 `masked_average`: out[0] = sum(l*m)/clamp(sum m, 1); out[1] = that denominator.
 * mask may be NULL (plain mean with max(numel,1)). */
int dsnt_masked_avg_fwd(const float* losses, const float* mask, float* out2, int64_t n,
                        void* stream);
/* g_losses[i] = g_out[0] * m_i / denom  (denom read from out2[1]). */
int dsnt_masked_avg_bwd(const float* g_out, const float* mask, const float* out2,
                        float* g_losses, int64_t n, void* stream);

/* Fused head, forward: model.py:288-291 (softmax preact + dsnt) in one pass over the logits.
 * Writes normalised heat-maps and coords. */
int dsnt_head_fwd(const float* logits, float* hm, float* coords, int64_t rows, int h, int w,
                  void* stream);
/* Fused head, loss: model.py:233-246 — per-row Euclidean distance and regulariser value
 * (reg_kind -1 = none) from the saved heat-maps; reductions by dsnt_masked_avg_fwd. */
int dsnt_head_loss_rows(const float* hm, const float* coords, const float* target,
                        float* dist, float* reg_row, int64_t rows, int h, int w, float sigma,
                        int reg_kind, void* stream);
/* Fused head, loss AND its gradient in one pass over the saved heat-maps (the train step; model.py:233-246,
 * train.py:358,381): per row the Euclidean distance and the regulariser value (as dsnt_head_loss_rows) and
 *   g_logits[row] = d( w_row (dist + reg_coeff reg) ) / d logits,  w_row = mask[row] / clamp(sum mask, 1)
 * (mask NULL: 1 / max(rows, 1) — nn.py:81-94), i.e. the gradient of this stack's loss for an upstream gradient of 1;
 * dsnt_scale_by_scalar applies another upstream value.  denom2 = the 2 floats dsnt_mask_denom left ({sum mask, its
 * clamp}: once per step, every stack shares the mask).  The head of a train step is then 4 HBM passes per stack
 * instead of 5.  H * W <= 4096. */
int dsnt_mask_denom(const float* mask, float* denom2, int64_t n, void* stream);
int dsnt_head_loss_grad(const float* hm, const float* coords, const float* target, const float* mask,
                        const float* denom2, float* dist, float* reg_row, float* g_logits, int64_t rows, int h, int w,
                        float sigma, int reg_kind, float reg_coeff, void* stream);
/* loss[0] = masked_average(dist) + reg_coeff * masked_average(reg_row) (reg_row NULL: no regulariser) in one launch;
 * e2 = {masked_average(dist), denominator} as dsnt_masked_avg_fwd leaves it. */
int dsnt_head_loss_reduce(const float* dist, const float* reg_row, const float* mask, const float* denom2,
                          float reg_coeff, float* loss, float* e2, int64_t rows, void* stream);
/* x[0..n) *= s[0] (device scalar); returns at once when s[0] == 1. */
int dsnt_scale_by_scalar(float* x, const float* s, int64_t n, void* stream);
/* Fused head, backward: d loss / d logits in one pass, given per-row upstream factors
 * g_dist[row] (for the Euclidean term) and g_reg[row] (for the regulariser), i.e.
 * model.py:233-246 + nn.py:66-78 + softmax backward collapsed. */
int dsnt_head_bwd(const float* hm, const float* coords, const float* target, const float* dist,
                  const float* g_dist, const float* g_reg, float* g_logits, int64_t rows,
                  int h, int w, float sigma, int reg_kind, void* stream);

/* --------------------------------------------------------------- convolutions
 * Implicit-GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32).  Replaces nn.Conv2d forward
 * and its autograd (hourglass.py:20-25,104-105,120-123,135-136,148; model.py:129). */
typedef struct {
    int N, H, W, Cin;          /* input  [N][H][W][Cin]   (Cin % 4 == 0) */
    int Ho, Wo, Cout;          /* output [N][Ho][Wo][Cout] */
    int R, S;                  /* filter taps */
    int stride, pad, dil;
} dsnt_conv_geom;

/* y = conv(act(x)) + bias + res1 + res2,  act(x) = relu?(x*in_scale[c] + in_shift[c]) when
 * in_scale != NULL (the BatchNorm2d+ReLU that precedes the conv in a pre-activation
 * Bottleneck, hourglass.py:33-43, folded into the operand load; zero padding is applied
 * AFTER the activation, as F.conv2d pads the activated tensor).
 * bias/res1/res2 may be NULL.  res* have y's shape; res1 may alias y (in-place accumulate).
 * stats_partial (may be NULL): [ceil(M/128)][2][Cout] per-tile column sums and sums of
 * squares of y — the batch statistics of the next BatchNorm (hourglass.py:19,21,24). */
int dsnt_conv_fwd(const float* x, const float* w, const float* bias, float* y,
                  const float* in_scale, const float* in_shift, int in_relu,
                  const float* res1, const float* res2, float* stats_partial,
                  const dsnt_conv_geom* g, void* stream);

/* Data-gradient launches can fuse the first half of the BatchNorm+ReLU backward into the epilogue:
 * with `bnb` != NULL the kernel writes dz = (conv result) * [relu mask of bn(x)] to y and per-tile
 * (sum dz, sum dz*xhat) to stats_partial (xhat = (x - mean) * invstd), ready for
 * dsnt_bn_bwd_finalize; dsnt_bn_act_bwd_apply then runs with relu = 0 on dz. */
typedef struct {
    const float* x;            /* the BatchNorm input, shape of y */
    const float* scale; const float* shift; const float* mean; const float* invstd;   /* [C] */
    int relu;
} dsnt_bn_bwd_epilogue;
int dsnt_conv_fwd_ex(const float* x, const float* w, const float* bias, float* y,
                     const float* in_scale, const float* in_shift, int in_relu,
                     const float* res1, const float* res2, float* stats_partial,
                     const dsnt_conv_geom* g, const dsnt_bn_bwd_epilogue* bnb, const dsnt_out_bounds* tail, void* stream);
/* (`tail`, here and in the other _ex variants: the operand bounds the launch leaves behind — dsnt_out_bounds.) */

/* dsnt_conv_fwd_ex with the BatchNorm of its A operand finalised in the launch's prologue (fp32-MFMA path: the
 * small-M kernels); in_scale / in_shift are pro->scale / pro->shift.  dsnt_conv_fwd_pro_ok(g, tiles, C) != 0 if the
 * limits hold for this geometry. */
int dsnt_conv_fwd_pro_ok(const dsnt_conv_geom* g, int tiles, int C);
int dsnt_conv_fwd_pro(const float* x, const float* w, const float* bias, float* y, const dsnt_bn_prologue* pro,
                      int in_relu, const float* res1, const float* res2, float* stats_partial,
                      const dsnt_conv_geom* g, const dsnt_out_bounds* tail, void* stream);

/* Rows per stats_partial tile that dsnt_conv_fwd uses for this geometry (128 or 32): the
 * caller sizes stats_partial as [ceil(M/bm)][2][Cout] and hands ceil(M/bm) to dsnt_bn_finalize. */
int dsnt_conv_fwd_bm(const dsnt_conv_geom* g);

/* bf16x6 variant: fp32-accurate convolution on the bf16 matrix cores.  Operands are split exactly
 * into three bf16 planes and the six product terms >= 2^-16 are accumulated in fp32 (error below
 * one fp32 rounding; 6/16 of the fp32-MFMA cost).  `w_planes` points at plane 0 of the OHWI weights
 * inside a dsnt_split_bf16x3 output; planes are `plane_stride` bf16 elements apart (a multiple of 8;
 * the whole parameter arena is split with one launch).  Every other argument as dsnt_conv_fwd.  Uses 128-row statistics
 * tiles.  dsnt_conv_bf16x6_ok(g) != 0 tells whether the geometry is supported. */
int dsnt_conv_bf16x6_ok(const dsnt_conv_geom* g);
int dsnt_split_bf16x3(const float* src, void* dst_planes, int64_t n, void* stream);
int dsnt_conv_fwd_bf16x6(const float* x, const void* w_planes, int64_t plane_stride, const float* bias, float* y,
                         const float* in_scale, const float* in_shift, int in_relu,
                         const float* res1, const float* res2, float* stats_partial,
                         const dsnt_conv_geom* g, void* stream);

int dsnt_conv_fwd_bf16x6_ex(const float* x, const void* w_planes, int64_t plane_stride, const float* bias,
                            float* y, const float* in_scale, const float* in_shift, int in_relu,
                            const float* res1, const float* res2, float* stats_partial,
                            const dsnt_conv_geom* g, const dsnt_bn_bwd_epilogue* bnb, const dsnt_out_bounds* tail,
                            void* stream);

/* fp16x3 variant: x * s = h1 + h2 on TWO fp16 planes after a power-of-two scale, three MFMAs per product (error vs
 * fp64 below a plain fp32 GEMM's: conv.hip, tools/split_numerics.py).  a_bound / w_bound are DEVICE scalars >= the
 * largest |value| of the A operand AFTER its BN+ReLU prologue (or of x when there is none) and of the weights; the
 * kernels derive the scales from them (pow2: bound * scale in [2^13, 2^14)), so a loose bound is fine and a bound
 * that is too SMALL overflows fp16.  w_planes: two fp16 planes made by dsnt_split_f16x2 with the SAME w_bound.
 * Same arguments otherwise as dsnt_conv_fwd_bf16x6_ex.
 * A "device scalar" bound is DSNT_BOUND_SLOTS = 64 floats whose MAXIMUM is the bound (producers that find it with
 * atomics spread them over the slots; the others write the same value 64 times). */
int dsnt_conv_fwd_f16x3_ex(const float* x, const void* w_planes, int64_t plane_stride, const float* w_bound,
                           const float* a_bound, const float* bias, float* y, const float* in_scale,
                           const float* in_shift, int in_relu, const float* res1, const float* res2,
                           float* stats_partial, const dsnt_conv_geom* g, const dsnt_bn_bwd_epilogue* bnb,
                           const dsnt_out_bounds* tail, void* stream);
/* The same call for a 3x3 / stride 1 / pad 1 convolution whose weight planes are in the STREAM layout of
 * dsnt_f16_prep_weights (row flag): the symmetric persistent kernel of csrc/conv3s.hip (cuDNN's 3x3 forward / data gradient of
 * /root/reference/src/dsnt/hourglass.py:22-23).  Needs dsnt_conv_fwd_stream_ok(g) (H % 4 == 0 and W % 32 == 0, or H % 8 == 0 and
 * W % 16 == 0; Cin % 32 == 0 and <= 128, Cout 64 or 128, tensors < 2 GiB) and res2 == NULL; DSNT_ERR_SHAPE otherwise.  The
 * convolution sums are bit-identical to dsnt_conv_fwd_f16x3_ex's; stats_partial rows are one per 128-pixel patch (4 x 32, or 8 x 16 when W % 32 != 0): [N * H * W / 128][2][Cout].
 * in_relu (this entry point and dsnt_conv_fwd_f16x3_ex): bit 0 = ReLU in the prologue; bit 1 (DSNT_CONV_SHARE_CHIP): the launch runs on
 * a stream of its own beside other work — the persistent kernels (3x3 stream kernel, streaming 1x1 kernel) then start fewer
 * workgroups (3/2 per CU; half of the CUs), so that CUs with free LDS remain for the other streams.  Results unchanged. */
#define DSNT_CONV_SHARE_CHIP 2
int dsnt_conv_fwd_f16x3_stream(const float* x, const void* w_planes, int64_t plane_stride, const float* w_bound,
                               const float* a_bound, const float* bias, float* y, const float* in_scale,
                               const float* in_shift, int in_relu, const float* res1, const float* res2,
                               float* stats_partial, const dsnt_conv_geom* g, const dsnt_bn_bwd_epilogue* bnb,
                               const dsnt_out_bounds* tail, void* stream);
int dsnt_conv_fwd_stream_ok(const dsnt_conv_geom* g);
/* Introspection (tests' launch census): the form of the csrc/conv3s.hip kernel a stream launch of `mode` takes on geometry g with
 * this process's tuning — mode 0 / 1: forward or plain data gradient (without / with res1), 3: with bnb, 4:
 * dsnt_conv_dgrad_f16x3_stream_apply.  Bit 0: 128 output columns as two 64-column halves per patch (few patches), bit 1: the
 * v_mfma_f32_16x16x32_f16 form with the LDS-DMA weight ring, bit 2: 8 x 16 pixel patches (W % 32 != 0).  -1: not a stream
 * geometry.  (/root/reference/src/dsnt/hourglass.py:22-23,36-40 is what every form computes.)  dsnt_version() >= 113. */
int dsnt_conv_fwd_stream_form(const dsnt_conv_geom* g, int mode);
/* out[0..63] = bound slots whose maximum is max |src[i]|; dst = two fp16 planes (plane_stride elements apart) of src * pow2(bound). */
int dsnt_amax(const float* src, int64_t n, float* out, void* stream);
int dsnt_split_f16x2(const float* src, void* dst, int64_t n, int64_t plane_stride, const float* bound, void* stream);
/* The same for many tensors per launch (device tables of int64 rows):
 * dsnt_f16_prep_weights: row of `row_ints` values {src float*, dst fp16 plane-0*, bound float*, count (% 4 == 0), plane stride
 *   (elements), stream Cout, stream Cin} (row_ints == 5: the last two are absent and read as 0; dsnt_version() >= 110): bound = max|src|, dst = the two fp16 planes of src * pow2(bound) — every conv's weights once
 *   per step; stream Cout > 0: src is an OHWI 3x3 filter [Cout][3][3][Cin] (Cin % 16 == 0) and each plane is written in
 *   STREAM order [Cin / 16][9 taps][Cout][16] for dsnt_conv_fwd_f16x3_stream (0, 0: element order kept);
 * dsnt_f16_prep_bn_bounds: row {gamma float*, beta float*, out float*, C, bits of float sqrt(M)}:
 *   out = max_c(|gamma_c| sqrt(M) + |beta_c|), an upper bound of |relu?(bn(x))| for a TRAIN-mode BatchNorm over M samples. */
int dsnt_f16_prep_weights(const int64_t* table, int rows, int row_ints, void* stream);   /* row_ints: 7, or 5 (no stream columns) */
int dsnt_f16_prep_bn_bounds(const int64_t* table, int rows, void* stream);

/* Re-pack OHWI weights for the data-gradient pass: wd[Cin][R][S][Cout] with taps flipped,
 * so that dgrad(dy) == dsnt_conv_fwd(dy, wd) with pad' = dil*(R-1) - pad (stride 1). */
int dsnt_conv_pack_dgrad(const float* w, float* wd, int Cout, int R, int S, int Cin,
                         void* stream);

/* The same for every convolution of a backward pass in one launch.  table[nconv][6] (device, int32) =
 * {src offset into params, dst offset, Cout, R, S, Cin}; writes fp32 into out[dst..] and the exact
 * bf16x3 split into planes (plane stride = total elements). */
int dsnt_conv_pack_dgrad_all(const int* table, int nconv, const float* params, float* out,
                             void* planes, int64_t total, void* stream);

/* Weight / bias gradient: dw[Cout][R][S][Cin] = sum_m act(x)[m, (r,s,c)] * dy[m, cout],
 * dbias[cout] = sum_m dy.  `ws` is caller workspace of dsnt_conv_wgrad_ws_floats() floats (enough for ANY of the
 * weight-gradient entry points on that geometry; dsnt_conv_wgrad_f16x3_ws_floats is the exact size of that call).
 * accumulate: bit 0 adds into dw/dbias instead of overwriting; bit 1 (DSNT_WGRAD_SHARE_CHIP, split-precision kernels):
 * the launch shares the chip with a dependency chain on another stream (loss.backward() of bin/train.py:380: weight
 * gradients feed nothing downstream) and keeps to ONE workgroup per CU, so that the chain's small kernels find a slot
 * instead of waiting for the whole weight gradient to retire.
 * dw == NULL (and dbias == NULL): only the split-M partial slabs are written to `ws`; the caller keeps `ws`
 * alive and reduces later with dsnt_wgrad_reduce_all (one launch for many convolutions). */
#define DSNT_WGRAD_SHARE_CHIP 2
/* bit 2, with DSNT_WGRAD_SHARE_CHIP, 3x3 halo kernel: half as many slabs / workgroups again (128 four-wave workgroups) — for
 * launches in the middle of backward, where the stream they run on has slack and the dependency chain beside them does not */
#define DSNT_WGRAD_NARROW 4
int64_t dsnt_conv_wgrad_ws_floats(const dsnt_conv_geom* g);
int dsnt_conv_wgrad(const float* x, const float* in_scale, const float* in_shift, int in_relu,
                    const float* dy, float* ws, float* dw, float* dbias, int accumulate,
                    const dsnt_conv_geom* g, void* stream);

/* bf16x6 variant of the weight gradient (same arguments and workspace; both operands are split into
 * bf16 planes while they are staged).  dsnt_conv_wgrad_bf16x6_ok(g) != 0 if supported. */
int dsnt_conv_wgrad_bf16x6_ok(const dsnt_conv_geom* g);
int dsnt_conv_wgrad_bf16x6(const float* x, const float* in_scale, const float* in_shift, int in_relu,
                           const float* dy, float* ws, float* dw, float* dbias, int accumulate,
                           const dsnt_conv_geom* g, void* stream);

/* Deferred reduction of the slabs of `rows` convolutions in one launch.  table[rows][7] (device, int64) =
 * {ws pointer, dw pointer, dbias pointer or 0, splits, Cout*R*S*Cin, Cout, accumulate} with `splits` =
 * dsnt_conv_wgrad_splits(g); max_blocks >= ceil((Cout*K/4 + ceil(Cout/4)) / 64) of the largest row.
 * Same summation order as the reduction inside dsnt_conv_wgrad (bit-identical results). */
int dsnt_conv_wgrad_splits(const dsnt_conv_geom* g);
int dsnt_wgrad_reduce_all(const int64_t* table, int rows, int max_blocks, void* stream);

/* Grouped weight gradients: many (small) convolutions in ONE launch.  dsnt_conv_wgrad_desc writes an opaque
 * descriptor of dsnt_conv_wgrad_desc_bytes() bytes (host memory) for one bf16x6 weight gradient with a
 * BN(+ReLU) prologue (same arguments as dsnt_conv_wgrad_bf16x6 with dw == NULL: slabs only) and returns its
 * number of workgroups (> 0) or -error.  The caller copies the descriptors back to back into device memory
 * and calls dsnt_conv_wgrad_group(table, nconv, max_blocks >= the largest workgroup count); x, dy, scale/shift
 * and ws must stay untouched until then; reduce afterwards with dsnt_wgrad_reduce_all.  Bit-identical to the
 * per-convolution launches.  Replaces the per-layer autograd weight-gradient calls of
 * /root/reference/src/dsnt/bin/train.py:380 (loss.backward()) for the low-resolution hourglass levels. */
int dsnt_conv_wgrad_desc_bytes(void);
int dsnt_conv_wgrad_desc(const float* x, const float* in_scale, const float* in_shift, int in_relu,
                         const float* dy, float* ws, const dsnt_conv_geom* g, void* desc_out);
int dsnt_conv_wgrad_group(const void* table, int nconv, int max_blocks, void* stream);

/* fp16x3 weight gradient (two fp16 planes per operand, three MFMAs per product, scales from the 64-slot bounds
 * a_bound >= max|act(x)| and g_bound >= max|dy|; the slabs are written already unscaled).  Same arguments otherwise as
 * dsnt_conv_wgrad_bf16x6; dsnt_conv_wgrad_desc_f16x3 is the descriptor form for dsnt_conv_wgrad_group (descriptors of
 * both kinds may share one table). */
int dsnt_conv_wgrad_f16x3(const float* x, const float* in_scale, const float* in_shift, int in_relu,
                          const float* dy, float* ws, float* dw, float* dbias, int accumulate,
                          const float* a_bound, const float* g_bound, const dsnt_conv_geom* g, void* stream);
int dsnt_conv_wgrad_desc_f16x3(const float* x, const float* in_scale, const float* in_shift, int in_relu,
                               const float* dy, float* ws, const float* a_bound, const float* g_bound,
                               const dsnt_conv_geom* g, void* desc_out);
/* 3x3 / stride 1 / pad 1 convolutions with Cin % 64 == 0, Cout % 64 == 0 and W in {16, 32, 64, 128, ...}
 * (dsnt_conv_wgrad_halo_ok != 0: conv2 of every Bottleneck, /root/reference/src/dsnt/hourglass.py:22-23) run
 * dsnt_conv_wgrad_f16x3 as a HALO kernel: a workgroup owns 64 input channels x all nine taps x 128 output channels of a
 * strip of pixel rows, so each operand element is BatchNorm-transformed and split once instead of nine times.  It cuts
 * the pixels into its own slabs: size `ws` with dsnt_conv_wgrad_f16x3_ws_floats and reduce
 * dsnt_conv_wgrad_f16x3_splits slabs, both asked with the `accumulate` flags of the launch (a DSNT_WGRAD_SHARE_CHIP
 * launch runs as four-wave workgroups, one per CU, over half as many slabs; both fall back to the plain plan for every
 * other geometry). */
int dsnt_conv_wgrad_halo_ok(const dsnt_conv_geom* g);
int dsnt_conv_wgrad_f16x3_splits(const dsnt_conv_geom* g, int accumulate);
int64_t dsnt_conv_wgrad_f16x3_ws_floats(const dsnt_conv_geom* g, int accumulate);

/* Forward of a 1x1 / stride 1 convolution with >= 4096 rows on the second streaming kernel (csrc/fwd1.hip): the same contract as
 * dsnt_conv_fwd_f16x3_ex for these shapes (BN+ReLU prologue or a raw operand, bias, ONE residual, operand bounds, `tail`) except
 * for the statistics: stats_partial is [rows][2][Cout] with rows = dsnt_conv1x1_fwd_stats_rows(g, in_relu) — one row per
 * WORKGROUP (<= 512) instead of one per 128 pixels; hand `rows` to dsnt_bn_finalize as its ntiles.  The activations are
 * loaded as whole rows (16 bytes per lane) and staged through LDS once for all output columns, where the first streaming
 * kernel (gemm1.hip, behind dsnt_conv_fwd_f16x3_ex) loads a row per lane.  in_relu: bit 0 = ReLU, bit 1 = DSNT_CONV_SHARE_CHIP.
 * dsnt_conv1x1_fwd_ok(g) != 0: (Cin, Cout) in {(256, 128), (128, 256), (128, 128), (64, 64), (64, 128), (256, 256)},
 * N*H*W % 32 == 0 and >= 16384.  Replaces nn.Conv2d 1x1 forward + the BatchNorm2d + ReLU in front of it
 * (/root/reference/src/dsnt/hourglass.py:20,25,33-43,44-48,146-153). */
int dsnt_conv1x1_fwd_ok(const dsnt_conv_geom* g);
int dsnt_conv1x1_fwd_stats_rows(const dsnt_conv_geom* g, int in_relu);
int dsnt_conv1x1_fwd_f16x3(const float* x, const void* w_planes, int64_t plane_stride, const float* w_bound,
                           const float* a_bound, const float* bias, float* y, const float* in_scale,
                           const float* in_shift, int in_relu, const float* res1, float* stats_partial,
                           const dsnt_conv_geom* g, const dsnt_out_bounds* tail, void* stream);
/* The stem (hourglass.py:157 `self.conv1`, 7x7 / stride 2 / pad 3 on the image) in its space-to-depth form: a 4x4 / stride 1 /
 * pad 1 convolution of the 16-channel image dsnt_s2d_input leaves ([N][H/2+1][W/2+1][16]), 64 output channels, fp16x3 — on a halo
 * kernel of its own (csrc/stem4.hip: the 7 x 35 input halo of a 4 x 32 output patch staged once for all 16 taps, weights in
 * registers).  dsnt_stem4_fwd_ok(g): 4x4, stride 1, pad 1, Cin 16, Cout 64, Ho % 4 == 0, Wo % 32 == 0.  Statistics: ONE row per
 * workgroup — dsnt_stem4_fwd_stats_rows(g) rows of [2][64], handed to dsnt_bn_finalize as ntiles.  w_planes: the plain
 * [Cout][4][4][16] planes of dsnt_s2d_weights_prep / dsnt_split_f16x2; tail: dsnt_out_bounds.amax only. */
int dsnt_stem4_fwd_ok(const dsnt_conv_geom* g);
int dsnt_stem4_fwd_stats_rows(const dsnt_conv_geom* g);
int dsnt_stem4_fwd_f16x3(const float* x, const void* w_planes, int64_t plane_stride, const float* w_bound,
                         const float* a_bound, const float* bias, float* y, float* stats_partial,
                         const dsnt_conv_geom* g, const dsnt_out_bounds* tail, void* stream);


/* The WHOLE backward of a 1x1 / stride 1 convolution y = conv(relu?(bn(x))) in one pass over its tensors — what
 * autograd runs as cuDNN backward-data + backward-filter of /root/reference/src/dsnt/hourglass.py:20,25 (conv1 / conv3 of
 * every Bottleneck) plus, with `ap`, the BatchNorm backward of the layer behind (hourglass.py:21,36-37):
 *   dz_out[m][c]  = (sum_n dY[m][n] W[n][c]) * [bn(x) > 0]     (the ReLU mask as dsnt_bn_bwd_epilogue defines it: xs)
 *   stats_partial = [splits][2][Cin] partial (sum dz_out, sum dz_out * xhat) for dsnt_bn_bwd_finalize (ntiles = splits)
 *   ws            = [splits][Cout][Cin] slabs of dW[n][c] = sum_m dY[m][n] act(x)[m][c], then [splits][Cout] bias partials
 *                   (reduce with dsnt_wgrad_reduce_all, `splits` = dsnt_conv1x1_bwd_splits(g))
 * dY is `dy` itself (ap == NULL) or is formed in registers from dy = dz of the BatchNorm that consumes y:
 *   dY = ap->scale * (dz - coef[0] - (y - mean) * invstd * coef[1])          (dsnt_bn_act_bwd_apply's arithmetic; that
 * launch and its 12 bytes per element disappear).  xs->x is read ONCE (mask, xhat and weight-gradient operand from the
 * same registers), dy (and ap->y) once: 402 MB instead of 737 MB for 256 -> 128 channels at 64 x 64, batch 32.
 * wd_planes: the data-gradient weights [Cin][Cout] (dsnt_conv_pack_dgrad) as two fp16 planes with bound w_bound
 * (dsnt_f16_prep_weights); a_bound >= max|act(x)|, g_bound >= max|dY| (64-slot bounds; with `ap`:
 * dsnt_bn_bwd_finalize_bound leaves it); dz_amax (may be NULL): raised to max|dz_out|.  flags: DSNT_CONV_SHARE_CHIP = the launch
 * runs on a stream of its own beside other work and keeps to half of the CUs (`splits` / ws size are asked with the same flags).
 * A convolution WITHOUT a BatchNorm in front of it (the projection shortcuts hourglass.py:44-48, the `fc` convolutions :146-153):
 * xs->scale == NULL (then shift / mean / invstd NULL too, ap == NULL, stats_partial unused): act(x) = x, dz_out = dL/dx itself,
 * added to what dz_out holds when bit 0 of `flags` is set.
 * dsnt_conv1x1_bwd_ok(g) != 0: (Cout, Cin) in {(128, 256), (256, 128), (128, 128), (64, 64), (128, 64), (256, 256)},
 * N*H*W % 32 == 0 and >= 16384. */
typedef struct {
    const float* y;            /* the convolution's own output = the BatchNorm's input, [M][Cout] */
    const float* scale; const float* mean; const float* invstd;   /* [Cout] */
    const float* coef;         /* [2][Cout] as dsnt_bn_bwd_finalize leaves it */
} dsnt_bn_bwd_apply;
int dsnt_conv1x1_bwd_ok(const dsnt_conv_geom* g);
int dsnt_conv1x1_bwd_splits(const dsnt_conv_geom* g, int flags);
int64_t dsnt_conv1x1_bwd_ws_floats(const dsnt_conv_geom* g, int flags);
int dsnt_conv1x1_bwd_f16x3(const dsnt_bn_bwd_epilogue* xs, const float* dy, const dsnt_bn_bwd_apply* ap,
                           const void* wd_planes, int64_t plane_stride, const float* w_bound, const float* a_bound,
                           const float* g_bound, float* dz_out, float* stats_partial, float* ws, float* dz_amax,
                           int flags, const dsnt_conv_geom* g, void* stream);
/* Data gradient of a 3x3 / stride 1 / pad 1 convolution whose OUTPUT y feeds a train-mode BatchNorm (conv2 of a Bottleneck,
 * /root/reference/src/dsnt/hourglass.py:36-40: bn3 behind conv2), with that BatchNorm's backward folded into the operand load
 * (csrc/conv3s.hip MODE 4): replaces dsnt_bn_act_bwd_apply(dz, y, ...) -> dy followed by dsnt_conv_fwd_f16x3_stream(dy, ...).
 *   dz      [N,H,W,C]  dL/d relu(bn(y)), ReLU-masked, its two BatchNorm sums already reduced into ap->coef = [c0 | c1]
 *   ap      the BatchNorm behind the convolution: y, scale (= gamma invstd), mean, invstd, coef (struct below)
 *   dy_out  [N,H,W,C]  receives dL/dy = scale (dz - c0 - (y - mean) invstd c1) (every pixel once); must not alias dz
 *   a_bound a bound of |dL/dy| (dsnt_bn_bwd_finalize_bound); w_planes / plane_stride / w_bound: the data-gradient filter in
 *           STREAM layout, as for dsnt_conv_fwd_f16x3_stream
 *   dx_dz, stats_partial, bnb (required), tail: the launch's output — the gradient behind the BatchNorm IN FRONT of the
 *           convolution, masked and with its two sums per 128-pixel patch — exactly as dsnt_conv_fwd_f16x3_stream with bnb
 *   flags   DSNT_CONV_SHARE_CHIP or 0
 * Needs dsnt_conv_fwd_stream_ok(g); DSNT_ERR_SHAPE otherwise.  dsnt_version() >= 112. */
int dsnt_conv_dgrad_f16x3_stream_apply(const float* dz, const dsnt_bn_bwd_apply* ap, float* dy_out,
                                       const void* w_planes, int64_t plane_stride, const float* w_bound,
                                       const float* a_bound, float* dx_dz, float* stats_partial, int flags,
                                       const dsnt_conv_geom* g, const dsnt_bn_bwd_epilogue* bnb,
                                       const dsnt_out_bounds* tail, void* stream);

/* ----------------------------------------------------- heat-map matching ("gauss" output strategy)
 * Rows = (image, joint) maps of h x w floats, target = normalised coordinates [rows][2].
 * dsnt_encode_heatmaps: /root/reference/src/dsnt/util.py:129-147 (encode_heatmaps) + :70-126 (draw_gaussian with
 *   clip_size 7, unnormalised): pixel = round_half_even((c + 1) * size/2 - 0.5), bump exp(-d^2 / (2 sigma^2)) on
 *   the 7x7 window, nothing when the centre lies more than 3.5 px outside the map.
 * dsnt_heatmap_mse_fwd: per_row[r] = sum_hw (hm - encode(target))^2 without materialising the target; the caller
 *   sums the rows and divides by rows*h*w (nn.functional.mse_loss of model.py:156 / :256).
 * dsnt_heatmap_mse_bwd: dhm = gscale[0] * 2 / (rows*h*w) * (hm - encode(target)); gscale is a device scalar.
 * dsnt_decode_heatmaps: util.py:150-198 (get_preds + decode_heatmaps): first arg-max pixel ((0,0) when the
 *   maximum is not positive; y = index / h as the reference), optional quarter-pixel shift towards the larger
 *   neighbour, then (p + 0.5) * 2/size - 1.  coords [rows][2]. */
int dsnt_encode_heatmaps(const float* target, float* out, int64_t rows, int h, int w, float sigma, void* stream);
int dsnt_heatmap_mse_fwd(const float* hm, const float* target, float* per_row, int64_t rows, int h, int w,
                         float sigma, void* stream);
int dsnt_heatmap_mse_bwd(const float* hm, const float* target, const float* gscale, float* dhm, int64_t rows,
                         int h, int w, float sigma, void* stream);
int dsnt_decode_heatmaps(const float* hm, float* coords, int64_t rows, int h, int w, int use_neighbours,
                         void* stream);

/* ----------------------------------------------------- batch-norm, elementwise
 * x viewed as [M][C] (M = N*H*W), C % 4 == 0. */

/* Column sums / sums of squares per 128-row tile: partial[ceil(M/128)][2][C]
 * (same format as dsnt_conv_fwd's stats_partial). */
int dsnt_bn_stats(const float* x, float* partial, int64_t M, int C, void* stream);

/* nn.BatchNorm2d forward bookkeeping (hourglass.py:19,21,24,106,147; torch defaults):
 * training: mean/var(biased) from partials -> mean, invstd = rsqrt(var+eps);
 *           running_mean/var updated with `momentum` (unbiased var);
 * eval:     mean/invstd from the running statistics.
 * Always: scale = gamma*invstd, shift = beta - mean*scale. */
int dsnt_bn_finalize(const float* partial, int ntiles, int64_t M, int C,
                     const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps,
                     int training, float* mean, float* invstd, float* scale, float* shift,
                     void* stream);

/* y = relu?(x*scale + shift) materialised (stem: hourglass.py:158-159). */
int dsnt_bn_act_fwd(const float* x, const float* scale, const float* shift, int relu,
                    float* y, int64_t M, int C, void* stream);

/* Eval-mode (running statistics) BatchNorm vectors of many layers in ONE launch: table rows of int64
 * {gamma*, beta*, running_mean*, running_var*, mean*, invstd*, scale*, shift*, C, bits of float eps}; per row what
 * dsnt_bn_finalize(training = 0) writes (inference.py:33-48 runs the model in eval mode). */
int dsnt_bn_eval_prep(const int64_t* table, int rows, void* stream);
/* Backward of y = relu?(bn(x)) given da = dL/dy, in three steps:
 *  reduce:   partial[tile][0][c] = sum dz, partial[tile][1][c] = sum dz*xhat,
 *            dz = da * (y > 0), xhat = (x - mean)*invstd;
 *  finalize: dgamma (+)= sum dz*xhat, dbeta (+)= sum dz, coef[0][c] = mean(dz),
 *            coef[1][c] = mean(dz*xhat);
 *  apply:    dx (+)= gamma*invstd * (dz - coef0 - xhat*coef1).
 * `accumulate` of dsnt_bn_bwd_finalize: bit 0 = add to dgamma / dbeta; DSNT_BN_FROZEN = the forward ran on FIXED statistics
 * (nn.BatchNorm2d in eval mode, i.e. a backward through `model.eval()`'s forward): coef is written as zeros, so that the apply
 * gives dx = gamma*invstd * dz; dgamma / dbeta are the same two sums. */
#define DSNT_BN_FROZEN 2
int dsnt_bn_act_bwd_reduce(const float* da, const float* x, const float* scale,
                           const float* shift, const float* mean, const float* invstd,
                           int relu, float* partial, int64_t M, int C, void* stream);
/* ... of y = relu?(bn(x) + skip) (`out = relu(bn2(conv2(..)) + identity)`, the tail of a torchvision BasicBlock / Bottleneck): the ReLU
 * mask is taken from the stored y, dz = da * (y > 0) is written (the skip branch's gradient and the apply read it) and the same tile sums
 * are left — dsnt_relu_bwd + dsnt_bn_act_bwd_reduce in one pass. */
int dsnt_bn_add_act_bwd_reduce(const float* da, const float* y, const float* x, const float* mean, const float* invstd,
                               int relu, float* dz, float* partial, int64_t M, int C, void* stream);
int dsnt_bn_bwd_finalize(const float* partial, int ntiles, int64_t M, int C,
                         float* dgamma, float* dbeta, int accumulate, float* coef,
                         void* stream);
/* The same, and the bound of dx = scale (dz - coef0 - xhat coef1) for dsnt_conv1x1_bwd_f16x3, which forms dx in registers and
 * needs its fp16x3 scale beforehand: max_c |scale_c| (max|dz| + |coef0_c| + |coef1_c| sqrt(M)) raised into bound_out (64 slots,
 * zeroed by the caller once per step); scale = the BatchNorm's forward scale (gamma * invstd), dz_amax = the 64-slot max |dz| the
 * data-gradient launch left through dsnt_out_bounds.amax.  hourglass.py:21,36-37 (bn2 of a Bottleneck, backward). */
int dsnt_bn_bwd_finalize_bound(const float* partial, int ntiles, int64_t M, int C, float* dgamma, float* dbeta,
                               int accumulate, float* coef, const float* scale, const float* dz_amax, float* bound_out,
                               void* stream);
int dsnt_bn_act_bwd_apply(const float* da, const float* x, const float* scale,
                          const float* shift, const float* mean, const float* invstd,
                          const float* coef, int relu, float* dx, int accumulate,
                          int64_t M, int C, void* stream);
/* The same, additionally raising amax[0] (device scalar, zeroed by the caller at the start of the step: dsnt_fill_zero)
 * (64 bound slots, slot = workgroup index mod 64) to max |dx| with integer atomic maxima on the floats' bits (order-independent): the operand bound of the fp16x3
 * data / weight gradient kernels that consume dx. */
int dsnt_bn_act_bwd_apply_amax(const float* da, const float* x, const float* scale, const float* shift,
                               const float* mean, const float* invstd, const float* coef, int relu, float* dx,
                               int accumulate, int64_t M, int C, float* amax, void* stream);
/* dsnt_bn_act_bwd_apply(_amax) with dsnt_bn_bwd_finalize folded into its prologue (same limits as dsnt_bn_prologue):
 * partial[ntiles][2][C] = (sum dz, sum dz * xhat) tiles; coef is scratch of 2 * C floats the launch fills and reads;
 * dgamma / dbeta are written (+= with accumulate_params) by workgroup 0. */
int dsnt_bn_act_bwd_apply_pro(const float* da, const float* x, const float* scale, const float* shift,
                              const float* mean, const float* invstd, const float* partial, int ntiles,
                              float* dgamma, float* dbeta, int accumulate_params, float* coef, int relu,
                              float* dx, int accumulate, int64_t M, int C, float* amax, void* stream);
/* Both with the result added to a tensor of its OWN instead of accumulated in place: dx = base + value, `base` read and never
 * written (amax may be NULL).  The gradient that dx continues — dL/d(block output) of a Bottleneck, hourglass.py:48 `out += residual`
 * — stays intact for a reader that comes later: the weight gradient of the block's conv3 then waits for its parameter bucket's grouped
 * launch like the other low-resolution weight gradients, instead of costing the dependency chain a launch of 2..32 workgroups. */
int dsnt_bn_act_bwd_apply_base(const float* da, const float* x, const float* scale, const float* shift,
                               const float* mean, const float* invstd, const float* coef, int relu,
                               const float* base, float* dx, int64_t M, int C, float* amax, void* stream);
int dsnt_bn_act_bwd_apply_pro_base(const float* da, const float* x, const float* scale, const float* shift,
                                   const float* mean, const float* invstd, const float* partial, int ntiles,
                                   float* dgamma, float* dbeta, int accumulate_params, float* coef, int relu,
                                   const float* base, float* dx, int64_t M, int C, float* amax, void* stream);
int dsnt_fill_zero(float* p, int64_t n, void* stream);
/* The other kernels that (re)write a whole gradient tensor, with the same amax side output: a tensor written by several
 * of them in turn is bounded by the maximum over their amaxes (each rewrites all of it), so gradients accumulated along
 * the residual stream keep a valid fp16x3 bound. */
int dsnt_axpy_amax(const float* x, float* y, float a, int accumulate, int64_t n, float* amax, void* stream);
int dsnt_maxpool2_bwd_amax(const float* dy, const uint8_t* idx, float* dx, int accumulate, int N, int H, int W, int C,
                           float* amax, void* stream);
/* ... with a SECOND gradient of x added in the same pass: dx (+)= extra + the routed dy (amax may be NULL).  The hourglass's
 * stack input (hourglass.py:66-77: `up1 = self.hg[n-1][0](x)`, `low1 = F.max_pool2d(x, 2)` — and x also feeds the residual sum of
 * hourglass.py:175) collects three gradients; the skip branch's arrives in a buffer of its own, and this launch adds it instead of a
 * separate x.grad += branch.grad pass. */
int dsnt_maxpool2_bwd_add(const float* dy, const uint8_t* idx, float* dx, int accumulate, const float* extra,
                          int N, int H, int W, int C, float* amax, void* stream);
int dsnt_upsample2_bwd_amax(const float* dout, float* dlow, int accumulate, int N, int H, int W, int C, float* amax,
                            void* stream);

/* F.max_pool2d(x, 2, stride=2) (hourglass.py:80,111,162): y [N][H/2][W/2][C],
 * idx = position (0..3) of the first maximum in scan order, for the backward. */
int dsnt_maxpool2_fwd(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C,
                      void* stream);
int dsnt_maxpool2_bwd(const float* dy, const uint8_t* idx, float* dx, int accumulate,
                      int N, int H, int W, int C, void* stream);

/* ResNet pieces (torchvision resnet consumed by model.py:79-136; third-party, restated):
 * F.max_pool2d(x, 3, stride=2, padding=1): y [N][(H-1)/2+1][(W-1)/2+1][C], idx = winning tap 0..8;
 * y = relu?(scale*x + shift + res) (block tail: bn -> += identity -> relu); dz = dy * [y > 0];
 * zero-stuffing of dy for the data gradient of a strided convolution: out[n][oh*s][ow*s][c] = dy[n][oh][ow][c]. */
int dsnt_maxpool3s2_fwd(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, void* stream);
int dsnt_maxpool3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int accumulate, int N, int H, int W,
                        int C, void* stream);
int dsnt_bn_add_act_fwd(const float* x, const float* scale, const float* shift, const float* res, int relu,
                        float* y, int64_t M, int C, void* stream);
int dsnt_relu_bwd(const float* dy, const float* y, float* dz, int64_t n, void* stream);
int dsnt_zero_insert(const float* dy, float* out, int N, int Ho, int Wo, int C, int Hs, int Ws, int stride,
                     void* stream);

/* Data gradient of a STRIDED convolution, native (csrc/dgrad_up.hip): dx [N][H][W][Cin] (+= when res1 == dx) from
 * dy [N][Ho][Wo][Cout] and wd = dsnt_conv_pack_dgrad(w) ([Cin][R][S][Cout], taps flipped), g = the FORWARD convolution
 * (2 <= stride <= 4, Cout % 16 == 0: ask dsnt_conv_dgrad_strided_ok).  The pixels of dx are computed phase by phase
 * ((ih % stride, iw % stride): a 3x3 / 2 convolution has phases of 1, 2, 2 and 4 taps), so neither the zero-stuffed
 * copy of dy nor the multiplications by its zeros exist: 1 / stride^2 of the work of dsnt_zero_insert + dsnt_conv_fwd,
 * which it replaces in the engine.  Exact fp32 (v_mfma_f32_32x32x2_f32).  Optional epilogues, as dsnt_conv_fwd_ex:
 * res1 (may alias dx), or bnb + stats_partial ([dsnt_conv_dgrad_strided_tiles(g)][2][Cin], every row written),
 * or tail->amax (the other dsnt_out_bounds fields must be unset).  Replaces cuDNN's backward-data of the torchvision ResNet
 * stride-2 convolutions (conv1, layerN[0].conv1 / .conv2, downsample[0]) consumed by /root/reference/src/dsnt/model.py:103-121. */
int dsnt_conv_dgrad_strided(const float* dy, const float* wd, float* dx, const float* res1, float* stats_partial,
                            const dsnt_conv_geom* g, const dsnt_bn_bwd_epilogue* bnb, const dsnt_out_bounds* tail, void* stream);
int dsnt_conv_dgrad_strided_ok(const dsnt_conv_geom* g);
int dsnt_conv_dgrad_strided_tiles(const dsnt_conv_geom* g);

/* out = up + nearest_upsample2x(low) (hourglass.py:58,88-89); low [N][H/2][W/2][C]. */
int dsnt_upsample2_add_fwd(const float* up, const float* low, float* out,
                           int N, int H, int W, int C, void* stream);
/* dlow (+)= 2x2 block sums of dout. */
int dsnt_upsample2_bwd(const float* dout, float* dlow, int accumulate,
                       int N, int H, int W, int C, void* stream);

/* The same two forward ops with the BatchNorm statistics of their output in the same pass (the next op of the
 * hourglass is a BatchNorm: hourglass.py:33,78-90): partial[ceil(M/128)][2][C] exactly as dsnt_bn_stats over the
 * stored result would give (bit-identical), M = output pixels.  partial may be NULL (eval mode: no statistics;
 * tail->amax / amax_bn can still ask for the next convolution's fp16x3 operand bound). */
int dsnt_maxpool2_fwd_stats(const float* x, float* y, uint8_t* idx, float* partial, int N, int H, int W, int C,
                            const dsnt_out_bounds* tail, void* stream);
int dsnt_upsample2_add_fwd_stats(const float* up, const float* low, float* out, float* partial, int N, int H, int W,
                                 int C, const dsnt_out_bounds* tail, void* stream);
/* y = relu?(x * scale + shift) (dsnt_bn_act_fwd: the stem's materialised BatchNorm + ReLU, hourglass.py:157-159) with the
 * statistics of y in the same pass — the first Bottleneck's BatchNorm reads y next — and, through tail->amax / amax_bn, the
 * fp16x3 bound of the skip projection that reads y raw.  partial: [ceil(M/128)][2][C], may be NULL. */
int dsnt_bn_act_fwd_stats(const float* x, const float* scale, const float* shift, int relu, float* y, float* partial,
                          int64_t M, int C, const dsnt_out_bounds* tail, void* stream);

/* The 7x7 / stride 2 / pad 3 stem convolution on a C <= 4 channel image (hourglass.py:106; torchvision resnet conv1) as a
 * 4x4 / stride 1 / pad 1 convolution: dsnt_s2d_input writes the image as [N][H/2+1][W/2+1][16] (2x2 pixel blocks -> 16 channels,
 * one zero block row / column in front; tail->amax, if given, receives max|image| as the fp16x3 operand bound);
 * dsnt_s2d_weights re-packs OHWI [Cout][7][7][4] weights into [Cout][4][4][16] (back = 0) or gathers a [Cout][4][4][16] weight
 * gradient back into [Cout][7][7][4] (back = 1).  Any convolution entry point then runs the stem with K = 256. */
int dsnt_s2d_input(const float* src_nchw, float* dst, int N, int C, int H, int W, const dsnt_out_bounds* tail, void* stream);
int dsnt_s2d_weights(const float* w, float* w2, int Cout, int back, void* stream);
/* dsnt_s2d_weights (back = 0) + the filter's maximum (bound[64]) + its two fp16 planes (scaled as dsnt_split_f16x2 does,
 * plane stride Cout*256) + its three bf16 planes (dsnt_split_bf16x3 layout) in one launch; w2 (fp32 copy) may be NULL. */
int dsnt_s2d_weights_prep(const float* w, float* w2, void* planes16, void* planes_bf16, float* bound, int Cout, void* stream);

/* y (+)= a*x, flat; n % 4 == 0 not required. */
int dsnt_axpy(const float* x, float* y, float a, int accumulate, int64_t n, void* stream);

/* Layout changes at the model boundary (logical NCHW surface, model.py:229-231,
 * tests/test_model.py:22-23): src [N][C][HW] -> dst [N][HW][Cpad] (zero-filled pad)
 * and back (dst [N][C][HW] <- src [N][HW][Cpad]). */
int dsnt_nchw_to_nhwc(const float* src, float* dst, int N, int C, int HW, int Cpad, void* stream);
int dsnt_nhwc_to_nchw(const float* src, float* dst, int N, int C, int HW, int Cpad, void* stream);

/* ------------------------------------------------------------------ optimiser
 * train.py:314-326 — flat multi-tensor updates over the whole parameter arena. */
int dsnt_rmsprop_step(float* p, const float* g, float* square_avg, int64_t n, float lr,
                      float alpha, float eps, float weight_decay, float grad_scale, void* stream);
int dsnt_sgd_step(float* p, const float* g, float* momentum_buf, int64_t n, float lr,
                  float momentum, float weight_decay, float grad_scale, int first_step,
                  void* stream);
/* The same updates under the step's non-finite guard (train.py:360-371 checks the loss for NaN before
 * `backward()` / `step()` and dumps the model): `flag` is a device int[2] the caller owns and zeroes.  If
 * flag[0] != 0 when the kernel starts (the loss check below fired earlier on this stream) NOTHING is updated — the
 * weights stay the last finite ones; an element whose own gradient is not finite is skipped and raises
 * DSNT_FLAG_GRAD in flag[1], which the next dsnt_nonfinite_flag call promotes into flag[0]. */
#define DSNT_FLAG_LOSS 1
#define DSNT_FLAG_GRAD 2
int dsnt_rmsprop_step_guarded(float* p, const float* g, float* square_avg, int64_t n, float lr, float alpha,
                              float eps, float weight_decay, float grad_scale, int* flag, void* stream);
int dsnt_sgd_step_guarded(float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum,
                          float weight_decay, float grad_scale, int first_step, int* flag, void* stream);
/* flag[0] |= code if any of x[0..n) is NaN or +-inf (train.py:360 `np.isnan(loss.data[0])`, without the
 * device-to-host synchronisation: the caller reads the flag asynchronously); also flag[0] |= flag[1]. */
int dsnt_nonfinite_flag(const float* x, int64_t n, int* flag, int code, void* stream);

/* ------------------------------------------------------------------ launch lists
 * The reference re-records its autograd graph and issues ~10^3 device ops from Python every step (train.py:355-384).
 * Here a step is a fixed sequence of the entry points above; a launch list captures that sequence ONCE and replays it
 * from C: between dsnt_list_begin and dsnt_list_end (per thread) every entry point validates its arguments as usual but
 * records its kernel launches (arguments by value) instead of enqueueing them, and its `stream` argument is read as a
 * LANE index 0..DSNT_MAX_LANES-1.  dsnt_list_sync appends "lane dst waits for everything lane src has been given so
 * far" (an event the list owns); dsnt_list_mark ends a segment (the caller does host work there — e.g. starts an
 * all-reduce — between replays) and returns the index of the segment that begins.  dsnt_list_replay enqueues one
 * segment (-1: all) on the caller's streams, lane i -> streams[i].  A list holds pointers, not memory: the caller keeps
 * every tensor it recorded alive and unchanged in address.  dsnt_amax (memset) cannot be recorded. */
#define DSNT_MAX_LANES 8
typedef struct dsnt_list dsnt_list;
dsnt_list* dsnt_list_create(void);
void dsnt_list_destroy(dsnt_list* l);
int dsnt_list_begin(dsnt_list* l);
int dsnt_list_end(void);
int dsnt_list_sync(dsnt_list* l, int src_lane, int dst_lane);
int dsnt_list_mark(dsnt_list* l);
int dsnt_list_segments(const dsnt_list* l);
int dsnt_list_size(const dsnt_list* l);
int dsnt_list_replay(const dsnt_list* l, int segment, void* const* streams, int nstreams);
/* Persistent stages (round 6; dsnt_version() >= 114).  On the 8 x 8 and 4 x 4 levels of an hourglass
 * (/root/reference/src/dsnt/hourglass.py:78-90) a step is ~70 dependent launches per hourglass of a few microseconds each, every
 * one behind a ~5-us launch boundary on the critical path.  dsnt_list_fuse rewrites a RECORDED list: every run of >= min_run
 * consecutive launches of one lane that can take part (the K-split convolution, the BatchNorm-backward apply, max-pool /
 * upsample + add with statistics and their backward; each with at most max_vgrid workgroups; no lane synchronisation or segment
 * mark touching the lane inside the run) becomes ONE launch of a persistent kernel of min(grid_cap, the run's widest launch)
 * co-resident 512-thread workgroups, which walk the recorded launches' workgroup indices through the same device functions
 * (bit-identical results) with a chip-wide barrier where a kernel boundary was.  workspace: caller-owned device memory of
 * dsnt_list_fuse_bytes(l, min_run, max_vgrid) bytes, 64-byte aligned, alive and untouched as long as the list is replayed (launch
 * tables + per-stage counters; written here by a blocking copy).  Returns the number of stage launches created (0: nothing to
 * fuse), negative on error.  A stage whose barrier gives up (2^21 polls: a workgroup never arrived) raises word 2 of its counter
 * line and every workgroup leaves the kernel: wrong results the caller can detect, never a hung device.
 * dsnt_list_stages: stage launches in the list and (launches_inside) the recorded launches they replaced;
 * dsnt_list_stage_errors: how many of them have given up since the list was fused (blocking device reads; 0 when healthy). */
int64_t dsnt_list_fuse_bytes(const dsnt_list* l, int min_run, int max_vgrid);
int dsnt_list_fuse(dsnt_list* l, void* workspace, int64_t bytes, int min_run, int max_vgrid, int grid_cap);
int dsnt_list_fuse_plan(const dsnt_list* l, int min_run, int max_vgrid, int* launches);   /* what fuse would do; no device needed */
int dsnt_list_stages(const dsnt_list* l, int* launches_inside);
int dsnt_list_stage_errors(const dsnt_list* l);

/* ------------------------------------------------------------------ metrics
 * evaluator.py:66-81 + train.py:243-258: PCKh hits on device.
 * pred/target [B][J][2] normalised coords; m [B][2][2], b [B][2] back-projection (f64 maths);
 * mask [B][J]; head [B]; out hits[B][J] (1/0), valid[B][J] (mask == 1). */
int dsnt_pckh(const float* pred, const float* target, const double* m, const double* b,
              const float* mask, const double* head, float threshold, float* hits,
              float* valid, int B, int J, void* stream);

#ifdef __cplusplus
}
#endif
#endif
