// HBM-bound NHWC kernels around the convolutions: batch-norm statistics / finalisation /
// activation and their backward, 2x2 max-pool, nearest-upsample + add, axpy, layout changes,
// the flat optimiser updates and the PCKh hit test.  All 16-byte vectorised (C % 4 == 0),
// grid-stride where the work is flat, one workgroup per 128-row tile where a per-channel
// reduction is produced (partials are combined in fp64 by the finalise kernels: deterministic,
// no atomics).
#include "common.h"
#include <string.h>
#include "bn_pro.h"
#include "conv_split.h"
#include "ew_bodies.h"
#include "stage.h"

// Thread mapping of the tile kernels below, shared so that their sums stay bit-identical to each other: a workgroup
// covers `cgs` float4 column groups x (256 / cgs) row lanes of one 128-row tile.  Large tensors: all columns in one
// workgroup (up to 256 groups per pass).  Small ones (fewer than 256 tiles — the low-resolution hourglass levels, where
// a 4..64-workgroup launch is pure latency): 16 column groups per workgroup, the rest of the columns on gridDim.y, so a
// thread walks 8 rows instead of 32 and the launch has 4x (C = 256) the workgroups.
static inline int tile_cgs(long tiles, int C4) { return tiles < 256 && C4 > 16 && C4 % 16 == 0 ? 16 : (C4 < 256 ? C4 : 256); }
static inline unsigned tile_grid_y(long tiles, int C4) { const int c = tile_cgs(tiles, C4); return c == 16 && C4 > 16 ? C4 / 16 : 1; }

// ---------------------------------------------------------------- per-channel tile reductions
// MODE 0: (sum x, sum x^2)          MODE 1: (sum dz, sum dz*xhat) for y = relu?(bn(x))
// MODE 2: the same sums for y = relu?(bn(x) + skip) (the tail of a torchvision residual block): the ReLU mask comes from the stored
// output (`ymask` > 0) and dz = a * mask is WRITTEN (`dz_out`: the skip branch's gradient and the apply pass read it)
template <int MODE>
__global__ __launch_bounds__(256) void tile_reduce_kernel(
    const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, int relu, float* __restrict__ partial, long M, int C, int cgs,
    const float* __restrict__ ymask = nullptr, float* __restrict__ dz_out = nullptr) {
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x;
    const int C4 = C >> 2;
    const int rpar = 256 / cgs;               // row lanes (cgs = column groups handled per pass: tile_cgs)
    const int cg_l = tid % cgs, rl = tid / cgs;
    const bool active = rl < rpar;
    const long row0 = (long)blockIdx.x * TILE_ROWS;
    const long row1 = row0 + TILE_ROWS < M ? row0 + TILE_ROWS : M;
    for (int cg0 = blockIdx.y * cgs; cg0 < C4; cg0 += cgs * gridDim.y) {
        const int cg = cg0 + cg_l;
        float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
        if (active && cg < C4) {
            float4 sc, sh, mu, is;
            if (MODE == 1) {
                sc = reinterpret_cast<const float4*>(scale)[cg];
                sh = reinterpret_cast<const float4*>(shift)[cg];
            }
            if (MODE >= 1) {
                mu = reinterpret_cast<const float4*>(mean)[cg];
                is = reinterpret_cast<const float4*>(invstd)[cg];
            }
            for (long r = row0 + rl; r < row1; r += rpar) {
                const float4 v = reinterpret_cast<const float4*>(a + r * C)[cg];
                if (MODE == 0) {
                    s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                    s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y);
                    s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
                } else {
                    const float4 xv = reinterpret_cast<const float4*>(x + r * C)[cg];
                    float4 dz = v;
                    if (MODE == 2) {
                        if (relu) {
                            const float4 yv = reinterpret_cast<const float4*>(ymask + r * C)[cg];
                            if (yv.x <= 0.f) dz.x = 0.f;
                            if (yv.y <= 0.f) dz.y = 0.f;
                            if (yv.z <= 0.f) dz.z = 0.f;
                            if (yv.w <= 0.f) dz.w = 0.f;
                        }
                        reinterpret_cast<float4*>(dz_out + r * C)[cg] = dz;
                    } else if (relu) {
                        if (fmaf(xv.x, sc.x, sh.x) <= 0.f) dz.x = 0.f;
                        if (fmaf(xv.y, sc.y, sh.y) <= 0.f) dz.y = 0.f;
                        if (fmaf(xv.z, sc.z, sh.z) <= 0.f) dz.z = 0.f;
                        if (fmaf(xv.w, sc.w, sh.w) <= 0.f) dz.w = 0.f;
                    }
                    s1.x += dz.x; s1.y += dz.y; s1.z += dz.z; s1.w += dz.w;
                    s2.x = fmaf(dz.x, (xv.x - mu.x) * is.x, s2.x);
                    s2.y = fmaf(dz.y, (xv.y - mu.y) * is.y, s2.y);
                    s2.z = fmaf(dz.z, (xv.z - mu.z) * is.z, s2.z);
                    s2.w = fmaf(dz.w, (xv.w - mu.w) * is.w, s2.w);
                }
            }
        }
        __syncthreads();
        float* mine = red + tid * 8;
        mine[0] = s1.x; mine[1] = s1.y; mine[2] = s1.z; mine[3] = s1.w;
        mine[4] = s2.x; mine[5] = s2.y; mine[6] = s2.z; mine[7] = s2.w;
        __syncthreads();
        if (tid < cgs && cg0 + tid < C4) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int j = 0; j < rpar; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += red[(j * cgs + tid) * 8 + e];
            float* p0 = partial + ((size_t)blockIdx.x * 2 + 0) * C + (size_t)(cg0 + tid) * 4;
            float* p1 = partial + ((size_t)blockIdx.x * 2 + 1) * C + (size_t)(cg0 + tid) * 4;
            *reinterpret_cast<float4*>(p0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4*>(p1) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
    }
}

extern "C" int dsnt_bn_stats(const float* x, float* partial, int64_t M, int C, void* stream) {
    DSNT_REQUIRE(x && partial && M > 0 && C > 0, DSNT_ERR_ARG, "dsnt_bn_stats: bad argument");
    DSNT_REQUIRE(C % 4 == 0 && dsnt_aligned16(x) && dsnt_aligned16(partial), DSNT_ERR_ALIGN,
                 "dsnt_bn_stats: C %% 4 and 16-byte alignment required");
    const int tiles = (int)((M + TILE_ROWS - 1) / TILE_ROWS);
    DSNT_LAUNCH(tile_reduce_kernel<0>, dim3(tiles, tile_grid_y(tiles, C / 4)), dim3(256), 0, (hipStream_t)stream, x,
                       nullptr, nullptr, nullptr, nullptr, nullptr, 0, partial, (long)M, C, tile_cgs(tiles, C / 4));
    DSNT_CHECK_LAUNCH("dsnt_bn_stats");
}

// 2x2 max-pool / nearest-upsample + add with the BatchNorm statistics of their OUTPUT in the same pass: one
// workgroup per 128 output rows (pixels), same thread mapping, accumulation order and partial format as
// tile_reduce_kernel<0>, so the sums are bit-identical to a separate dsnt_bn_stats over the stored tensor (which
// cost one more read of it: 18 launches per hg2 step).  OP 0: y = maxpool2(a) (+ arg-max byte), a is [N][2Ho][2Wo][C];
// OP 1: y = a + upsample2(b), b is [N][Ho/2][Wo/2][C].  OP 2: y = relu?(a * bn_scale + bn_shift) (the stem's materialised
// BatchNorm + ReLU, hourglass.py:157-159).  Ho, Wo: OUTPUT size.
extern "C" int dsnt_maxpool2_fwd_stats(const float* x, float* y, uint8_t* idx, float* partial, int N, int H, int W,
                                       int C, const dsnt_out_bounds* g_tail, void* stream) {
    OutBoundsP tail;
    if (int e = out_bounds_fill(tail, g_tail, "dsnt_maxpool2_fwd_stats")) return e;
    DSNT_REQUIRE(x && y && idx && N > 0 && H > 0 && W > 0 && C > 0, DSNT_ERR_ARG,
                 "dsnt_maxpool2_fwd_stats: bad argument");
    DSNT_REQUIRE(H % 2 == 0 && W % 2 == 0, DSNT_ERR_SHAPE, "dsnt_maxpool2_fwd_stats: H and W must be even (got %dx%d)", H, W);
    DSNT_REQUIRE(C % 4 == 0 && dsnt_aligned16(x) && dsnt_aligned16(y) && dsnt_aligned16(partial) &&
                 (((uintptr_t)idx) & 3) == 0, DSNT_ERR_ALIGN, "dsnt_maxpool2_fwd_stats: alignment");
    const long M = (long)N * (H / 2) * (W / 2);
    const long tiles = (M + TILE_ROWS - 1) / TILE_ROWS;
    const TileOpP q{x, nullptr, y, idx, partial, N, H / 2, W / 2, C, tile_cgs(tiles, C / 4), tail, nullptr, nullptr, 0};
    DSNT_LAUNCH_OP(DSNT_ST_TILE_POOL, tile_op_stats_kernel<0>, dim3((unsigned)tiles, tile_grid_y(tiles, C / 4)), dim3(256), 0, stream, q);
    DSNT_CHECK_LAUNCH("dsnt_maxpool2_fwd_stats");
}

extern "C" int dsnt_upsample2_add_fwd_stats(const float* up, const float* low, float* out, float* partial, int N,
                                            int H, int W, int C, const dsnt_out_bounds* g_tail, void* stream) {
    OutBoundsP tail;
    if (int e = out_bounds_fill(tail, g_tail, "dsnt_upsample2_add_fwd_stats")) return e;
    DSNT_REQUIRE(up && low && out && N > 0 && H > 0 && W > 0 && C > 0, DSNT_ERR_ARG,
                 "dsnt_upsample2_add_fwd_stats: bad argument");
    DSNT_REQUIRE(H % 2 == 0 && W % 2 == 0, DSNT_ERR_SHAPE, "dsnt_upsample2_add_fwd_stats: H and W must be even");
    DSNT_REQUIRE(C % 4 == 0 && dsnt_aligned16(up) && dsnt_aligned16(low) && dsnt_aligned16(out) &&
                 dsnt_aligned16(partial), DSNT_ERR_ALIGN, "dsnt_upsample2_add_fwd_stats: alignment");
    const long M = (long)N * H * W;
    const long tiles = (M + TILE_ROWS - 1) / TILE_ROWS;
    const TileOpP q{up, low, out, nullptr, partial, N, H, W, C, tile_cgs(tiles, C / 4), tail, nullptr, nullptr, 0};
    DSNT_LAUNCH_OP(DSNT_ST_TILE_UPADD, tile_op_stats_kernel<1>, dim3((unsigned)tiles, tile_grid_y(tiles, C / 4)), dim3(256), 0, stream, q);
    DSNT_CHECK_LAUNCH("dsnt_upsample2_add_fwd_stats");
}

extern "C" int dsnt_bn_act_fwd_stats(const float* x, const float* scale, const float* shift, int relu, float* y,
                                     float* partial, int64_t M, int C, const dsnt_out_bounds* g_tail, void* stream) {
    OutBoundsP tail;
    if (int e = out_bounds_fill(tail, g_tail, "dsnt_bn_act_fwd_stats")) return e;
    DSNT_REQUIRE(x && scale && shift && y && M > 0 && C > 0, DSNT_ERR_ARG, "dsnt_bn_act_fwd_stats: bad argument");
    DSNT_REQUIRE(C % 4 == 0 && dsnt_aligned16(x) && dsnt_aligned16(y) && dsnt_aligned16(scale) && dsnt_aligned16(shift) &&
                 dsnt_aligned16(partial), DSNT_ERR_ALIGN, "dsnt_bn_act_fwd_stats: alignment");
    const long tiles = ((long)M + TILE_ROWS - 1) / TILE_ROWS;
    // rows are flat here: N = 1, Ho = 1, Wo = M would overflow int for nothing — the kernel only needs M = N*Ho*Wo
    DSNT_REQUIRE(M < (1ll << 31), DSNT_ERR_SHAPE, "dsnt_bn_act_fwd_stats: M too large");
    const TileOpP q{x, nullptr, y, nullptr, partial, 1, 1, (int)M, C, tile_cgs(tiles, C / 4), tail, scale, shift, relu};
    DSNT_LAUNCH_OP(DSNT_ST_NONE, tile_op_stats_kernel<2>, dim3((unsigned)tiles, tile_grid_y(tiles, C / 4)), dim3(256), 0, stream, q);
    DSNT_CHECK_LAUNCH("dsnt_bn_act_fwd_stats");
}

extern "C" int dsnt_bn_act_bwd_reduce(const float* da, const float* x, const float* scale,
                                      const float* shift, const float* mean, const float* invstd,
                                      int relu, float* partial, int64_t M, int C, void* stream) {
    DSNT_REQUIRE(da && x && scale && shift && mean && invstd && partial && M > 0 && C > 0,
                 DSNT_ERR_ARG, "dsnt_bn_act_bwd_reduce: bad argument");
    DSNT_REQUIRE(C % 4 == 0 && dsnt_aligned16(x) && dsnt_aligned16(da) && dsnt_aligned16(partial) &&
                 dsnt_aligned16(scale) && dsnt_aligned16(shift) && dsnt_aligned16(mean) &&
                 dsnt_aligned16(invstd), DSNT_ERR_ALIGN,
                 "dsnt_bn_act_bwd_reduce: C %% 4 and 16-byte alignment required");
    const int tiles = (int)((M + TILE_ROWS - 1) / TILE_ROWS);
    DSNT_LAUNCH(tile_reduce_kernel<1>, dim3(tiles, tile_grid_y(tiles, C / 4)), dim3(256), 0, (hipStream_t)stream, da, x,
                       scale, shift, mean, invstd, relu, partial, (long)M, C, tile_cgs(tiles, C / 4));
    DSNT_CHECK_LAUNCH("dsnt_bn_act_bwd_reduce");
}

// The backward of y = relu?(bn(x) + skip) up to the BatchNorm's two reductions, in one pass: dz = da * (y > 0) written, and the
// tile sums (sum dz, sum dz * xhat) for dsnt_bn_bwd_finalize — dsnt_relu_bwd + dsnt_bn_act_bwd_reduce(relu = 0) as one launch.
extern "C" int dsnt_bn_add_act_bwd_reduce(const float* da, const float* y, const float* x, const float* mean, const float* invstd,
                                          int relu, float* dz, float* partial, int64_t M, int C, void* stream) {
    DSNT_REQUIRE(da && y && x && mean && invstd && dz && partial && M > 0 && C > 0, DSNT_ERR_ARG, "dsnt_bn_add_act_bwd_reduce: bad argument");
    DSNT_REQUIRE(C % 4 == 0 && dsnt_aligned16(x) && dsnt_aligned16(da) && dsnt_aligned16(y) && dsnt_aligned16(dz) && dsnt_aligned16(partial) &&
                 dsnt_aligned16(mean) && dsnt_aligned16(invstd), DSNT_ERR_ALIGN, "dsnt_bn_add_act_bwd_reduce: C %% 4 and 16-byte alignment required");
    const int tiles = (int)((M + TILE_ROWS - 1) / TILE_ROWS);
    DSNT_LAUNCH(tile_reduce_kernel<2>, dim3(tiles, tile_grid_y(tiles, C / 4)), dim3(256), 0, (hipStream_t)stream, da, x,
                       nullptr, nullptr, mean, invstd, relu, partial, (long)M, C, tile_cgs(tiles, C / 4), y, dz);
    DSNT_CHECK_LAUNCH("dsnt_bn_add_act_bwd_reduce");
}

extern "C" int dsnt_bn_finalize(const float* partial, int ntiles, int64_t M, int C,
                                const float* gamma, const float* beta, float* running_mean,
                                float* running_var, float momentum, float eps, int training,
                                float* mean, float* invstd, float* scale, float* shift,
                                void* stream) {
    DSNT_REQUIRE(mean && invstd && scale && shift && C > 0 && M > 0, DSNT_ERR_ARG,
                 "dsnt_bn_finalize: bad argument");
    DSNT_REQUIRE(training ? (partial != nullptr && ntiles > 0) : (running_mean && running_var),
                 DSNT_ERR_ARG, "dsnt_bn_finalize: missing statistics source");
    DSNT_REQUIRE((running_mean == nullptr) == (running_var == nullptr), DSNT_ERR_ARG,
                 "dsnt_bn_finalize: running_mean/var must be given together");
    const double unbias = M > 1 ? (double)M / (double)(M - 1) : 1.0;
    const BnFinP q{partial, ntiles, 1.0 / (double)M, unbias, C, gamma, beta, running_mean, running_var, momentum, eps, training,
                   mean, invstd, scale, shift, 0, BnBoundP{nullptr, nullptr, 0.f, nullptr}};
    DSNT_LAUNCH_OP(DSNT_ST_FIN_FWD, bn_finalize_kernel<0>, dim3((C + 15) / 16), dim3(FIN_T), 0, stream, q);
    DSNT_CHECK_LAUNCH("dsnt_bn_finalize");
}

// Eval-mode BatchNorm vectors of MANY layers in one launch (table rows of int64: {gamma*, beta*, running_mean*,
// running_var*, mean*, invstd*, scale*, shift*, C, bits of float eps}): what dsnt_bn_finalize(training = 0) computes per
// layer — 96 launches per hg2 forward otherwise, the larger part of a batch-1 inference (inference.py:33-48).
__global__ __launch_bounds__(256) void bn_eval_prep_kernel(const long long* __restrict__ table) {
    const long long* t = table + (size_t)blockIdx.x * 10;
    const float* gamma = reinterpret_cast<const float*>(t[0]);
    const float* beta = reinterpret_cast<const float*>(t[1]);
    const float* rm = reinterpret_cast<const float*>(t[2]);
    const float* rv = reinterpret_cast<const float*>(t[3]);
    float* mean = reinterpret_cast<float*>(t[4]);
    float* invstd = reinterpret_cast<float*>(t[5]);
    float* scale = reinterpret_cast<float*>(t[6]);
    float* shift = reinterpret_cast<float*>(t[7]);
    const int C = (int)t[8];
    const float eps = __uint_as_float((unsigned)t[9]);
    for (int c = threadIdx.x; c < C; c += 256) {
        const double var = rv[c];
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        const float mu = rm[c];
        const float sc = gamma ? gamma[c] * is : is;
        mean[c] = mu; invstd[c] = is; scale[c] = sc;
        shift[c] = (beta ? beta[c] : 0.f) - mu * sc;
    }
}

extern "C" int dsnt_bn_eval_prep(const int64_t* table, int rows, void* stream) {
    DSNT_REQUIRE(table && rows > 0, DSNT_ERR_ARG, "dsnt_bn_eval_prep: bad argument");
    DSNT_LAUNCH(bn_eval_prep_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const long long*)table);
    DSNT_CHECK_LAUNCH("dsnt_bn_eval_prep");
}

extern "C" int dsnt_bn_bwd_finalize(const float* partial, int ntiles, int64_t M, int C,
                                    float* dgamma, float* dbeta, int accumulate, float* coef,
                                    void* stream) {
    DSNT_REQUIRE(partial && coef && ntiles > 0 && C > 0 && M > 0, DSNT_ERR_ARG,
                 "dsnt_bn_bwd_finalize: bad argument");
    // DSNT_BN_FROZEN: the forward ran on fixed (running) statistics — dx = scale dz, both coefficients zero; dgamma / dbeta as always
    const double invM = (accumulate & DSNT_BN_FROZEN) ? 0.0 : 1.0 / (double)M;
    const BnFinP q{partial, ntiles, invM, 1.0, C, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 1, dgamma, dbeta, coef, nullptr,
                   accumulate & 1, BnBoundP{nullptr, nullptr, 0.f, nullptr}};
    DSNT_LAUNCH_OP(DSNT_ST_FIN_BWD, bn_finalize_kernel<1>, dim3((C + 15) / 16), dim3(FIN_T), 0, stream, q);
    DSNT_CHECK_LAUNCH("dsnt_bn_bwd_finalize");
}

// dsnt_bn_bwd_finalize that also leaves the bound of the BatchNorm's dx for a consumer that never sees dx in memory
// (dsnt_conv1x1_bwd_f16x3 with a dsnt_bn_bwd_apply): scale = the BatchNorm's forward scale vector (gamma * invstd), dz_amax = the
// 64-slot max |dz| its data-gradient producer left (dsnt_out_bounds.amax), bound_out = 64 slots, zeroed by the caller once per step.
extern "C" int dsnt_bn_bwd_finalize_bound(const float* partial, int ntiles, int64_t M, int C, float* dgamma, float* dbeta,
                                          int accumulate, float* coef, const float* scale, const float* dz_amax,
                                          float* bound_out, void* stream) {
    DSNT_REQUIRE(partial && coef && scale && dz_amax && bound_out && ntiles > 0 && C > 0 && M > 0, DSNT_ERR_ARG,
                 "dsnt_bn_bwd_finalize_bound: bad argument");
    const BnFinP q{partial, ntiles, 1.0 / (double)M, 1.0, C, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 1, dgamma, dbeta, coef, nullptr,
                   accumulate, BnBoundP{scale, dz_amax, sqrtf((float)M), reinterpret_cast<unsigned*>(bound_out)}};
    DSNT_LAUNCH_OP(DSNT_ST_FIN_BWD, bn_finalize_kernel<1>, dim3((C + 15) / 16), dim3(FIN_T), 0, stream, q);
    DSNT_CHECK_LAUNCH("dsnt_bn_bwd_finalize_bound");
}

// ---------------------------------------------------------------- flat elementwise
__global__ void bn_act_fwd_kernel(const float4* __restrict__ x, const float4* __restrict__ scale,
                                  const float4* __restrict__ shift, int relu, float4* __restrict__ y,
                                  long n4, int C4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % C4);
        const float4 v = x[i], sc = scale[cg], sh = shift[cg];
        float4 o = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z),
                               fmaf(v.w, sc.w, sh.w));
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        y[i] = o;
    }
}

static inline int flat_grid(long n, int block) {
    long g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" int dsnt_bn_act_fwd(const float* x, const float* scale, const float* shift, int relu,
                               float* y, int64_t M, int C, void* stream) {
    DSNT_REQUIRE(x && scale && shift && y && M > 0 && C > 0, DSNT_ERR_ARG, "dsnt_bn_act_fwd: bad argument");
    DSNT_REQUIRE(C % 4 == 0 && dsnt_aligned16(x) && dsnt_aligned16(y) && dsnt_aligned16(scale) &&
                 dsnt_aligned16(shift), DSNT_ERR_ALIGN, "dsnt_bn_act_fwd: alignment");
    const long n4 = (long)M * C / 4;
    DSNT_LAUNCH(bn_act_fwd_kernel, dim3(flat_grid(n4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x, (const float4*)scale, (const float4*)shift, relu, (float4*)y,
                       n4, C / 4);
    DSNT_CHECK_LAUNCH("dsnt_bn_act_fwd");
}

static int bn_act_bwd_apply_impl(const float* da, const float* x, const float* scale, const float* shift,
                                 const float* mean, const float* invstd, const float* coef, int relu, float* dx,
                                 int accumulate, int64_t M, int C, float* amax, void* stream,
                                 const BnBwdProP* pro = nullptr, const float* base = nullptr);

extern "C" int dsnt_bn_act_bwd_apply_amax(const float* da, const float* x, const float* scale,
                                          const float* shift, const float* mean, const float* invstd,
                                          const float* coef, int relu, float* dx, int accumulate,
                                          int64_t M, int C, float* amax, void* stream) {
    return bn_act_bwd_apply_impl(da, x, scale, shift, mean, invstd, coef, relu, dx, accumulate, M, C, amax, stream);
}

__global__ void fill_zero_kernel(float* p, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0.f;
}

extern "C" int dsnt_fill_zero(float* p, int64_t n, void* stream) {
    DSNT_REQUIRE(p && n > 0, DSNT_ERR_ARG, "dsnt_fill_zero: bad argument");
    DSNT_LAUNCH(fill_zero_kernel, dim3(flat_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, (long)n);
    DSNT_CHECK_LAUNCH("dsnt_fill_zero");
}

extern "C" int dsnt_bn_act_bwd_apply(const float* da, const float* x, const float* scale,
                                     const float* shift, const float* mean, const float* invstd,
                                     const float* coef, int relu, float* dx, int accumulate,
                                     int64_t M, int C, void* stream) {
    return bn_act_bwd_apply_impl(da, x, scale, shift, mean, invstd, coef, relu, dx, accumulate, M, C, nullptr, stream);
}

static int bn_act_bwd_apply_impl(const float* da, const float* x, const float* scale, const float* shift,
                                 const float* mean, const float* invstd, const float* coef, int relu, float* dx,
                                 int accumulate, int64_t M, int C, float* amax, void* stream, const BnBwdProP* pro, const float* base) {
    DSNT_REQUIRE(da && x && scale && shift && mean && invstd && coef && dx && M > 0 && C > 0,
                 DSNT_ERR_ARG, "dsnt_bn_act_bwd_apply: bad argument");
    DSNT_REQUIRE(C % 4 == 0 && dsnt_aligned16(da) && dsnt_aligned16(x) && dsnt_aligned16(dx) &&
                 dsnt_aligned16(coef) && dsnt_aligned16(base), DSNT_ERR_ALIGN, "dsnt_bn_act_bwd_apply: alignment");
    if (!base && accumulate) base = dx;
    const long n4 = (long)M * C / 4;
    int grid = flat_grid(n4, 256);
    BnBwdProP q;
    memset(&q, 0, sizeof(q));
    if (pro) {
        q = *pro;
        // every workgroup re-reads the tile sums in its prologue: few, fat workgroups (these launches are latency-bound)
        if (grid > 128) grid = 128;
    }
    const BnApplyP ap{(const float4*)da, (const float4*)x, (const float4*)scale, (const float4*)shift, (const float4*)mean,
                      (const float4*)invstd, (const float4*)coef, relu, (float4*)dx, (const float4*)base, n4, C / 4, (unsigned*)amax, q};
    if (((long)grid * 256) % (C / 4) == 0)
        DSNT_LAUNCH_OP(DSNT_ST_APPLY_FIXED, bn_act_bwd_apply_kernel<true>, dim3(grid), dim3(256), 0, stream, ap);
    else
        DSNT_LAUNCH_OP(DSNT_ST_APPLY, bn_act_bwd_apply_kernel<false>, dim3(grid), dim3(256), 0, stream, ap);
    DSNT_CHECK_LAUNCH("dsnt_bn_act_bwd_apply");
}

// The same with dsnt_bn_bwd_finalize folded into its prologue (tile sums of <= 64 KB: the 8x8 / 4x4 hourglass levels,
// where a finalise launch between two 10-us kernels costs the chain ~8 us): every workgroup sums partial[tiles][2][C]
// itself (fp64, fixed order) into coef, workgroup 0 also writes dgamma / dbeta (+= with accumulate_params).
extern "C" int dsnt_bn_act_bwd_apply_pro(const float* da, const float* x, const float* scale, const float* shift,
                                         const float* mean, const float* invstd, const float* partial, int ntiles,
                                         float* dgamma, float* dbeta, int accumulate_params, float* coef, int relu,
                                         float* dx, int accumulate, int64_t M, int C, float* amax, void* stream) {
    DSNT_REQUIRE(partial && ntiles > 0 && coef && C <= 256 && (long)ntiles * C <= 16384, DSNT_ERR_ARG,
                 "dsnt_bn_act_bwd_apply_pro: needs partial sums of at most 256 channels / 128 KB and a coef buffer");
    BnBwdProP q;
    q.partial = partial; q.tiles = ntiles; q.C = C; q.invM = 1.0 / (double)M;
    q.dgamma = dgamma; q.dbeta = dbeta; q.accumulate = accumulate_params; q.coef = coef;
    return bn_act_bwd_apply_impl(da, x, scale, shift, mean, invstd, coef, relu, dx, accumulate, M, C, amax, stream, &q);
}
// dx = base + value with `base` a tensor of its own (read, never written): the gradient that dx continues stays intact — for a
// weight gradient that reads it at the end of its parameter bucket (the grouped launch), after dx has long been written.
// amax may be NULL.  The _pro form: dsnt_bn_act_bwd_apply_pro likewise.
extern "C" int dsnt_bn_act_bwd_apply_base(const float* da, const float* x, const float* scale, const float* shift,
                                          const float* mean, const float* invstd, const float* coef, int relu,
                                          const float* base, float* dx, int64_t M, int C, float* amax, void* stream) {
    DSNT_REQUIRE(base && base != dx, DSNT_ERR_ARG, "dsnt_bn_act_bwd_apply_base: `base` must be a second tensor");
    return bn_act_bwd_apply_impl(da, x, scale, shift, mean, invstd, coef, relu, dx, 1, M, C, amax, stream, nullptr, base);
}
extern "C" int dsnt_bn_act_bwd_apply_pro_base(const float* da, const float* x, const float* scale, const float* shift,
                                              const float* mean, const float* invstd, const float* partial, int ntiles,
                                              float* dgamma, float* dbeta, int accumulate_params, float* coef, int relu,
                                              const float* base, float* dx, int64_t M, int C, float* amax, void* stream) {
    DSNT_REQUIRE(base && base != dx, DSNT_ERR_ARG, "dsnt_bn_act_bwd_apply_pro_base: `base` must be a second tensor");
    DSNT_REQUIRE(partial && ntiles > 0 && coef && C <= 256 && (long)ntiles * C <= 16384, DSNT_ERR_ARG,
                 "dsnt_bn_act_bwd_apply_pro_base: needs partial sums of at most 256 channels / 128 KB and a coef buffer");
    BnBwdProP q;
    q.partial = partial; q.tiles = ntiles; q.C = C; q.invM = 1.0 / (double)M;
    q.dgamma = dgamma; q.dbeta = dbeta; q.accumulate = accumulate_params; q.coef = coef;
    return bn_act_bwd_apply_impl(da, x, scale, shift, mean, invstd, coef, relu, dx, 1, M, C, amax, stream, &q, base);
}
// ---------------------------------------------------------------- pooling / upsampling
__global__ void maxpool2_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ y,
                                    uchar4* __restrict__ idx, int N, int H, int W, int C4) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % C4);
        long t = i / C4;
        const int ow = (int)(t % Wo); t /= Wo;
        const int oh = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const float4* base = x + (((long)n * H + 2 * oh) * W + 2 * ow) * C4 + cg;
        const float4 v0 = base[0], v1 = base[C4], v2 = base[(long)W * C4], v3 = base[(long)W * C4 + C4];
        float4 m = v0;
        uchar4 k = make_uchar4(0, 0, 0, 0);
#define POOL_STEP(V, P)                                  \
        if (V.x > m.x || V.x != V.x) { m.x = V.x; k.x = P; } \
        if (V.y > m.y || V.y != V.y) { m.y = V.y; k.y = P; } \
        if (V.z > m.z || V.z != V.z) { m.z = V.z; k.z = P; } \
        if (V.w > m.w || V.w != V.w) { m.w = V.w; k.w = P; }
        POOL_STEP(v1, 1) POOL_STEP(v2, 2) POOL_STEP(v3, 3)
#undef POOL_STEP
        y[i] = m;
        idx[i] = k;
    }
}

extern "C" int dsnt_maxpool2_fwd(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C,
                                 void* stream) {
    DSNT_REQUIRE(x && y && idx && N > 0 && H > 0 && W > 0 && C > 0, DSNT_ERR_ARG, "dsnt_maxpool2_fwd: bad argument");
    DSNT_REQUIRE(H % 2 == 0 && W % 2 == 0, DSNT_ERR_SHAPE, "dsnt_maxpool2_fwd: H and W must be even (got %dx%d)", H, W);
    DSNT_REQUIRE(C % 4 == 0 && dsnt_aligned16(x) && dsnt_aligned16(y) && (((uintptr_t)idx) & 3) == 0,
                 DSNT_ERR_ALIGN, "dsnt_maxpool2_fwd: alignment");
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    DSNT_LAUNCH(maxpool2_fwd_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x, (float4*)y, (uchar4*)idx, N, H, W, C / 4);
    DSNT_CHECK_LAUNCH("dsnt_maxpool2_fwd");
}

__global__ void maxpool2_bwd_kernel(PoolBwdP q) { maxpool2_bwd_body(q, blockIdx.x, gridDim.x); }

static int maxpool2_bwd_impl(const float* dy, const uint8_t* idx, float* dx, int accumulate, const float* extra, int N, int H, int W,
                             int C, float* amax, void* stream);
extern "C" int dsnt_maxpool2_bwd(const float* dy, const uint8_t* idx, float* dx, int accumulate,
                                 int N, int H, int W, int C, void* stream) {
    return maxpool2_bwd_impl(dy, idx, dx, accumulate, nullptr, N, H, W, C, nullptr, stream);
}
extern "C" int dsnt_maxpool2_bwd_amax(const float* dy, const uint8_t* idx, float* dx, int accumulate,
                                      int N, int H, int W, int C, float* amax, void* stream) {
    return maxpool2_bwd_impl(dy, idx, dx, accumulate, nullptr, N, H, W, C, amax, stream);
}
extern "C" int dsnt_maxpool2_bwd_add(const float* dy, const uint8_t* idx, float* dx, int accumulate, const float* extra,
                                     int N, int H, int W, int C, float* amax, void* stream) {
    DSNT_REQUIRE(extra && extra != dx && dsnt_aligned16(extra), DSNT_ERR_ARG, "dsnt_maxpool2_bwd_add: `extra` must be a second, aligned tensor");
    return maxpool2_bwd_impl(dy, idx, dx, accumulate, extra, N, H, W, C, amax, stream);
}
static int maxpool2_bwd_impl(const float* dy, const uint8_t* idx, float* dx, int accumulate, const float* extra, int N, int H, int W,
                             int C, float* amax, void* stream) {
    DSNT_REQUIRE(dy && idx && dx && N > 0 && H > 0 && W > 0 && C > 0, DSNT_ERR_ARG, "dsnt_maxpool2_bwd: bad argument");
    DSNT_REQUIRE(H % 2 == 0 && W % 2 == 0, DSNT_ERR_SHAPE, "dsnt_maxpool2_bwd: H and W must be even");
    DSNT_REQUIRE(C % 4 == 0 && dsnt_aligned16(dy) && dsnt_aligned16(dx), DSNT_ERR_ALIGN, "dsnt_maxpool2_bwd: alignment");
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    const PoolBwdP q{(const float4*)dy, (const uchar4*)idx, (float4*)dx, accumulate, (const float4*)extra, N, H, W, C / 4, (unsigned*)amax};
    DSNT_LAUNCH_OP(DSNT_ST_POOL_BWD, maxpool2_bwd_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, stream, q);
    DSNT_CHECK_LAUNCH("dsnt_maxpool2_bwd");
}

__global__ void upsample2_add_fwd_kernel(const float4* __restrict__ up, const float4* __restrict__ low,
                                         float4* __restrict__ out, int N, int H, int W, int C4) {
    const long total = (long)N * H * W * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % C4);
        long t = i / C4;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        const float4 a = up[i];
        const float4 b = low[(((long)n * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1)) * C4 + cg];
        out[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

extern "C" int dsnt_upsample2_add_fwd(const float* up, const float* low, float* out, int N, int H,
                                      int W, int C, void* stream) {
    DSNT_REQUIRE(up && low && out && N > 0 && H > 0 && W > 0 && C > 0, DSNT_ERR_ARG, "dsnt_upsample2_add_fwd: bad argument");
    DSNT_REQUIRE(H % 2 == 0 && W % 2 == 0, DSNT_ERR_SHAPE, "dsnt_upsample2_add_fwd: H and W must be even");
    DSNT_REQUIRE(C % 4 == 0 && dsnt_aligned16(up) && dsnt_aligned16(low) && dsnt_aligned16(out),
                 DSNT_ERR_ALIGN, "dsnt_upsample2_add_fwd: alignment");
    const long total = (long)N * H * W * (C / 4);
    DSNT_LAUNCH(upsample2_add_fwd_kernel, dim3(flat_grid(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)up, (const float4*)low, (float4*)out, N, H, W, C / 4);
    DSNT_CHECK_LAUNCH("dsnt_upsample2_add_fwd");
}

__global__ void upsample2_bwd_kernel(UpBwdP q) { upsample2_bwd_body(q, blockIdx.x, gridDim.x); }

static int upsample2_bwd_impl(const float* dout, float* dlow, int accumulate, int N, int H, int W, int C, float* amax,
                              void* stream);
extern "C" int dsnt_upsample2_bwd(const float* dout, float* dlow, int accumulate, int N, int H, int W,
                                  int C, void* stream) {
    return upsample2_bwd_impl(dout, dlow, accumulate, N, H, W, C, nullptr, stream);
}
extern "C" int dsnt_upsample2_bwd_amax(const float* dout, float* dlow, int accumulate, int N, int H, int W,
                                       int C, float* amax, void* stream) {
    return upsample2_bwd_impl(dout, dlow, accumulate, N, H, W, C, amax, stream);
}
static int upsample2_bwd_impl(const float* dout, float* dlow, int accumulate, int N, int H, int W, int C, float* amax,
                              void* stream) {
    DSNT_REQUIRE(dout && dlow && N > 0 && H > 0 && W > 0 && C > 0, DSNT_ERR_ARG, "dsnt_upsample2_bwd: bad argument");
    DSNT_REQUIRE(H % 2 == 0 && W % 2 == 0, DSNT_ERR_SHAPE, "dsnt_upsample2_bwd: H and W must be even");
    DSNT_REQUIRE(C % 4 == 0 && dsnt_aligned16(dout) && dsnt_aligned16(dlow), DSNT_ERR_ALIGN, "dsnt_upsample2_bwd: alignment");
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    const UpBwdP q{(const float4*)dout, (float4*)dlow, accumulate, N, H, W, C / 4, (unsigned*)amax};
    DSNT_LAUNCH_OP(DSNT_ST_UP_BWD, upsample2_bwd_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, stream, q);
    DSNT_CHECK_LAUNCH("dsnt_upsample2_bwd");
}

__global__ void axpy_kernel(const float* __restrict__ x, float* y, float a, int accumulate, long n, unsigned* amax) {
    const long n4 = n >> 2;
    float am = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        float4 o = make_float4(a * v.x, a * v.y, a * v.z, a * v.w);
        if (accumulate) {
            const float4 c = reinterpret_cast<float4*>(y)[i];
            o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w;
        }
        reinterpret_cast<float4*>(y)[i] = o;
        am = fmaxf(fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))), am);
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        const float o = accumulate ? y[i] + a * x[i] : a * x[i];
        y[i] = o;
        am = fmaxf(am, fabsf(o));
    }
    if (amax) amax_commit(am, amax);
}

static int axpy_impl(const float* x, float* y, float a, int accumulate, int64_t n, float* amax, void* stream) {
    DSNT_REQUIRE(x && y && n > 0, DSNT_ERR_ARG, "dsnt_axpy: bad argument");
    DSNT_REQUIRE(dsnt_aligned16(x) && dsnt_aligned16(y), DSNT_ERR_ALIGN, "dsnt_axpy: alignment");
    DSNT_LAUNCH(axpy_kernel, dim3(flat_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                       a, accumulate, (long)n, (unsigned*)amax);
    DSNT_CHECK_LAUNCH("dsnt_axpy");
}
extern "C" int dsnt_axpy(const float* x, float* y, float a, int accumulate, int64_t n, void* stream) {
    return axpy_impl(x, y, a, accumulate, n, nullptr, stream);
}
extern "C" int dsnt_axpy_amax(const float* x, float* y, float a, int accumulate, int64_t n, float* amax, void* stream) {
    return axpy_impl(x, y, a, accumulate, n, amax, stream);
}

// ---------------------------------------------------------------- layout changes
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C,
                                    int HW, int Cpad) {
    const long total = (long)N * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const long n = i / HW;
        for (int c = 0; c < Cpad; ++c)
            dst[i * Cpad + c] = c < C ? src[(n * C + c) * HW + p] : 0.f;
    }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C,
                                    int HW, int Cpad) {
    const long total = (long)N * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const long n = i / HW;
        for (int c = 0; c < C; ++c) dst[(n * C + c) * HW + p] = src[i * Cpad + c];
    }
}

extern "C" int dsnt_nchw_to_nhwc(const float* src, float* dst, int N, int C, int HW, int Cpad, void* stream) {
    DSNT_REQUIRE(src && dst && N > 0 && C > 0 && HW > 0 && Cpad >= C, DSNT_ERR_ARG, "dsnt_nchw_to_nhwc: bad argument");
    DSNT_LAUNCH(nchw_to_nhwc_kernel, dim3(flat_grid((long)N * HW, 256)), dim3(256), 0,
                       (hipStream_t)stream, src, dst, N, C, HW, Cpad);
    DSNT_CHECK_LAUNCH("dsnt_nchw_to_nhwc");
}
extern "C" int dsnt_nhwc_to_nchw(const float* src, float* dst, int N, int C, int HW, int Cpad, void* stream) {
    DSNT_REQUIRE(src && dst && N > 0 && C > 0 && HW > 0 && Cpad >= C, DSNT_ERR_ARG, "dsnt_nhwc_to_nchw: bad argument");
    DSNT_LAUNCH(nhwc_to_nchw_kernel, dim3(flat_grid((long)N * HW, 256)), dim3(256), 0,
                       (hipStream_t)stream, src, dst, N, C, HW, Cpad);
    DSNT_CHECK_LAUNCH("dsnt_nhwc_to_nchw");
}

// ---------------------------------------------------------------- 7x7 / stride 2 stem as a 4x4 / stride 1 convolution
// The stem (hourglass.py:106 / torchvision's conv1: 7x7, stride 2, pad 3 on a 3-channel image) has K = 7*7*4 = 196 with four
// channels per tap — nothing the 16-channel K-steps of the split-precision kernels can use, so it ran on the fp32 MFMA at a
// third of that pipe's peak.  Space-to-depth turns it into an ordinary convolution: 2x2 pixel blocks become 16 channels
// ((dy*2+dx)*4 + c), the 7x7 filter — extended by a zero row and column at the TOP / LEFT to 8x8 — becomes 4x4 block taps at
// block offsets -2..+1, and with one zero block row / column in front of the image that is a 4x4, stride 1, pad 1 convolution
// on [N][H/2+1][W/2+1][16]: K = 256 (23 % zeros), every large-tile fp16x3 / bf16x6 kernel applies.
//   dst[n][1+i][1+j][(dy*2+dx)*4 + c] = src[n][c][2i+dy][2j+dx]   (c < C <= 4; channel C..3, block row 0, block column 0: zero)
__global__ void s2d_input_kernel(const float* __restrict__ src, float4* __restrict__ dst, int N, int C, int H, int W,
                                 unsigned* __restrict__ amax) {
    const int Hb = H / 2 + 1, Wb = W / 2 + 1;
    const long total = (long)N * Hb * Wb * 4;             // one float4 (= one pixel of a block) per item
    float am = 0.f;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int q = (int)(t & 3);                       // dy*2 + dx
        long b = t >> 2;
        const int j = (int)(b % Wb); b /= Wb;
        const int i = (int)(b % Hb);
        const long n = b / Hb;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i > 0 && j > 0) {
            const int y = 2 * (i - 1) + (q >> 1), x = 2 * (j - 1) + (q & 1);
            const float* s0 = src + ((n * C) * H + y) * (long)W + x;
            v.x = s0[0];
            if (C > 1) v.y = s0[(long)H * W];
            if (C > 2) v.z = s0[2l * H * W];
            if (C > 3) v.w = s0[3l * H * W];
        }
        dst[t] = v;
        am = fmaxf(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))), am);
    }
    if (amax) amax_commit(am, amax);
}

extern "C" int dsnt_s2d_input(const float* src_nchw, float* dst, int N, int C, int H, int W, const dsnt_out_bounds* tail,
                              void* stream) {
    DSNT_REQUIRE(src_nchw && dst && N > 0 && C > 0 && C <= 4 && H > 0 && W > 0, DSNT_ERR_ARG, "dsnt_s2d_input: bad argument");
    DSNT_REQUIRE(H % 2 == 0 && W % 2 == 0, DSNT_ERR_SHAPE, "dsnt_s2d_input: H and W must be even (got %dx%d)", H, W);
    DSNT_REQUIRE(dsnt_aligned16(dst), DSNT_ERR_ALIGN, "dsnt_s2d_input: dst must be 16-byte aligned");
    const long total = (long)N * (H / 2 + 1) * (W / 2 + 1) * 4;
    DSNT_LAUNCH(s2d_input_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, src_nchw, (float4*)dst, N, C,
                H, W, (unsigned*)(tail ? tail->amax : nullptr));
    DSNT_CHECK_LAUNCH("dsnt_s2d_input");
}

// w2[co][R][S][(dy*2+dx)*4 + c] = w[co][2R+dy-1][2S+dx-1][c] (OHWI, 4 stored channels; 0 outside the 7x7);  back != 0: the
// inverse gather for the weight gradient, dw[co][r][s][c] = dw2[co][(r+1)/2][(s+1)/2][(((r+1)&1)*2 + ((s+1)&1))*4 + c].
__global__ void s2d_weights_kernel(const float* __restrict__ w, float* __restrict__ w2, int Cout, int back) {
    const int total = back ? Cout * 49 * 4 : Cout * 256;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        if (back) {
            const int c = t & 3, rs = (t >> 2) % 49, co = (t >> 2) / 49;
            const int r = rs / 7 + 1, s_ = rs % 7 + 1;
            w2[t] = w[((co * 4 + (r >> 1)) * 4 + (s_ >> 1)) * 16 + ((r & 1) * 2 + (s_ & 1)) * 4 + c];
        } else {
            const int k = t & 15, S = (t >> 4) & 3, R = (t >> 6) & 3, co = t >> 8;
            const int c = k & 3, dx = (k >> 2) & 1, dy = k >> 3;
            const int r = 2 * R + dy - 1, s_ = 2 * S + dx - 1;
            w2[t] = (r >= 0 && s_ >= 0) ? w[((co * 7 + r) * 7 + s_) * 4 + c] : 0.f;
        }
    }
}

extern "C" int dsnt_s2d_weights(const float* w, float* w2, int Cout, int back, void* stream) {
    DSNT_REQUIRE(w && w2 && Cout > 0, DSNT_ERR_ARG, "dsnt_s2d_weights: bad argument");
    const int total = back ? Cout * 196 : Cout * 256;
    DSNT_LAUNCH(s2d_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, w2, Cout, back);
    DSNT_CHECK_LAUNCH("dsnt_s2d_weights");
}

// ---------------------------------------------------------------- optimiser (flat arena)
// `flag` (nullable, int[2]): the step's non-finite guard.  flag[0] != 0 at kernel start = an earlier check on this
// stream fired (non-finite loss): nothing is updated.  A non-finite gradient element is skipped and raises
// DSNT_FLAG_GRAD in flag[1] (a second word, so that workgroups starting later do not see a half-raised flag[0]:
// the update stays an element-wise, order-independent function of its inputs).
__device__ __forceinline__ bool finite_f(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }

__global__ void rmsprop_kernel(float* p, const float* __restrict__ g, float* sq, long n, float lr,
                               float alpha, float eps, float wd, float gscale, int* flag) {
    if (flag && __builtin_nontemporal_load(flag) != 0) return;
    bool bad = false;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i] * gscale;
        if (flag && !finite_f(gi)) { bad = true; continue; }
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float s = alpha * sq[i] + (1.f - alpha) * gi * gi;
        sq[i] = s;
        p[i] = pi - lr * gi / (sqrtf(s) + eps);
    }
    if (bad) atomicOr(flag + 1, DSNT_FLAG_GRAD);
}
static int rmsprop_impl(float* p, const float* g, float* square_avg, int64_t n, float lr, float alpha, float eps,
                        float weight_decay, float grad_scale, int* flag, void* stream) {
    DSNT_REQUIRE(p && g && square_avg && n > 0, DSNT_ERR_ARG, "dsnt_rmsprop_step: bad argument");
    DSNT_LAUNCH(rmsprop_kernel, dim3(flat_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g,
                       square_avg, (long)n, lr, alpha, eps, weight_decay, grad_scale, flag);
    DSNT_CHECK_LAUNCH("dsnt_rmsprop_step");
}
extern "C" int dsnt_rmsprop_step(float* p, const float* g, float* square_avg, int64_t n, float lr,
                                 float alpha, float eps, float weight_decay, float grad_scale, void* stream) {
    return rmsprop_impl(p, g, square_avg, n, lr, alpha, eps, weight_decay, grad_scale, nullptr, stream);
}
extern "C" int dsnt_rmsprop_step_guarded(float* p, const float* g, float* square_avg, int64_t n, float lr, float alpha,
                                         float eps, float weight_decay, float grad_scale, int* flag, void* stream) {
    DSNT_REQUIRE(flag, DSNT_ERR_ARG, "dsnt_rmsprop_step_guarded: null flag");
    return rmsprop_impl(p, g, square_avg, n, lr, alpha, eps, weight_decay, grad_scale, flag, stream);
}

__global__ void sgd_kernel(float* p, const float* __restrict__ g, float* buf, long n, float lr,
                           float momentum, float wd, float gscale, int first, int* flag) {
    if (flag && __builtin_nontemporal_load(flag) != 0) return;
    bool bad = false;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i] * gscale;
        if (flag && !finite_f(gi)) { bad = true; continue; }
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        if (buf) {
            const float b = first ? gi : momentum * buf[i] + gi;
            buf[i] = b;
            gi = b;
        }
        p[i] = pi - lr * gi;
    }
    if (bad) atomicOr(flag + 1, DSNT_FLAG_GRAD);
}
static int sgd_impl(float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum,
                    float weight_decay, float grad_scale, int first_step, int* flag, void* stream) {
    DSNT_REQUIRE(p && g && n > 0, DSNT_ERR_ARG, "dsnt_sgd_step: bad argument");
    DSNT_LAUNCH(sgd_kernel, dim3(flat_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g,
                       momentum != 0.f ? momentum_buf : nullptr, (long)n, lr, momentum, weight_decay,
                       grad_scale, first_step, flag);
    DSNT_CHECK_LAUNCH("dsnt_sgd_step");
}
extern "C" int dsnt_sgd_step(float* p, const float* g, float* momentum_buf, int64_t n, float lr,
                             float momentum, float weight_decay, float grad_scale, int first_step,
                             void* stream) {
    return sgd_impl(p, g, momentum_buf, n, lr, momentum, weight_decay, grad_scale, first_step, nullptr, stream);
}
extern "C" int dsnt_sgd_step_guarded(float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum,
                                     float weight_decay, float grad_scale, int first_step, int* flag, void* stream) {
    DSNT_REQUIRE(flag, DSNT_ERR_ARG, "dsnt_sgd_step_guarded: null flag");
    return sgd_impl(p, g, momentum_buf, n, lr, momentum, weight_decay, grad_scale, first_step, flag, stream);
}

__global__ void nonfinite_flag_kernel(const float* __restrict__ x, long n, int* flag, int code) {
    bool bad = false;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        bad |= !finite_f(x[i]);
    if (bad) atomicOr(flag, code);
    // a gradient flag left by the previous step's optimiser becomes blocking from this step on
    if (blockIdx.x == 0 && threadIdx.x == 0 && flag[1] != 0) atomicOr(flag, flag[1]);
}
extern "C" int dsnt_nonfinite_flag(const float* x, int64_t n, int* flag, int code, void* stream) {
    DSNT_REQUIRE(x && flag && n > 0 && code != 0, DSNT_ERR_ARG, "dsnt_nonfinite_flag: bad argument");
    DSNT_LAUNCH(nonfinite_flag_kernel, dim3(flat_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, (long)n,
                       flag, code);
    DSNT_CHECK_LAUNCH("dsnt_nonfinite_flag");
}

// ---------------------------------------------------------------- PCKh hits
__global__ void pckh_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                            const double* __restrict__ m, const double* __restrict__ b,
                            const float* __restrict__ mask, const double* __restrict__ head,
                            float thr, float* hits, float* valid, int B, int J) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * J) return;
    const int n = i / J;
    const double* mm = m + (size_t)n * 4;
    const double* bb = b + (size_t)n * 2;
    const double px = pred[2 * i], py = pred[2 * i + 1], tx = target[2 * i], ty = target[2 * i + 1];
    // row-vector times matrix plus offset (train.py:243-258: bmm(norm, transform_m) + transform_b)
    const double ox = px * mm[0] + py * mm[2] + bb[0], oy = px * mm[1] + py * mm[3] + bb[1];
    const double gx = tx * mm[0] + ty * mm[2] + bb[0], gy = tx * mm[1] + ty * mm[3] + bb[1];
    const double d = sqrt((ox - gx) * (ox - gx) + (oy - gy) * (oy - gy)) / head[n];
    const bool v = mask[i] == 1.f;
    valid[i] = v ? 1.f : 0.f;
    hits[i] = (v && d <= (double)thr) ? 1.f : 0.f;
}
extern "C" int dsnt_pckh(const float* pred, const float* target, const double* m, const double* b,
                         const float* mask, const double* head, float threshold, float* hits,
                         float* valid, int B, int J, void* stream) {
    DSNT_REQUIRE(pred && target && m && b && mask && head && hits && valid && B > 0 && J > 0,
                 DSNT_ERR_ARG, "dsnt_pckh: bad argument");
    DSNT_LAUNCH(pckh_kernel, dim3((B * J + 255) / 256), dim3(256), 0, (hipStream_t)stream, pred,
                       target, m, b, mask, head, threshold, hits, valid, B, J);
    DSNT_CHECK_LAUNCH("dsnt_pckh");
}

// ---------------------------------------------------------------- ResNet pieces
// 3x3 / stride 2 / pad 1 max-pool (torchvision resnet.maxpool; reference model.py:123 keeps it in `fcn`).
// idx = winning tap 0..8 in scan order (first maximum wins, NaN propagates: ATen's max_pool2d rule).
__global__ void maxpool3s2_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ y,
                                      uchar4* __restrict__ idx, int N, int H, int W, int Ho, int Wo, int C4) {
    const long total = (long)N * Ho * Wo * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % C4);
        long t = i / C4;
        const int ow = (int)(t % Wo); t /= Wo;
        const int oh = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        uchar4 k = make_uchar4(255, 255, 255, 255);
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            const int ih = 2 * oh - 1 + p / 3, iw = 2 * ow - 1 + p % 3;
            if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
            const float4 v = x[(((long)n * H + ih) * W + iw) * C4 + cg];
            // ATen: `if (val > maxval || isnan(val))`, maxindex starts at the window's first valid element
            if (k.x == 255) k.x = p;
            if (k.y == 255) k.y = p;
            if (k.z == 255) k.z = p;
            if (k.w == 255) k.w = p;
            if (v.x > m.x || v.x != v.x) { m.x = v.x; k.x = p; }
            if (v.y > m.y || v.y != v.y) { m.y = v.y; k.y = p; }
            if (v.z > m.z || v.z != v.z) { m.z = v.z; k.z = p; }
            if (v.w > m.w || v.w != v.w) { m.w = v.w; k.w = p; }
        }
        y[i] = m;
        idx[i] = k;
    }
}

extern "C" int dsnt_maxpool3s2_fwd(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, void* stream) {
    DSNT_REQUIRE(x && y && idx && N > 0 && H > 0 && W > 0 && C > 0, DSNT_ERR_ARG, "dsnt_maxpool3s2_fwd: bad argument");
    DSNT_REQUIRE(C % 4 == 0 && dsnt_aligned16(x) && dsnt_aligned16(y) && (((uintptr_t)idx) & 3) == 0,
                 DSNT_ERR_ALIGN, "dsnt_maxpool3s2_fwd: alignment");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;        // floor((H + 2 - 3) / 2) + 1
    const long total = (long)N * Ho * Wo * (C / 4);
    DSNT_LAUNCH(maxpool3s2_fwd_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x, (float4*)y, (uchar4*)idx, N, H, W, Ho, Wo, C / 4);
    DSNT_CHECK_LAUNCH("dsnt_maxpool3s2_fwd");
}

// Gather form of the backward: every input pixel sums the gradients of the (at most 2 x 2) windows that
// picked it — no atomics, deterministic.
__global__ void maxpool3s2_bwd_kernel(const float4* __restrict__ dy, const uchar4* __restrict__ idx, float4* dx,
                                      int accumulate, int N, int H, int W, int Ho, int Wo, int C4) {
    const long total = (long)N * H * W * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % C4);
        long t = i / C4;
        const int iw = (int)(t % W); t /= W;
        const int ih = (int)(t % H);
        const int n = (int)(t / H);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        // windows with 2*oh - 1 <= ih <= 2*oh + 1
        for (int oh = ih / 2; oh <= (ih + 1) / 2; ++oh) {
            if (oh >= Ho) continue;
            const int r = ih - (2 * oh - 1);
            for (int ow = iw / 2; ow <= (iw + 1) / 2; ++ow) {
                if (ow >= Wo) continue;
                const int p = r * 3 + (iw - (2 * ow - 1));
                const long o = (((long)n * Ho + oh) * Wo + ow) * C4 + cg;
                const uchar4 k = idx[o];
                const float4 v = dy[o];
                if (k.x == p) g.x += v.x;
                if (k.y == p) g.y += v.y;
                if (k.z == p) g.z += v.z;
                if (k.w == p) g.w += v.w;
            }
        }
        if (accumulate) { const float4 c = dx[i]; g.x += c.x; g.y += c.y; g.z += c.z; g.w += c.w; }
        dx[i] = g;
    }
}

extern "C" int dsnt_maxpool3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int accumulate, int N, int H,
                                   int W, int C, void* stream) {
    DSNT_REQUIRE(dy && idx && dx && N > 0 && H > 0 && W > 0 && C > 0, DSNT_ERR_ARG, "dsnt_maxpool3s2_bwd: bad argument");
    DSNT_REQUIRE(C % 4 == 0 && dsnt_aligned16(dy) && dsnt_aligned16(dx), DSNT_ERR_ALIGN, "dsnt_maxpool3s2_bwd: alignment");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)N * H * W * (C / 4);
    DSNT_LAUNCH(maxpool3s2_bwd_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)dy, (const uchar4*)idx, (float4*)dx, accumulate, N, H, W, Ho, Wo, C / 4);
    DSNT_CHECK_LAUNCH("dsnt_maxpool3s2_bwd");
}

// y = relu?(scale * x + shift + res): the tail of a torchvision BasicBlock / Bottleneck (bn -> += identity -> relu)
__global__ void bn_add_act_fwd_kernel(const float4* __restrict__ x, const float4* __restrict__ scale,
                                      const float4* __restrict__ shift, const float4* __restrict__ res, int relu,
                                      float4* __restrict__ y, long n4, int C4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % C4);
        const float4 v = x[i], sc = scale[cg], sh = shift[cg], r = res[i];
        float4 o = make_float4(fmaf(v.x, sc.x, sh.x) + r.x, fmaf(v.y, sc.y, sh.y) + r.y, fmaf(v.z, sc.z, sh.z) + r.z,
                               fmaf(v.w, sc.w, sh.w) + r.w);
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        y[i] = o;
    }
}

extern "C" int dsnt_bn_add_act_fwd(const float* x, const float* scale, const float* shift, const float* res,
                                   int relu, float* y, int64_t M, int C, void* stream) {
    DSNT_REQUIRE(x && scale && shift && res && y && M > 0 && C > 0, DSNT_ERR_ARG, "dsnt_bn_add_act_fwd: bad argument");
    DSNT_REQUIRE(C % 4 == 0 && dsnt_aligned16(x) && dsnt_aligned16(y) && dsnt_aligned16(res) && dsnt_aligned16(scale) &&
                 dsnt_aligned16(shift), DSNT_ERR_ALIGN, "dsnt_bn_add_act_fwd: alignment");
    const long n4 = (long)M * C / 4;
    DSNT_LAUNCH(bn_add_act_fwd_kernel, dim3(flat_grid(n4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x, (const float4*)scale, (const float4*)shift, (const float4*)res, relu,
                       (float4*)y, n4, C / 4);
    DSNT_CHECK_LAUNCH("dsnt_bn_add_act_fwd");
}

// dz = dy where y > 0, else 0 (backward of the block-output ReLU; ATen's threshold_backward keeps dy for y > 0)
__global__ void relu_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ y, float4* __restrict__ dz,
                                long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 g = dy[i], v = y[i];
        dz[i] = make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f);
    }
}

extern "C" int dsnt_relu_bwd(const float* dy, const float* y, float* dz, int64_t n, void* stream) {
    DSNT_REQUIRE(dy && y && dz && n > 0 && n % 4 == 0, DSNT_ERR_ARG, "dsnt_relu_bwd: bad argument");
    DSNT_REQUIRE(dsnt_aligned16(dy) && dsnt_aligned16(y) && dsnt_aligned16(dz), DSNT_ERR_ALIGN, "dsnt_relu_bwd: alignment");
    DSNT_LAUNCH(relu_bwd_kernel, dim3(flat_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)dy, (const float4*)y, (float4*)dz, (long)(n / 4));
    DSNT_CHECK_LAUNCH("dsnt_relu_bwd");
}

// out[n][oh*s][ow*s][c] = dy[n][oh][ow][c], zeros elsewhere (out is [N][Hs][Ws][C]): the data gradient of a
// stride-s convolution is the stride-1 data gradient of this zero-stuffed tensor.
__global__ void zero_insert_kernel(const float4* __restrict__ dy, float4* __restrict__ out, int N, int Ho, int Wo,
                                   int Hs, int Ws, int s, int C4) {
    const long total = (long)N * Hs * Ws * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % C4);
        long t = i / C4;
        const int w = (int)(t % Ws); t /= Ws;
        const int h = (int)(t % Hs);
        const int n = (int)(t / Hs);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h % s == 0 && w % s == 0 && h / s < Ho && w / s < Wo)
            v = dy[(((long)n * Ho + h / s) * Wo + w / s) * C4 + cg];
        out[i] = v;
    }
}

extern "C" int dsnt_zero_insert(const float* dy, float* out, int N, int Ho, int Wo, int C, int Hs, int Ws, int stride,
                                void* stream) {
    DSNT_REQUIRE(dy && out && N > 0 && Ho > 0 && Wo > 0 && C > 0 && stride >= 1, DSNT_ERR_ARG, "dsnt_zero_insert: bad argument");
    DSNT_REQUIRE(Hs >= (Ho - 1) * stride + 1 && Ws >= (Wo - 1) * stride + 1, DSNT_ERR_SHAPE,
                 "dsnt_zero_insert: %dx%d does not hold %dx%d at stride %d", Hs, Ws, Ho, Wo, stride);
    DSNT_REQUIRE(C % 4 == 0 && dsnt_aligned16(dy) && dsnt_aligned16(out), DSNT_ERR_ALIGN, "dsnt_zero_insert: alignment");
    const long total = (long)N * Hs * Ws * (C / 4);
    DSNT_LAUNCH(zero_insert_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)dy, (float4*)out, N, Ho, Wo, Hs, Ws, stride, C / 4);
    DSNT_CHECK_LAUNCH("dsnt_zero_insert");
}
