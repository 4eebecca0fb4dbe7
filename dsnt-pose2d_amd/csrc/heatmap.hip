// Heat-map matching ("gauss" output strategy, the hourglass builder's default): target rendering, the
// mean-squared-error loss against it and its gradient, and arg-max decoding of predicted heat-maps.
// Replaces /root/reference/src/dsnt/util.py:70-198 (draw_gaussian / encode_heatmaps / get_preds /
// decode_heatmaps: CPU loops over B x J maps with a D2H copy per step) and the loss of
// /root/reference/src/dsnt/model.py:147-156, 247-258.  HBM-bound, one 256-thread workgroup per
// (image, joint) row of H*W floats; the 7x7 target bump is evaluated in-register, never materialised
// for the loss.
#include "common.h"
#include <math.h>

// The reference's coordinate arithmetic is a chain of separately rounded fp32 tensor ops (add_, mul_, add_);
// hipcc contracts a*b+c into one FMA by default (and HIP's __fmul_rn / __fadd_rn are plain operators), which
// moves results by an ulp and can move a rounded pixel.  This file is compiled with -ffp-contract=off
// (build.py; HIP's default -ffp-contract=fast ignores `#pragma clang fp contract`).

#define HB 256

// Centre of the target bump of a row: util.py:133-146 — pixel = (c + 1) * (W/2) - 0.5 in fp32, three separately
// rounded operations (no FMA contraction), then Python round() = round-half-even; draw_gaussian (:70-126) with
// clip_size 7 (radius 3.5) skips bumps whose centre is more than 3.5 px outside the map.
struct Bump {
    int x, y, x0, x1, y0, y1;   // centre and clipped window [x0, x1) x [y0, y1); empty window when skipped
};
__device__ __forceinline__ Bump bump_of(const float* __restrict__ target, long row, int h, int w) {
    // w / 2 and h / 2 are exact in fp32 (integers or halves)
    const float cx = target[2 * row], cy = target[2 * row + 1];
    const float px = __fadd_rn(__fmul_rn(__fadd_rn(cx, 1.f), (float)(w / 2.0)), -0.5f);
    const float py = __fadd_rn(__fmul_rn(__fadd_rn(cy, 1.f), (float)(h / 2.0)), -0.5f);
    Bump b;
    b.x = (int)rintf(px); b.y = (int)rintf(py);
    const float radius = 3.5f;
    const bool skip = !(px == px) || !(py == py) || b.x <= -radius || b.y <= -radius ||
                      b.x >= (w - 1) + radius || b.y >= (h - 1) + radius;
    b.x0 = max(0, b.x - 3); b.x1 = min(w, b.x + 4);
    b.y0 = max(0, b.y - 3); b.y1 = min(h, b.y + 4);
    if (skip) { b.x0 = b.x1 = 0; b.y0 = b.y1 = 0; }
    return b;
}
__device__ __forceinline__ float bump_at(const Bump& b, int r, int c, float k) {
    if (c < b.x0 || c >= b.x1 || r < b.y0 || r >= b.y1) return 0.f;
    const float dx = (float)(c - b.x), dy = (float)(r - b.y);
    return expf(__fmul_rn(__fadd_rn(dx * dx, dy * dy), k));        // small integers: dx*dx, dy*dy exact
}

__global__ __launch_bounds__(HB) void encode_heatmaps_kernel(const float* __restrict__ target, float* __restrict__ out,
                                                              int h, int w, float k) {
    const long row = blockIdx.x;
    const Bump b = bump_of(target, row, h, w);
    float* o = out + (size_t)row * h * w;
    for (int i = threadIdx.x; i < h * w; i += HB) {
        const int r = i / w, c = i - r * w;
        o[i] = bump_at(b, r, c, k);
    }
}

// per_row[row] = sum_i (hm[i] - g[i])^2
__global__ __launch_bounds__(HB) void heatmap_mse_fwd_kernel(const float* __restrict__ hm, const float* __restrict__ target,
                                                              float* __restrict__ per_row, int h, int w, float k) {
    __shared__ float red[8];
    const long row = blockIdx.x;
    const Bump b = bump_of(target, row, h, w);
    const float* x = hm + (size_t)row * h * w;
    float s[1] = {0.f};
    for (int i = threadIdx.x; i < h * w; i += HB) {
        const int r = i / w, c = i - r * w;
        const float d = x[i] - bump_at(b, r, c, k);
        s[0] = fmaf(d, d, s[0]);
    }
    block_sum<1>(s, red);
    if (threadIdx.x == 0) per_row[row] = s[0];
}

// dhm[i] = gscale * coef * (hm[i] - g[i]),  coef = 2 / numel
__global__ __launch_bounds__(HB) void heatmap_mse_bwd_kernel(const float* __restrict__ hm, const float* __restrict__ target,
                                                              const float* __restrict__ gscale, float* __restrict__ dhm,
                                                              int h, int w, float k, float coef) {
    const long row = blockIdx.x;
    const Bump b = bump_of(target, row, h, w);
    const float* x = hm + (size_t)row * h * w;
    float* o = dhm + (size_t)row * h * w;
    const float gs = gscale[0] * coef;
    for (int i = threadIdx.x; i < h * w; i += HB) {
        const int r = i / w, c = i - r * w;
        o[i] = gs * (x[i] - bump_at(b, r, c, k));
    }
}

// util.py:150-198: first arg-max pixel, (0, 0) when the maximum is not positive, +-0.25 px towards the larger
// neighbour for interior pixels, then pixel -> normalised coordinates ((p + 0.5) * (2/size) - 1, fp32 steps).
__global__ __launch_bounds__(HB) void decode_heatmaps_kernel(const float* __restrict__ hm, float* __restrict__ coords,
                                                              int h, int w, int use_neighbours, float sx, float sy) {
    __shared__ float rv[4];
    __shared__ int ri[4];
    const long row = blockIdx.x;
    const float* x = hm + (size_t)row * h * w;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < h * w; i += HB) {
        const float v = x[i];
        if (v > best) { best = v; bi = i; }            // ascending i per thread: keeps the first maximum
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { rv[wave] = best; ri[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < HB / 64; ++wv)
            if (rv[wv] > best || (rv[wv] == best && ri[wv] < bi)) { best = rv[wv]; bi = ri[wv]; }
        if (bi == 0x7fffffff) bi = 0;                    // all -inf / NaN rows
        float px = (float)(bi % w), py = (float)(bi / h);          // `idx / height` as the reference (util.py:161)
        if (!(best > 0.f)) { px = 0.f; py = 0.f; }
        if (use_neighbours) {
            const int xi = (int)px, yi = (int)py;
            if (xi > 0 && xi < w - 1 && yi > 0 && yi < h - 1) {
                const float dxn = x[yi * w + xi + 1] - x[yi * w + xi - 1];
                const float dyn = x[(yi + 1) * w + xi] - x[(yi - 1) * w + xi];
                px += 0.25f * (float)((dxn > 0.f) - (dxn < 0.f));
                py += 0.25f * (float)((dyn > 0.f) - (dyn < 0.f));
            }
        }
        // sx = float(2 / w), sy = float(2 / h) rounded on the host exactly like the reference's Python scalars
        coords[2 * row + 0] = __fadd_rn(__fmul_rn(__fadd_rn(px, 0.5f), sx), -1.f);
        coords[2 * row + 1] = __fadd_rn(__fmul_rn(__fadd_rn(py, 0.5f), sy), -1.f);
    }
}

static int check_rows_hm(const char* who, int64_t rows, int h, int w, float sigma) {
    DSNT_REQUIRE(rows > 0 && rows < (1LL << 31), DSNT_ERR_SHAPE, "%s: rows=%lld out of range", who, (long long)rows);
    DSNT_REQUIRE(h > 0 && w > 0 && (long)h * w < (1L << 24), DSNT_ERR_SHAPE, "%s: bad map size %dx%d", who, h, w);
    DSNT_REQUIRE(sigma > 0.f, DSNT_ERR_ARG, "%s: sigma must be positive", who);
    return DSNT_OK;
}
static inline float bump_k(float sigma) { return (float)(-0.5 * (1.0 / (double)sigma) * (1.0 / (double)sigma)); }

extern "C" int dsnt_encode_heatmaps(const float* target, float* out, int64_t rows, int h, int w, float sigma,
                                    void* stream) {
    DSNT_REQUIRE(target && out, DSNT_ERR_ARG, "dsnt_encode_heatmaps: null tensor");
    if (int e = check_rows_hm("dsnt_encode_heatmaps", rows, h, w, sigma)) return e;
    DSNT_LAUNCH(encode_heatmaps_kernel, dim3((unsigned)rows), dim3(HB), 0, (hipStream_t)stream, target, out,
                       h, w, bump_k(sigma));
    DSNT_CHECK_LAUNCH("dsnt_encode_heatmaps");
}

extern "C" int dsnt_heatmap_mse_fwd(const float* hm, const float* target, float* per_row, int64_t rows, int h,
                                    int w, float sigma, void* stream) {
    DSNT_REQUIRE(hm && target && per_row, DSNT_ERR_ARG, "dsnt_heatmap_mse_fwd: null tensor");
    if (int e = check_rows_hm("dsnt_heatmap_mse_fwd", rows, h, w, sigma)) return e;
    DSNT_LAUNCH(heatmap_mse_fwd_kernel, dim3((unsigned)rows), dim3(HB), 0, (hipStream_t)stream, hm, target,
                       per_row, h, w, bump_k(sigma));
    DSNT_CHECK_LAUNCH("dsnt_heatmap_mse_fwd");
}

extern "C" int dsnt_heatmap_mse_bwd(const float* hm, const float* target, const float* gscale, float* dhm,
                                    int64_t rows, int h, int w, float sigma, void* stream) {
    DSNT_REQUIRE(hm && target && gscale && dhm, DSNT_ERR_ARG, "dsnt_heatmap_mse_bwd: null tensor");
    if (int e = check_rows_hm("dsnt_heatmap_mse_bwd", rows, h, w, sigma)) return e;
    const float coef = (float)(2.0 / ((double)rows * h * w));
    DSNT_LAUNCH(heatmap_mse_bwd_kernel, dim3((unsigned)rows), dim3(HB), 0, (hipStream_t)stream, hm, target,
                       gscale, dhm, h, w, bump_k(sigma), coef);
    DSNT_CHECK_LAUNCH("dsnt_heatmap_mse_bwd");
}

extern "C" int dsnt_decode_heatmaps(const float* hm, float* coords, int64_t rows, int h, int w, int use_neighbours,
                                    void* stream) {
    DSNT_REQUIRE(hm && coords, DSNT_ERR_ARG, "dsnt_decode_heatmaps: null tensor");
    if (int e = check_rows_hm("dsnt_decode_heatmaps", rows, h, w, 1.f)) return e;
    DSNT_LAUNCH(decode_heatmaps_kernel, dim3((unsigned)rows), dim3(HB), 0, (hipStream_t)stream, hm, coords, h,
                       w, use_neighbours, (float)(2.0 / (double)w), (float)(2.0 / (double)h));
    DSNT_CHECK_LAUNCH("dsnt_decode_heatmaps");
}
