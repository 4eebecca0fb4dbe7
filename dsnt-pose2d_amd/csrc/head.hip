// DSNT head kernels: heat-map normalisation (spatial softmax and its variants), coordinate
// expectation against the X/Y meshgrids, Gaussian targets, JS / KL / MSE / variance
// regularisers, Euclidean loss, masked average — and the fused forward / backward of the whole
// head.  HBM-bound: one 256-thread workgroup per (image, joint) row of H*W floats; the row is
// read from HBM once and kept in registers (16 floats per thread for H*W <= 4096, 16-byte loads
// when H*W % 4 == 0) while wavefront reductions produce the softmax denominator, the
// coordinate moments and the divergence sums.  Longer rows fall back to re-reading the row
// (served by L2).  Meshgrids are never materialised: x_w = (2w - (W-1))/W, y_h = (2h - (H-1))/H.
#include "common.h"
#include <math.h>

#define HB 256   // threads per row

template <int VEC, bool CACHED>
struct Row {
    float v[16];
    const float* src;
    int hw;
    __device__ __forceinline__ void load(const float* row, int n) {
        src = row; hw = n;
        if (CACHED) {
            const int tid = threadIdx.x;
            if (VEC == 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = (k * HB + tid) * 4;
                    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (i < hw) t = *reinterpret_cast<const float4*>(row + i);
                    v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int i = k * HB + tid;
                    v[k] = i < hw ? row[i] : 0.f;
                }
            }
        }
    }
    // f(slot, index, value) for every element this thread owns (CACHED rows only): slot = 0..15 is a compile-time
    // constant after unrolling, so per-element temporaries indexed by it live in registers
    template <typename F>
    __device__ __forceinline__ void each_slot(F f) const {
        const int tid = threadIdx.x;
        if (VEC == 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = (k * HB + tid) * 4;
                if (i < hw) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) f(4 * k + e, i + e, v[4 * k + e]);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = k * HB + tid;
                if (i < hw) f(k, i, v[k]);
            }
        }
    }
    // f(index, value) for every element this thread owns
    template <typename F>
    __device__ __forceinline__ void each(F f) const {
        const int tid = threadIdx.x;
        if (CACHED) {
            if (VEC == 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = (k * HB + tid) * 4;
                    if (i < hw) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) f(i + e, v[4 * k + e]);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int i = k * HB + tid;
                    if (i < hw) f(i, v[k]);
                }
            }
        } else {
            for (int i = tid; i < hw; i += HB) f(i, src[i]);
        }
    }
};

// exp(x) for x <= 0 on the hardware exp2 (v_exp_f32, <= 1 ulp) with a compensated x * log2(e): the product is formed
// as t + r with t = fl(x L), r = fma(x, L, -t) + x L_lo, so the argument error (|x| 2^-24 for a plain multiply: 2e-6
// relative at x = -30) does not reach the result: e^x = 2^t (1 + r ln 2).  ~6 instructions instead of libm's ~25.
__device__ __forceinline__ float fast_exp(float x) {
    const float L = 1.44269502162933349609375f, Ll = 1.92596299112661746e-8f;
    const float t = x * L;
    const float r = fmaf(x, Ll, fmaf(x, L, -t));
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, r * 0.69314718055994530942f, e);
}

struct Grid2 {
    int W, H; float offx, offy;
    __device__ __forceinline__ Grid2(int h, int w)
        : W(w), H(h), offx((float)(w - 1)), offy((float)(h - 1)) {}
    __device__ __forceinline__ void xy(int i, float& x, float& y) const {
        const int r = i / W, c = i - r * W;
        x = (2.f * c - offx) / (float)W;   // exact closed form of linspace(-(W-1)/W, (W-1)/W, W)
        y = (2.f * r - offy) / (float)H;
    }
};

// ------------------------------------------------------------------ preact (model.py:24-45)
template <int VEC, bool CACHED>
__global__ __launch_bounds__(HB) void preact_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         int hw, int mode, float thr, float eps) {
    __shared__ float red[16];
    const size_t off = (size_t)blockIdx.x * hw;
    Row<VEC, CACHED> row;
    row.load(x + off, hw);
    float* out = y + off;
    if (mode <= 1) {
        float m = -INFINITY;
        row.each([&](int, float v) { m = fmaxf(m, v); });
        m = block_max(m, red);
        float s[1] = {0.f};
        row.each([&](int, float v) {
            const float e = expf(v - m);
            s[0] += (mode == 1 && !(v >= thr)) ? 0.f : e;
        });
        block_sum<1>(s, red);
        const float denom = mode == 1 ? s[0] + eps : s[0];
        row.each([&](int i, float v) {
            const float e = (mode == 1 && !(v >= thr)) ? 0.f : expf(v - m);
            out[i] = e / denom;
        });
    } else {
        auto f = [&](float v) {
            return mode == 2 ? fabsf(v) : mode == 3 ? fmaxf(v, 0.f) : 1.f / (1.f + expf(-v));
        };
        float s[1] = {0.f};
        row.each([&](int, float v) { s[0] += f(v); });
        block_sum<1>(s, red);
        const float denom = s[0] + eps;
        row.each([&](int i, float v) { out[i] = f(v) / denom; });
    }
}

template <int VEC, bool CACHED>
__global__ __launch_bounds__(HB) void preact_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ gy, float* __restrict__ gx,
                                                         int hw, int mode, float eps) {
    __shared__ float red[16];
    const size_t off = (size_t)blockIdx.x * hw;
    Row<VEC, CACHED> py;
    py.load(y + off, hw);
    const float* g = gy + off;
    float* out = gx + off;
    if (mode <= 1) {
        float s[1] = {0.f};
        py.each([&](int i, float p) { s[0] = fmaf(p, g[i], s[0]); });
        block_sum<1>(s, red);
        py.each([&](int i, float p) { out[i] = p * (g[i] - s[0]); });
    } else {
        const float* xr = x + off;
        auto f = [&](float v) {
            return mode == 2 ? fabsf(v) : mode == 3 ? fmaxf(v, 0.f) : 1.f / (1.f + expf(-v));
        };
        float s[2] = {0.f, 0.f};   // sum f(x), sum g*y
        py.each([&](int i, float p) { s[0] += f(xr[i]); s[1] = fmaf(p, g[i], s[1]); });
        block_sum<2>(s, red);
        const float inv = 1.f / (s[0] + eps);
        py.each([&](int i, float) {
            const float v = xr[i];
            float d;
            if (mode == 2) d = v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
            else if (mode == 3) d = v > 0.f ? 1.f : 0.f;
            else { const float sg = 1.f / (1.f + expf(-v)); d = sg * (1.f - sg); }
            out[i] = d * (g[i] - s[1]) * inv;
        });
    }
}

// ------------------------------------------------------------------ dsnt (nn.py:25-78)
template <int VEC, bool CACHED>
__global__ __launch_bounds__(HB) void expect_fwd_kernel(const float* __restrict__ hm, float* __restrict__ coords,
                                                         int h, int w) {
    __shared__ float red[16];
    const int hw = h * w;
    Row<VEC, CACHED> row;
    row.load(hm + (size_t)blockIdx.x * hw, hw);
    const Grid2 g(h, w);
    float s[2] = {0.f, 0.f};
    row.each([&](int i, float p) {
        float x, y; g.xy(i, x, y);
        s[0] = fmaf(x, p, s[0]); s[1] = fmaf(y, p, s[1]);
    });
    block_sum<2>(s, red);
    if (threadIdx.x == 0) { coords[2 * (size_t)blockIdx.x] = s[0]; coords[2 * (size_t)blockIdx.x + 1] = s[1]; }
}

__global__ __launch_bounds__(HB) void expect_bwd_kernel(const float* __restrict__ gc, float* __restrict__ ghm,
                                                         int h, int w) {
    const int hw = h * w;
    const float gx = gc[2 * (size_t)blockIdx.x], gy = gc[2 * (size_t)blockIdx.x + 1];
    float* out = ghm + (size_t)blockIdx.x * hw;
    const Grid2 g(h, w);
    for (int i = threadIdx.x; i < hw; i += HB) {
        float x, y; g.xy(i, x, y);
        out[i] = gx * x + gy * y;
    }
}

// ------------------------------------------------------------------ make_gauss (nn.py:168-205)
__global__ __launch_bounds__(HB) void make_gauss_kernel(const float* __restrict__ coords, float* __restrict__ out,
                                                         int h, int w, float k) {
    __shared__ float red[16];
    const int hw = h * w;
    const float mx = coords[2 * (size_t)blockIdx.x], my = coords[2 * (size_t)blockIdx.x + 1];
    float* o = out + (size_t)blockIdx.x * hw;
    const Grid2 g(h, w);
    float s[1] = {0.f};
    for (int i = threadIdx.x; i < hw; i += HB) {
        float x, y; g.xy(i, x, y);
        s[0] += expf(((x - mx) * (x - mx) + (y - my) * (y - my)) * k);
    }
    block_sum<1>(s, red);
    const float z = s[0] + 1e-24f;
    for (int i = threadIdx.x; i < hw; i += HB) {
        float x, y; g.xy(i, x, y);
        o[i] = expf(((x - mx) * (x - mx) + (y - my) * (y - my)) * k) / z;
    }
}

// d/d(mu) of the above (the reference's make_gauss is differentiable in `coords`, nn.py:180-203):
//   g_i = e_i / Z,  d g_i / d mu_x = g_i ((x_i - mu_x) - sum_j g_j (x_j - mu_x)) / sigma^2
//   dL/d mu_x = (sum_i G_i g_i (x_i - mu_x) - m_x sum_i G_i g_i) / sigma^2,  m_x = sum_j g_j (x_j - mu_x)
// The Gaussian is re-evaluated in registers (one pass for Z, one over the incoming gradient G): one HBM read.
__global__ __launch_bounds__(HB) void make_gauss_bwd_kernel(const float* __restrict__ coords, const float* __restrict__ gout,
                                                             float* __restrict__ gcoords, int h, int w, float k) {
    __shared__ float red[32];
    const int hw = h * w;
    const float mx = coords[2 * (size_t)blockIdx.x], my = coords[2 * (size_t)blockIdx.x + 1];
    const float* G = gout + (size_t)blockIdx.x * hw;
    const Grid2 g(h, w);
    float z[1] = {0.f};
    for (int i = threadIdx.x; i < hw; i += HB) {
        float x, y; g.xy(i, x, y);
        z[0] += expf(((x - mx) * (x - mx) + (y - my) * (y - my)) * k);
    }
    block_sum<1>(z, red);
    const float inv = 1.f / (z[0] + 1e-24f);
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};   // sum G g, sum G g dx, sum G g dy, sum g dx, sum g dy
    for (int i = threadIdx.x; i < hw; i += HB) {
        float x, y; g.xy(i, x, y);
        const float dx = x - mx, dy = y - my;
        const float q = expf((dx * dx + dy * dy) * k) * inv;
        const float gq = G[i] * q;
        s[0] += gq; s[1] = fmaf(gq, dx, s[1]); s[2] = fmaf(gq, dy, s[2]);
        s[3] = fmaf(q, dx, s[3]); s[4] = fmaf(q, dy, s[4]);
    }
    block_sum<5>(s, red);
    if (threadIdx.x == 0) {
        const float is2 = -2.f * k;            // 1 / sigma^2
        gcoords[2 * (size_t)blockIdx.x] = (s[1] - s[3] * s[0]) * is2;
        gcoords[2 * (size_t)blockIdx.x + 1] = (s[2] - s[4] * s[0]) * is2;
    }
}

// ------------------------------------------------------------------ regularisers (nn.py:208-298)
#define REG_EPS 1e-24f

// Per-row context shared by forward and backward: Gaussian normaliser, or the moments for `var`.
struct RegCtx { float z, mx, my, sp, vx, vy; };

template <typename ROW>
__device__ __forceinline__ RegCtx reg_context(const ROW& row, const Grid2& g, float tx, float ty, float k,
                                              int kind, float* red) {
    RegCtx c = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (kind == 3) {
        float s[3] = {0.f, 0.f, 0.f};
        row.each([&](int i, float p) {
            float x, y; g.xy(i, x, y);
            s[0] = fmaf(x, p, s[0]); s[1] = fmaf(y, p, s[1]); s[2] += p;
        });
        block_sum<3>(s, red);
        c.mx = s[0]; c.my = s[1]; c.sp = s[2];
        float v[2] = {0.f, 0.f};
        row.each([&](int i, float p) {
            float x, y; g.xy(i, x, y);
            v[0] = fmaf((x - c.mx) * (x - c.mx), p, v[0]);
            v[1] = fmaf((y - c.my) * (y - c.my), p, v[1]);
        });
        block_sum<2>(v, red);
        c.vx = v[0]; c.vy = v[1];
    } else {
        float s[1] = {0.f};
        row.each([&](int i, float) {
            float x, y; g.xy(i, x, y);
            s[0] += expf(((x - tx) * (x - tx) + (y - ty) * (y - ty)) * k);
        });
        block_sum<1>(s, red);
        c.z = s[0] + 1e-24f;
    }
    return c;
}

template <typename ROW>
__device__ __forceinline__ float reg_value(const ROW& row, const Grid2& g, const RegCtx& c, float tx,
                                           float ty, float k, float sigma, int kind, float* red) {
    if (kind == 3) {
        const float s2 = sigma * sigma;
        return (c.vx - s2) * (c.vx - s2) + (c.vy - s2) * (c.vy - s2);
    }
    float s[1] = {0.f};
    row.each([&](int i, float p) {
        float x, y; g.xy(i, x, y);
        const float q = expf(((x - tx) * (x - tx) + (y - ty) * (y - ty)) * k) / c.z;
        if (kind == 0) {
            const float m = 0.5f * (p + q), lm = logf(m + REG_EPS);
            s[0] += 0.5f * (p * (logf(p + REG_EPS) - lm)) + 0.5f * (q * (logf(q + REG_EPS) - lm));
        } else if (kind == 1) {
            s[0] += p * (logf(p + REG_EPS) - logf(q + REG_EPS));
        } else {
            s[0] += (p - q) * (p - q);
        }
    });
    block_sum<1>(s, red);
    return s[0];
}

// d(reg)/d(p_i)
__device__ __forceinline__ float reg_grad(float p, float x, float y, const RegCtx& c, float tx, float ty,
                                          float k, float sigma, int kind) {
    if (kind == 3) {
        const float s2 = sigma * sigma;
        const float dvx = (x - c.mx) * (x - c.mx) - 2.f * x * c.mx * (1.f - c.sp);
        const float dvy = (y - c.my) * (y - c.my) - 2.f * y * c.my * (1.f - c.sp);
        return 2.f * (c.vx - s2) * dvx + 2.f * (c.vy - s2) * dvy;
    }
    const float q = expf(((x - tx) * (x - tx) + (y - ty) * (y - ty)) * k) / c.z;
    if (kind == 0) {
        const float m = 0.5f * (p + q);
        return 0.5f * (logf(p + REG_EPS) - logf(m + REG_EPS) + p / (p + REG_EPS) - m / (m + REG_EPS));
    }
    if (kind == 1) return logf(p + REG_EPS) - logf(q + REG_EPS) + p / (p + REG_EPS);
    return 2.f * (p - q);
}

template <int VEC, bool CACHED>
__global__ __launch_bounds__(HB) void reg_fwd_kernel(const float* __restrict__ hm, const float* __restrict__ target,
                                                      float* __restrict__ per_row, int h, int w, float sigma,
                                                      float k, int kind) {
    __shared__ float red[16];
    const int hw = h * w;
    Row<VEC, CACHED> row;
    row.load(hm + (size_t)blockIdx.x * hw, hw);
    const Grid2 g(h, w);
    float tx = 0.f, ty = 0.f;
    if (kind != 3) { tx = target[2 * (size_t)blockIdx.x]; ty = target[2 * (size_t)blockIdx.x + 1]; }
    const RegCtx c = reg_context(row, g, tx, ty, k, kind, red);
    const float val = reg_value(row, g, c, tx, ty, k, sigma, kind, red);
    if (threadIdx.x == 0) per_row[blockIdx.x] = val;
}

template <int VEC, bool CACHED>
__global__ __launch_bounds__(HB) void reg_bwd_kernel(const float* __restrict__ hm, const float* __restrict__ target,
                                                      const float* __restrict__ g_row, float* __restrict__ ghm,
                                                      int h, int w, float sigma, float k, int kind) {
    __shared__ float red[16];
    const int hw = h * w;
    Row<VEC, CACHED> row;
    row.load(hm + (size_t)blockIdx.x * hw, hw);
    const Grid2 g(h, w);
    float tx = 0.f, ty = 0.f;
    if (kind != 3) { tx = target[2 * (size_t)blockIdx.x]; ty = target[2 * (size_t)blockIdx.x + 1]; }
    const RegCtx c = reg_context(row, g, tx, ty, k, kind, red);
    const float gr = g_row[blockIdx.x];
    float* out = ghm + (size_t)blockIdx.x * hw;
    row.each([&](int i, float p) {
        float x, y; g.xy(i, x, y);
        out[i] = gr * reg_grad(p, x, y, c, tx, ty, k, sigma, kind);
    });
}

// d(reg row)/d(mu_t): the reference builds its target as make_gauss(mu_t, ...) inside autograd (nn.py:219-271), so
// kl / js / mse are differentiable in the target means.  With D_i = d div / d q_i (the divergence's derivative in the
// TARGET pixel) this is make_gauss's backward with G = D, composed in registers: one read of the heat-map, nothing
// materialised.   d/d mu_x = (sum_i D_i q_i dx_i - (sum_j q_j dx_j)(sum_i D_i q_i)) / sigma^2
template <int VEC, bool CACHED>
__global__ __launch_bounds__(HB) void reg_bwd_mu_kernel(const float* __restrict__ hm, const float* __restrict__ target,
                                                         const float* __restrict__ g_row, float* __restrict__ gmu,
                                                         int h, int w, float k, int kind) {
    __shared__ float red[32];
    const int hw = h * w;
    Row<VEC, CACHED> row;
    row.load(hm + (size_t)blockIdx.x * hw, hw);
    const Grid2 g(h, w);
    const float tx = target[2 * (size_t)blockIdx.x], ty = target[2 * (size_t)blockIdx.x + 1];
    const RegCtx c = reg_context(row, g, tx, ty, k, kind, red);
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};   // sum D q, sum D q dx, sum D q dy, sum q dx, sum q dy
    row.each([&](int i, float p) {
        float x, y; g.xy(i, x, y);
        const float dx = x - tx, dy = y - ty;
        const float q = expf((dx * dx + dy * dy) * k) / c.z;
        float D;
        if (kind == 0) {
            const float m = 0.5f * (p + q);
            D = 0.5f * (logf(q + REG_EPS) - logf(m + REG_EPS) + q / (q + REG_EPS) - m / (m + REG_EPS));
        } else if (kind == 1) {
            D = -p / (q + REG_EPS);
        } else {
            D = -2.f * (p - q);
        }
        const float dq = D * q;
        s[0] += dq; s[1] = fmaf(dq, dx, s[1]); s[2] = fmaf(dq, dy, s[2]);
        s[3] = fmaf(q, dx, s[3]); s[4] = fmaf(q, dy, s[4]);
    });
    block_sum<5>(s, red);
    if (threadIdx.x == 0) {
        const float f = g_row[blockIdx.x] * (-2.f * k);          // upstream gradient / sigma^2
        gmu[2 * (size_t)blockIdx.x] = (s[1] - s[3] * s[0]) * f;
        gmu[2 * (size_t)blockIdx.x + 1] = (s[2] - s[4] * s[0]) * f;
    }
}

// ------------------------------------------------------------------ euclid / masked average
__global__ void euclid_fwd_kernel(const float* __restrict__ a, const float* __restrict__ t, float* __restrict__ dist,
                                  long n, int d) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int j = 0; j < d; ++j) { const float df = a[i * d + j] - t[i * d + j]; s += df * df; }
    dist[i] = sqrtf(s);
}
__global__ void euclid_bwd_kernel(const float* __restrict__ a, const float* __restrict__ t,
                                  const float* __restrict__ dist, const float* __restrict__ gd,
                                  float* __restrict__ ga, long n, int d) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // d sqrt(s)/da = (a-t)/dist; at dist == 0 the reference's autograd gives 0 * inf = NaN
    const float f = gd[i] / (2.f * dist[i]);
    for (int j = 0; j < d; ++j) ga[i * d + j] = f * (2.f * (a[i * d + j] - t[i * d + j]));
}

__global__ __launch_bounds__(HB) void masked_avg_fwd_kernel(const float* __restrict__ l, const float* __restrict__ m,
                                                             float* __restrict__ out2, long n) {
    __shared__ float red[16];
    float s[2] = {0.f, 0.f};
    for (long i = threadIdx.x; i < n; i += HB) {
        const float w = m ? m[i] : 1.f;
        s[0] += m ? l[i] * w : l[i];
        s[1] += w;
    }
    block_sum<2>(s, red);
    if (threadIdx.x == 0) {
        const float denom = fmaxf(s[1], 1.f);
        out2[0] = s[0] / denom;
        out2[1] = denom;
    }
}
__global__ void masked_avg_bwd_kernel(const float* __restrict__ g, const float* __restrict__ m,
                                      const float* __restrict__ out2, float* __restrict__ gl, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    gl[i] = g[0] * (m ? m[i] : 1.f) / out2[1];
}

// ------------------------------------------------------------------ fused head
// forward: softmax over the row + coordinate moments, one HBM read of the logits.
template <int VEC, bool CACHED>
__global__ __launch_bounds__(HB) void head_fwd_kernel(const float* __restrict__ logits, float* __restrict__ hm,
                                                       float* __restrict__ coords, int h, int w) {
    __shared__ float red[16];
    const int hw = h * w;
    const size_t off = (size_t)blockIdx.x * hw;
    Row<VEC, CACHED> row;
    row.load(logits + off, hw);
    float m = -INFINITY;
    row.each([&](int, float v) { m = fmaxf(m, v); });
    m = block_max(m, red);
    float s[1] = {0.f};
    if (CACHED) {
        // one exponential per element: they replace the logits in the row registers
#pragma unroll
        for (int k = 0; k < 16; ++k) row.v[k] = fast_exp(row.v[k] - m);
        row.each([&](int, float e) { s[0] += e; });
    } else {
        row.each([&](int, float v) { s[0] += expf(v - m); });
    }
    block_sum<1>(s, red);
    const float denom = s[0];
    const Grid2 g(h, w);
    float* out = hm + off;
    float c[2] = {0.f, 0.f};
    if (CACHED && VEC == 4 && (w & 3) == 0) {
        // four consecutive pixels of one heat-map row per 16-byte store: one division for the position, the
        // normalisation as a multiplication by 1 / sum (one more rounding than e / sum: <= 1 ulp)
        const float inv = 1.f / denom;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = (k * HB + threadIdx.x) * 4;
            if (i < hw) {
                const int rr = i / w, cc = i - rr * w;
                const float y = (2.f * rr - g.offy) / (float)h;
                float p[4], px = 0.f, ps = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p[e] = row.v[4 * k + e] * inv;
                    const float x = (2.f * (cc + e) - g.offx) / (float)w;
                    px = fmaf(x, p[e], px);
                    ps += p[e];
                }
                c[0] += px; c[1] = fmaf(y, ps, c[1]);
                *reinterpret_cast<float4*>(out + i) = make_float4(p[0], p[1], p[2], p[3]);
            }
        }
    } else if (CACHED && VEC == 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = (k * HB + threadIdx.x) * 4;
            if (i < hw) {
                float p[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p[e] = row.v[4 * k + e] / denom;
                    float x, y; g.xy(i + e, x, y);
                    c[0] = fmaf(x, p[e], c[0]); c[1] = fmaf(y, p[e], c[1]);
                }
                *reinterpret_cast<float4*>(out + i) = make_float4(p[0], p[1], p[2], p[3]);
            }
        }
    } else {
        row.each([&](int i, float v) {
            const float p = (CACHED ? v : expf(v - m)) / denom;
            out[i] = p;
            float x, y; g.xy(i, x, y);
            c[0] = fmaf(x, p, c[0]); c[1] = fmaf(y, p, c[1]);
        });
    }
    block_sum<2>(c, red);
    if (threadIdx.x == 0) { coords[2 * (size_t)blockIdx.x] = c[0]; coords[2 * (size_t)blockIdx.x + 1] = c[1]; }
}

template <int VEC, bool CACHED>
__global__ __launch_bounds__(HB) void head_loss_rows_kernel(const float* __restrict__ hm, const float* __restrict__ coords,
                                                             const float* __restrict__ target, float* __restrict__ dist,
                                                             float* __restrict__ reg_row, int h, int w, float sigma,
                                                             float k, int kind) {
    __shared__ float red[16];
    const size_t r = blockIdx.x;
    const float tx = target[2 * r], ty = target[2 * r + 1];
    if (threadIdx.x == 0) {
        const float dx = coords[2 * r] - tx, dy = coords[2 * r + 1] - ty;
        dist[r] = sqrtf(dx * dx + dy * dy);
    }
    if (kind < 0) return;
    const int hw = h * w;
    Row<VEC, CACHED> row;
    row.load(hm + r * hw, hw);
    const Grid2 g(h, w);
    const RegCtx c = reg_context(row, g, tx, ty, k, kind, red);
    const float val = reg_value(row, g, c, tx, ty, k, sigma, kind, red);
    if (threadIdx.x == 0) reg_row[r] = val;
}

// backward: dL/dp_i = g_dist*((mu-t)/dist . (x_i,y_i)) + g_reg * dreg/dp_i, then softmax backward
// dz_i = p_i (dL/dp_i - sum_j p_j dL/dp_j).  One read of the saved heat-map, one write.
template <int VEC, bool CACHED>
__global__ __launch_bounds__(HB) void head_bwd_kernel(const float* __restrict__ hm, const float* __restrict__ coords,
                                                       const float* __restrict__ target, const float* __restrict__ dist,
                                                       const float* __restrict__ g_dist, const float* __restrict__ g_reg,
                                                       float* __restrict__ g_logits, int h, int w, float sigma, float k,
                                                       int kind) {
    __shared__ float red[16];
    const size_t r = blockIdx.x;
    const int hw = h * w;
    Row<VEC, CACHED> row;
    row.load(hm + r * hw, hw);
    const Grid2 g(h, w);
    const float tx = target[2 * r], ty = target[2 * r + 1];
    const float d = dist[r], gd = g_dist[r];
    // un-guarded like the reference: dist == 0 with gd != 0 gives NaN
    const float f = gd / (2.f * d);
    const float ax = f * (2.f * (coords[2 * r] - tx));
    const float ay = f * (2.f * (coords[2 * r + 1] - ty));
    const float gr = (kind >= 0 && g_reg) ? g_reg[r] : 0.f;
    RegCtx c = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (kind >= 0 && gr != 0.f) c = reg_context(row, g, tx, ty, k, kind, red);
    auto dLdp = [&](int i, float p) {
        float x, y; g.xy(i, x, y);
        float v = ax * x + ay * y;
        if (kind >= 0 && gr != 0.f) v += gr * reg_grad(p, x, y, c, tx, ty, k, sigma, kind);
        return v;
    };
    float s[1] = {0.f};
    row.each([&](int i, float p) { s[0] = fmaf(p, dLdp(i, p), s[0]); });
    block_sum<1>(s, red);
    float* out = g_logits + r * hw;
    row.each([&](int i, float p) { out[i] = p * (dLdp(i, p) - s[0]); });
}

// ------------------------------------------------------------------ fused head loss + its gradient (train step)
// One pass over the saved heat-maps produces, per row, the Euclidean distance, the regulariser value AND
//   G0[row] = d( w_row * (dist + reg_coeff * reg) ) / d logits,   w_row = mask_row / clamp(sum mask, 1)
// i.e. the gradient of this stack's loss for an upstream gradient of 1 (what `loss.backward()` sends;
// dsnt_scale_by_scalar applies any other value): the train step's head is then 4 HBM passes per stack — logits in,
// heat-maps out (dsnt_head_fwd), heat-maps in, d logits out — instead of 5 (the heat-maps were read once for the
// loss rows and once more in backward).  Per-element values are computed ONCE and kept in registers between the
// reduction sum_j p_j dL/dp_j and the output pass.  JS (the regulariser of BASELINE configs 3-5) has a fast form:
// the target Gaussian is separable, exp(k((x-tx)^2 + (y-ty)^2)) = ex[w] * ey[h] — W + H exponentials per row in LDS
// instead of H*W — and the three logarithms per element use the hardware log2 (v_log_f32, <= 1 ulp in log2;
// arguments are >= 1e-24, normal numbers): with libm logf the kernel was ALU-bound at twice its HBM time.
#define HEAD_SEP_MAX 512          // W + H up to which the separable factors fit the static LDS array
#define LN2 0.69314718055994530942f

template <int VEC>
__global__ __launch_bounds__(HB) void head_loss_grad_kernel(const float* __restrict__ hm, const float* __restrict__ coords,
                                                            const float* __restrict__ target, const float* __restrict__ mask,
                                                            const float* __restrict__ denom_p, float* __restrict__ dist_out,
                                                            float* __restrict__ reg_out, float* __restrict__ g0, int h, int w,
                                                            float sigma, float k, int kind, float reg_coeff) {
    __shared__ float red[16];
    __shared__ __attribute__((aligned(16))) float exy[HEAD_SEP_MAX];
    __shared__ __attribute__((aligned(16))) float pos[HEAD_SEP_MAX];
    const size_t r = blockIdx.x;
    const int hw = h * w;
    // the row's scalars first, then the 16 KB row: everything is in flight together
    const float tx = target[2 * r], ty = target[2 * r + 1];
    const float cx = coords[2 * r], cy = coords[2 * r + 1];
    const float mk = mask ? mask[r] : 1.f;
    const float den = denom_p[1];
    Row<VEC, true> row;
    row.load(hm + r * hw, hw);
    const Grid2 g(h, w);
    const float dxm = cx - tx, dym = cy - ty;
    const float d = sqrtf(dxm * dxm + dym * dym);
    const float wm = mk / den;
    // un-guarded like the reference: dist == 0 with wm != 0 gives NaN (nn.py:113-114)
    const float f = wm / (2.f * d);
    const float ax = f * (2.f * dxm), ay = f * (2.f * dym);
    const float gr = kind >= 0 ? wm * reg_coeff : 0.f;
    float gv[16];
    float acc[2] = {0.f, 0.f};              // regulariser value, sum_j p_j dL/dp_j
    // the pixel positions x_w = (2w - (W-1)) / W, y_h = (2h - (H-1)) / H once per workgroup (W + H true divisions) instead of
    // one or two per element: round 6 — with them in the element loop the kernel was ALU-bound at 1.4-1.8x its HBM time
    const bool tab = w + h <= HEAD_SEP_MAX;
    if (tab) {
        for (int i = threadIdx.x; i < w + h; i += HB)
            pos[i] = i < w ? (2.f * i - g.offx) / (float)w : (2.f * (i - w) - g.offy) / (float)h;
    }
    if (kind == 0 && tab) {
        // separable target Gaussian: ex[0..w), ey[0..h)
        for (int i = threadIdx.x; i < w + h; i += HB) {
            const float t = (i < w ? (2.f * i - g.offx) / (float)w - tx : (2.f * (i - w) - g.offy) / (float)h - ty);
            exy[i] = expf(t * t * k);
        }
        __syncthreads();
        float sxy[2] = {0.f, 0.f};
        for (int i = threadIdx.x; i < w + h; i += HB) sxy[i < w ? 0 : 1] += exy[i];
        block_sum<2>(sxy, red);
        const float invz = 1.f / (sxy[0] * sxy[1] + 1e-24f);
        auto finish = [&](int slot, float p, float x, float y, float dr) {
            const float v = fmaf(ax, x, ay * y) + gr * dr;
            gv[slot] = v;
            acc[1] = fmaf(p, v, acc[1]);
        };
        auto elem = [&](int slot, float p, float x, float y, float q) {
            const float m = 0.5f * (p + q);
            // logarithms in the log2 domain (v_log_f32), ln 2 folded into the two places they are used
            const float lm = __builtin_amdgcn_logf(m + REG_EPS);
            const float dp = __builtin_amdgcn_logf(p + REG_EPS) - lm, dq = __builtin_amdgcn_logf(q + REG_EPS) - lm;
            acc[0] = fmaf(0.5f * LN2, fmaf(p, dp, q * dq), acc[0]);
            finish(slot, p, x, y, 0.5f * (fmaf(LN2, dp, p * __builtin_amdgcn_rcpf(p + REG_EPS)) - m * __builtin_amdgcn_rcpf(m + REG_EPS)));
        };
        if (VEC == 4 && (w & 3) == 0) {
            // four consecutive pixels of one heat-map row per thread and chunk: one division for the position
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int i = (kk * HB + threadIdx.x) * 4;
                if (i < hw) {
                    const int rr = i / w, cc = i - rr * w;
                    const float y = pos[w + rr];
                    const float qy = exy[w + rr] * invz;
                    const float4 qx = *reinterpret_cast<const float4*>(exy + cc);
                    const float4 xv = *reinterpret_cast<const float4*>(pos + cc);
                    const float qs[4] = {qx.x * qy, qx.y * qy, qx.z * qy, qx.w * qy};
                    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
                    const float* pv = row.v + 4 * kk;
                    // The five transcendentals per element (3 v_log_f32, 2 v_rcp_f32: quarter rate) are the kernel's ALU time; two
                    // wave-uniform shortcuts that are EXACT in fp32 take most of them away:
                    //  * every p of the wave's 256 pixels > 1e-16: p / (p + 1e-24) and m / (m + 1e-24) (m >= p / 2) are 1 - <= 2e-8,
                    //    which rounds to 1.0f — the two reciprocals drop out, d reg / d p = (ln 2 / 2) (log2(p + eps) - log2(m + eps));
                    //  * ... and every q < 1e-30 (the target Gaussian has underflowed: all but ~26 of the 64 rows at sigma = 1 px):
                    //    m = p / 2 exactly, log2(p + eps) - log2(m + eps) = 1, q (..) < 1e-28 — no transcendental at all.
                    const float pmin = fminf(fminf(pv[0], pv[1]), fminf(pv[2], pv[3]));
                    const float qmax = fmaxf(fmaxf(qs[0], qs[1]), fmaxf(qs[2], qs[3]));
                    const bool big = pmin > 1e-16f, far = qmax < 1e-30f;
                    if (__all(big && far)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[0] = fmaf(0.5f * LN2, pv[e], acc[0]);
                            finish(4 * kk + e, pv[e], xs[e], y, 0.5f * LN2);
                        }
                    } else if (__all(big)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float pe = pv[e], q = qs[e];
                            const float m = 0.5f * (pe + q);
                            const float lm = __builtin_amdgcn_logf(m + REG_EPS);
                            const float dp = __builtin_amdgcn_logf(pe + REG_EPS) - lm, dq = __builtin_amdgcn_logf(q + REG_EPS) - lm;
                            acc[0] = fmaf(0.5f * LN2, fmaf(pe, dp, q * dq), acc[0]);
                            finish(4 * kk + e, pe, xs[e], y, (0.5f * LN2) * dp);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) elem(4 * kk + e, pv[e], xs[e], y, qs[e]);
                    }
                }
            }
        } else {
            row.each_slot([&](int slot, int i, float p) {
                const int rr = i / w, cc = i - rr * w;
                elem(slot, p, pos[cc], pos[w + rr], exy[cc] * (exy[w + rr] * invz));
            });
        }
        block_sum<2>(acc, red);
    } else {
        RegCtx c = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float val = 0.f;
        if (kind >= 0) {
            c = reg_context(row, g, tx, ty, k, kind, red);
            val = reg_value(row, g, c, tx, ty, k, sigma, kind, red);
        }
        if (tab) __syncthreads();               // `pos` (written above; reg_context / reg_value may not have synchronised)
        auto one = [&](int slot, float p, float x, float y) {
            float v = ax * x + ay * y;
            if (kind >= 0 && gr != 0.f) v += gr * reg_grad(p, x, y, c, tx, ty, k, sigma, kind);
            gv[slot] = v;
            acc[1] = fmaf(p, v, acc[1]);
        };
        if (VEC == 4 && (w & 3) == 0 && tab) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int i = (kk * HB + threadIdx.x) * 4;
                if (i < hw) {
                    const int rr = i / w, cc = i - rr * w;
                    const float y = pos[w + rr];
                    const float4 xv = *reinterpret_cast<const float4*>(pos + cc);
                    one(4 * kk, row.v[4 * kk], xv.x, y); one(4 * kk + 1, row.v[4 * kk + 1], xv.y, y);
                    one(4 * kk + 2, row.v[4 * kk + 2], xv.z, y); one(4 * kk + 3, row.v[4 * kk + 3], xv.w, y);
                }
            }
        } else {
            row.each_slot([&](int slot, int i, float p) {
                float x, y; g.xy(i, x, y);
                one(slot, p, x, y);
            });
        }
        float s1[1] = {acc[1]};
        block_sum<1>(s1, red);
        acc[0] = val; acc[1] = s1[0];
    }
    float* out = g0 + r * hw;
    if (VEC == 4) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int i = (kk * HB + threadIdx.x) * 4;
            if (i < hw)
                *reinterpret_cast<float4*>(out + i) = make_float4(row.v[4 * kk] * (gv[4 * kk] - acc[1]), row.v[4 * kk + 1] * (gv[4 * kk + 1] - acc[1]),
                                                                  row.v[4 * kk + 2] * (gv[4 * kk + 2] - acc[1]), row.v[4 * kk + 3] * (gv[4 * kk + 3] - acc[1]));
        }
    } else {
        row.each_slot([&](int slot, int i, float p) { out[i] = p * (gv[slot] - acc[1]); });
    }
    if (threadIdx.x == 0) {
        dist_out[r] = d;
        if (reg_out) reg_out[r] = acc[0];
    }
}

// out2 = {sum m / max(sum m, 1) [unused], max(sum m, 1)} for a mask (or n for no mask): the denominator of
// masked_average (nn.py:81-94) as a device scalar for the kernel above
__global__ __launch_bounds__(HB) void mask_denom_kernel(const float* __restrict__ m, float* __restrict__ out2, long n) {
    __shared__ float red[16];
    float s[1] = {0.f};
    for (long i = threadIdx.x; i < n; i += HB) s[0] += m ? m[i] : 1.f;
    block_sum<1>(s, red);
    if (threadIdx.x == 0) { out2[0] = s[0]; out2[1] = fmaxf(s[0], 1.f); }
}

// loss[0] = sum(dist m) / denom + reg_coeff sum(reg m) / denom; e2 = {sum(dist m) / denom, denom} (what
// dsnt_masked_avg_fwd leaves for its backward): the two masked averages and their combination in one launch
__global__ __launch_bounds__(HB) void head_loss_reduce_kernel(const float* __restrict__ dist, const float* __restrict__ reg,
                                                              const float* __restrict__ m, const float* __restrict__ denom2,
                                                              float reg_coeff, float* __restrict__ loss, float* __restrict__ e2,
                                                              long n) {
    __shared__ float red[16];
    float s[2] = {0.f, 0.f};
    for (long i = threadIdx.x; i < n; i += HB) {
        const float wgt = m ? m[i] : 1.f;
        s[0] += m ? dist[i] * wgt : dist[i];
        if (reg) s[1] += m ? reg[i] * wgt : reg[i];
    }
    block_sum<2>(s, red);
    if (threadIdx.x == 0) {
        const float den = denom2[1];
        const float a = s[0] / den;
        e2[0] = a; e2[1] = den;
        loss[0] = reg ? a + reg_coeff * (s[1] / den) : a;
    }
}

// x *= s[0] unless s[0] == 1 (then the kernel returns at once: the usual `loss.backward()` costs no memory pass)
__global__ void scale_by_scalar_kernel(float4* __restrict__ x, const float* __restrict__ s, long n4) {
    const float a = s[0];
    if (a == 1.f) return;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = x[i];
        v.x *= a; v.y *= a; v.z *= a; v.w *= a;
        x[i] = v;
    }
}

// ------------------------------------------------------------------ host wrappers
#define ROW_DISPATCH(KERNEL, rows, hw, ptr_ok, ...)                                                   \
    do {                                                                                               \
        const bool vec_ = ((hw) % 4 == 0) && (ptr_ok);                                                 \
        const bool cached_ = (hw) <= 4096;                                                             \
        hipStream_t st_ = (hipStream_t)stream;                                                         \
        if (vec_ && cached_) DSNT_LAUNCH((KERNEL<4, true>), dim3(rows), dim3(HB), 0, st_, __VA_ARGS__);   \
        else if (cached_) DSNT_LAUNCH((KERNEL<1, true>), dim3(rows), dim3(HB), 0, st_, __VA_ARGS__);      \
        else DSNT_LAUNCH((KERNEL<1, false>), dim3(rows), dim3(HB), 0, st_, __VA_ARGS__);                  \
    } while (0)

static int check_rows(const char* who, int64_t rows, int h, int w) {
    DSNT_REQUIRE(rows > 0 && rows < (1LL << 31), DSNT_ERR_SHAPE, "%s: rows=%lld out of range", who, (long long)rows);
    DSNT_REQUIRE(h > 0 && w > 0 && (long)h * w < (1L << 24), DSNT_ERR_SHAPE, "%s: bad map size %dx%d", who, h, w);
    return DSNT_OK;
}

extern "C" int dsnt_preact_fwd(const float* x, float* y, int64_t rows, int hw, int mode, float threshold,
                               float eps, void* stream) {
    DSNT_REQUIRE(x && y, DSNT_ERR_ARG, "dsnt_preact_fwd: null tensor");
    DSNT_REQUIRE(mode >= 0 && mode <= 4, DSNT_ERR_ARG, "dsnt_preact_fwd: unknown mode %d", mode);
    if (int e = check_rows("dsnt_preact_fwd", rows, 1, hw)) return e;
    ROW_DISPATCH(preact_fwd_kernel, (int)rows, hw, dsnt_aligned16(x), x, y, hw, mode, threshold, eps);
    DSNT_CHECK_LAUNCH("dsnt_preact_fwd");
}

extern "C" int dsnt_preact_bwd(const float* x, const float* y, const float* gy, float* gx, int64_t rows,
                               int hw, int mode, float threshold, float eps, void* stream) {
    (void)threshold;
    DSNT_REQUIRE(y && gy && gx && (mode <= 1 || x), DSNT_ERR_ARG, "dsnt_preact_bwd: null tensor");
    DSNT_REQUIRE(mode >= 0 && mode <= 4, DSNT_ERR_ARG, "dsnt_preact_bwd: unknown mode %d", mode);
    if (int e = check_rows("dsnt_preact_bwd", rows, 1, hw)) return e;
    ROW_DISPATCH(preact_bwd_kernel, (int)rows, hw, dsnt_aligned16(y), x, y, gy, gx, hw, mode, eps);
    DSNT_CHECK_LAUNCH("dsnt_preact_bwd");
}

extern "C" int dsnt_expect_fwd(const float* hm, float* coords, int64_t rows, int h, int w, void* stream) {
    DSNT_REQUIRE(hm && coords, DSNT_ERR_ARG, "dsnt_expect_fwd: null tensor");
    if (int e = check_rows("dsnt_expect_fwd", rows, h, w)) return e;
    ROW_DISPATCH(expect_fwd_kernel, (int)rows, h * w, dsnt_aligned16(hm), hm, coords, h, w);
    DSNT_CHECK_LAUNCH("dsnt_expect_fwd");
}

extern "C" int dsnt_expect_bwd(const float* gcoords, float* ghm, int64_t rows, int h, int w, void* stream) {
    DSNT_REQUIRE(gcoords && ghm, DSNT_ERR_ARG, "dsnt_expect_bwd: null tensor");
    if (int e = check_rows("dsnt_expect_bwd", rows, h, w)) return e;
    DSNT_LAUNCH(expect_bwd_kernel, dim3((int)rows), dim3(HB), 0, (hipStream_t)stream, gcoords, ghm, h, w);
    DSNT_CHECK_LAUNCH("dsnt_expect_bwd");
}

extern "C" int dsnt_make_gauss(const float* coords, float* out, int64_t rows, int h, int w, float sigma,
                               void* stream) {
    DSNT_REQUIRE(coords && out, DSNT_ERR_ARG, "dsnt_make_gauss: null tensor");
    DSNT_REQUIRE(sigma > 0.f, DSNT_ERR_ARG, "dsnt_make_gauss: sigma must be positive");
    if (int e = check_rows("dsnt_make_gauss", rows, h, w)) return e;
    const float k = (float)(-0.5 * (1.0 / (double)sigma) * (1.0 / (double)sigma));
    DSNT_LAUNCH(make_gauss_kernel, dim3((int)rows), dim3(HB), 0, (hipStream_t)stream, coords, out, h, w, k);
    DSNT_CHECK_LAUNCH("dsnt_make_gauss");
}

extern "C" int dsnt_make_gauss_bwd(const float* coords, const float* g_out, float* g_coords, int64_t rows, int h, int w,
                                   float sigma, void* stream) {
    DSNT_REQUIRE(coords && g_out && g_coords, DSNT_ERR_ARG, "dsnt_make_gauss_bwd: null tensor");
    DSNT_REQUIRE(sigma > 0.f, DSNT_ERR_ARG, "dsnt_make_gauss_bwd: sigma must be positive");
    if (int e = check_rows("dsnt_make_gauss_bwd", rows, h, w)) return e;
    const float k = (float)(-0.5 * (1.0 / (double)sigma) * (1.0 / (double)sigma));
    DSNT_LAUNCH(make_gauss_bwd_kernel, dim3((int)rows), dim3(HB), 0, (hipStream_t)stream, coords, g_out,
                       g_coords, h, w, k);
    DSNT_CHECK_LAUNCH("dsnt_make_gauss_bwd");
}

static inline float gauss_k(float sigma) { return (float)(-0.5 * (1.0 / (double)sigma) * (1.0 / (double)sigma)); }

extern "C" int dsnt_reg_fwd(const float* hm, const float* target, float* per_row, int64_t rows, int h, int w,
                            float sigma, int kind, void* stream) {
    DSNT_REQUIRE(hm && per_row && (kind == 3 || target), DSNT_ERR_ARG, "dsnt_reg_fwd: null tensor");
    DSNT_REQUIRE(kind >= 0 && kind <= 3, DSNT_ERR_ARG, "dsnt_reg_fwd: unknown kind %d", kind);
    if (int e = check_rows("dsnt_reg_fwd", rows, h, w)) return e;
    ROW_DISPATCH(reg_fwd_kernel, (int)rows, h * w, dsnt_aligned16(hm), hm, target, per_row, h, w, sigma,
                 gauss_k(sigma), kind);
    DSNT_CHECK_LAUNCH("dsnt_reg_fwd");
}

extern "C" int dsnt_reg_bwd(const float* hm, const float* target, const float* g_row, float* ghm, int64_t rows,
                            int h, int w, float sigma, int kind, void* stream) {
    DSNT_REQUIRE(hm && g_row && ghm && (kind == 3 || target), DSNT_ERR_ARG, "dsnt_reg_bwd: null tensor");
    DSNT_REQUIRE(kind >= 0 && kind <= 3, DSNT_ERR_ARG, "dsnt_reg_bwd: unknown kind %d", kind);
    if (int e = check_rows("dsnt_reg_bwd", rows, h, w)) return e;
    ROW_DISPATCH(reg_bwd_kernel, (int)rows, h * w, dsnt_aligned16(hm), hm, target, g_row, ghm, h, w, sigma,
                 gauss_k(sigma), kind);
    DSNT_CHECK_LAUNCH("dsnt_reg_bwd");
}

extern "C" int dsnt_reg_bwd_mu(const float* hm, const float* target, const float* g_row, float* gmu, int64_t rows,
                               int h, int w, float sigma, int kind, void* stream) {
    DSNT_REQUIRE(hm && g_row && gmu && target, DSNT_ERR_ARG, "dsnt_reg_bwd_mu: null tensor");
    DSNT_REQUIRE(kind >= 0 && kind <= 2, DSNT_ERR_ARG, "dsnt_reg_bwd_mu: kind %d has no target Gaussian", kind);
    if (int e = check_rows("dsnt_reg_bwd_mu", rows, h, w)) return e;
    ROW_DISPATCH(reg_bwd_mu_kernel, (int)rows, h * w, dsnt_aligned16(hm), hm, target, g_row, gmu, h, w,
                 gauss_k(sigma), kind);
    DSNT_CHECK_LAUNCH("dsnt_reg_bwd_mu");
}

extern "C" int dsnt_euclid_fwd(const float* actual, const float* target, float* dist, int64_t n, int d,
                               void* stream) {
    DSNT_REQUIRE(actual && target && dist && n > 0 && d > 0, DSNT_ERR_ARG, "dsnt_euclid_fwd: bad argument");
    DSNT_LAUNCH(euclid_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       actual, target, dist, (long)n, d);
    DSNT_CHECK_LAUNCH("dsnt_euclid_fwd");
}

extern "C" int dsnt_euclid_bwd(const float* actual, const float* target, const float* dist, const float* g_dist,
                               float* g_actual, int64_t n, int d, void* stream) {
    DSNT_REQUIRE(actual && target && dist && g_dist && g_actual && n > 0 && d > 0, DSNT_ERR_ARG,
                 "dsnt_euclid_bwd: bad argument");
    DSNT_LAUNCH(euclid_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       actual, target, dist, g_dist, g_actual, (long)n, d);
    DSNT_CHECK_LAUNCH("dsnt_euclid_bwd");
}

extern "C" int dsnt_masked_avg_fwd(const float* losses, const float* mask, float* out2, int64_t n, void* stream) {
    DSNT_REQUIRE(losses && out2 && n > 0, DSNT_ERR_ARG, "dsnt_masked_avg_fwd: bad argument");
    DSNT_LAUNCH(masked_avg_fwd_kernel, dim3(1), dim3(HB), 0, (hipStream_t)stream, losses, mask, out2, (long)n);
    DSNT_CHECK_LAUNCH("dsnt_masked_avg_fwd");
}

extern "C" int dsnt_masked_avg_bwd(const float* g_out, const float* mask, const float* out2, float* g_losses,
                                   int64_t n, void* stream) {
    DSNT_REQUIRE(g_out && out2 && g_losses && n > 0, DSNT_ERR_ARG, "dsnt_masked_avg_bwd: bad argument");
    DSNT_LAUNCH(masked_avg_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       g_out, mask, out2, g_losses, (long)n);
    DSNT_CHECK_LAUNCH("dsnt_masked_avg_bwd");
}

extern "C" int dsnt_head_fwd(const float* logits, float* hm, float* coords, int64_t rows, int h, int w,
                             void* stream) {
    DSNT_REQUIRE(logits && hm && coords, DSNT_ERR_ARG, "dsnt_head_fwd: null tensor");
    if (int e = check_rows("dsnt_head_fwd", rows, h, w)) return e;
    ROW_DISPATCH(head_fwd_kernel, (int)rows, h * w, dsnt_aligned16(logits) && dsnt_aligned16(hm), logits, hm,
                 coords, h, w);
    DSNT_CHECK_LAUNCH("dsnt_head_fwd");
}

extern "C" int dsnt_head_loss_rows(const float* hm, const float* coords, const float* target, float* dist,
                                   float* reg_row, int64_t rows, int h, int w, float sigma, int reg_kind,
                                   void* stream) {
    DSNT_REQUIRE(hm && coords && target && dist && (reg_kind < 0 || reg_row), DSNT_ERR_ARG,
                 "dsnt_head_loss_rows: null tensor");
    DSNT_REQUIRE(reg_kind >= -1 && reg_kind <= 3, DSNT_ERR_ARG, "dsnt_head_loss_rows: unknown regulariser %d", reg_kind);
    if (int e = check_rows("dsnt_head_loss_rows", rows, h, w)) return e;
    ROW_DISPATCH(head_loss_rows_kernel, (int)rows, h * w, dsnt_aligned16(hm), hm, coords, target, dist, reg_row,
                 h, w, sigma, gauss_k(sigma), reg_kind);
    DSNT_CHECK_LAUNCH("dsnt_head_loss_rows");
}

extern "C" int dsnt_head_bwd(const float* hm, const float* coords, const float* target, const float* dist,
                             const float* g_dist, const float* g_reg, float* g_logits, int64_t rows, int h, int w,
                             float sigma, int reg_kind, void* stream) {
    DSNT_REQUIRE(hm && coords && target && dist && g_dist && g_logits, DSNT_ERR_ARG, "dsnt_head_bwd: null tensor");
    DSNT_REQUIRE(reg_kind >= -1 && reg_kind <= 3, DSNT_ERR_ARG, "dsnt_head_bwd: unknown regulariser %d", reg_kind);
    if (int e = check_rows("dsnt_head_bwd", rows, h, w)) return e;
    ROW_DISPATCH(head_bwd_kernel, (int)rows, h * w, dsnt_aligned16(hm), hm, coords, target, dist, g_dist, g_reg,
                 g_logits, h, w, sigma, gauss_k(sigma), reg_kind);
    DSNT_CHECK_LAUNCH("dsnt_head_bwd");
}

extern "C" int dsnt_mask_denom(const float* mask, float* denom2, int64_t n, void* stream) {
    DSNT_REQUIRE(denom2 && n > 0, DSNT_ERR_ARG, "dsnt_mask_denom: bad argument");
    DSNT_LAUNCH(mask_denom_kernel, dim3(1), dim3(HB), 0, (hipStream_t)stream, mask, denom2, (long)n);
    DSNT_CHECK_LAUNCH("dsnt_mask_denom");
}

extern "C" int dsnt_head_loss_grad(const float* hm, const float* coords, const float* target, const float* mask,
                                   const float* denom2, float* dist, float* reg_row, float* g_logits, int64_t rows, int h, int w,
                                   float sigma, int reg_kind, float reg_coeff, void* stream) {
    DSNT_REQUIRE(hm && coords && target && denom2 && dist && g_logits && (reg_kind < 0 || reg_row), DSNT_ERR_ARG,
                 "dsnt_head_loss_grad: null tensor");
    DSNT_REQUIRE(reg_kind >= -1 && reg_kind <= 3, DSNT_ERR_ARG, "dsnt_head_loss_grad: unknown regulariser %d", reg_kind);
    if (int e = check_rows("dsnt_head_loss_grad", rows, h, w)) return e;
    DSNT_REQUIRE((long)h * w <= 4096, DSNT_ERR_SHAPE, "dsnt_head_loss_grad: heat-maps of up to 4096 pixels (got %dx%d); use "
                 "dsnt_head_loss_rows + dsnt_head_bwd for larger ones", h, w);
    hipStream_t st = (hipStream_t)stream;
    const bool vec = ((h * w) % 4 == 0) && dsnt_aligned16(hm) && dsnt_aligned16(g_logits);
    if (vec) DSNT_LAUNCH(head_loss_grad_kernel<4>, dim3((int)rows), dim3(HB), 0, st, hm, coords, target, mask, denom2, dist,
                                reg_row, g_logits, h, w, sigma, gauss_k(sigma), reg_kind, reg_coeff);
    else DSNT_LAUNCH(head_loss_grad_kernel<1>, dim3((int)rows), dim3(HB), 0, st, hm, coords, target, mask, denom2, dist,
                            reg_row, g_logits, h, w, sigma, gauss_k(sigma), reg_kind, reg_coeff);
    DSNT_CHECK_LAUNCH("dsnt_head_loss_grad");
}

extern "C" int dsnt_head_loss_reduce(const float* dist, const float* reg_row, const float* mask, const float* denom2,
                                     float reg_coeff, float* loss, float* e2, int64_t rows, void* stream) {
    DSNT_REQUIRE(dist && denom2 && loss && e2 && rows > 0, DSNT_ERR_ARG, "dsnt_head_loss_reduce: bad argument");
    DSNT_LAUNCH(head_loss_reduce_kernel, dim3(1), dim3(HB), 0, (hipStream_t)stream, dist, reg_row, mask, denom2, reg_coeff,
                       loss, e2, (long)rows);
    DSNT_CHECK_LAUNCH("dsnt_head_loss_reduce");
}

extern "C" int dsnt_scale_by_scalar(float* x, const float* s, int64_t n, void* stream) {
    DSNT_REQUIRE(x && s && n > 0 && n % 4 == 0 && dsnt_aligned16(x), DSNT_ERR_ARG,
                 "dsnt_scale_by_scalar: n must be a positive multiple of 4, x 16-byte aligned");
    long gsz = (n / 4 + 255) / 256;
    if (gsz > 4096) gsz = 4096;
    DSNT_LAUNCH(scale_by_scalar_kernel, dim3((unsigned)gsz), dim3(256), 0, (hipStream_t)stream, (float4*)x, s, (long)(n / 4));
    DSNT_CHECK_LAUNCH("dsnt_scale_by_scalar");
}

// ---------------------------------------------------------------- 'fc' output strategy
// out[row][k] = sum_i hm[row][i] * W[k][i] + b[k], k = 0,1 — `out_fc = nn.Linear(H*W, 2)` applied to the flattened
// heat-maps (reference model.py:222-223, 293-303; :196-198 for ResNet).  One workgroup per row.
__global__ __launch_bounds__(256) void fc2_fwd_kernel(const float* __restrict__ hm, const float* __restrict__ w,
                                                      const float* __restrict__ b, float* __restrict__ out, int hw) {
    __shared__ float red[2][4];
    const float* row = hm + (size_t)blockIdx.x * hw;
    float a0 = 0.f, a1 = 0.f;
    for (int i = threadIdx.x; i < hw; i += 256) {
        const float v = row[i];
        a0 = fmaf(v, w[i], a0);
        a1 = fmaf(v, w[hw + i], a1);
    }
    for (int o = 32; o >= 1; o >>= 1) { a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a0; red[1][threadIdx.x >> 6] = a1; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const float s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        out[(size_t)blockIdx.x * 2 + threadIdx.x] = s + (b ? b[threadIdx.x] : 0.f);
    }
}

extern "C" int dsnt_fc2_fwd(const float* hm, const float* w, const float* b, float* out, int64_t rows, int hw,
                            void* stream) {
    DSNT_REQUIRE(hm && w && out && rows > 0 && rows < (1LL << 31) && hw > 0, DSNT_ERR_ARG, "dsnt_fc2_fwd: bad argument");
    DSNT_LAUNCH(fc2_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, hm, w, b, out, hw);
    DSNT_CHECK_LAUNCH("dsnt_fc2_fwd");
}

// ghm[row][i] = g[row][0] W[0][i] + g[row][1] W[1][i];  gW[k][i] = sum_row g[row][k] hm[row][i];  gb[k] = sum_row g[row][k].
// One thread per column i walks the rows in order (deterministic); rows = B * 16 is small.
__global__ __launch_bounds__(256) void fc2_bwd_kernel(const float* __restrict__ g, const float* __restrict__ hm,
                                                      const float* __restrict__ w, float* __restrict__ ghm,
                                                      float* __restrict__ gw, float* __restrict__ gb, int rows, int hw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < hw) {
        const float w0 = w[i], w1 = w[hw + i];
        float s0 = 0.f, s1 = 0.f;
        for (int r = 0; r < rows; ++r) {
            const float g0 = g[2 * r], g1 = g[2 * r + 1];
            const float v = hm[(size_t)r * hw + i];
            if (ghm) ghm[(size_t)r * hw + i] = fmaf(g0, w0, g1 * w1);
            s0 = fmaf(g0, v, s0);
            s1 = fmaf(g1, v, s1);
        }
        gw[i] = s0;
        gw[hw + i] = s1;
    }
    if (gb && blockIdx.x == 0 && threadIdx.x < 2) {
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s += g[2 * r + threadIdx.x];
        gb[threadIdx.x] = s;
    }
}

extern "C" int dsnt_fc2_bwd(const float* g, const float* hm, const float* w, float* ghm, float* gw, float* gb,
                            int64_t rows, int hw, void* stream) {
    DSNT_REQUIRE(g && hm && w && gw && rows > 0 && rows < (1LL << 31) && hw > 0, DSNT_ERR_ARG, "dsnt_fc2_bwd: bad argument");
    DSNT_LAUNCH(fc2_bwd_kernel, dim3((hw + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, hm, w, ghm, gw, gb,
                       (int)rows, hw);
    DSNT_CHECK_LAUNCH("dsnt_fc2_bwd");
}
