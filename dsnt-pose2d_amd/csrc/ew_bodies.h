// Bodies of the small NHWC kernels that can also run inside a persistent stage (stage.h): device functions of a VIRTUAL workgroup
// index, so that the stand-alone launch (elementwise.hip) and the stage's loop (conv.hip) execute the same instructions.  Each is
// written for 256 live threads; in a 512-thread stage workgroup the upper half skips the work and keeps the barriers.
#pragma once
#include "common.h"
#include "bn_pro.h"

#define TILE_ROWS 128

struct TileOpP {
    const float* a; const float* b; float* y; unsigned char* idx; float* partial;
    int N, Ho, Wo, C, cgs;
    OutBoundsP tail;
    const float* bn_scale; const float* bn_shift; int bn_relu;
};
// (bx, by, gy: the workgroup's indices and the grid's second dimension — blockIdx / gridDim of the stand-alone launch, the virtual
// ones of a persistent stage, stage.h; red: 256 * 8 floats of LDS.  Threads >= 256 — a stage workgroup has 512 — fall out of the
// row mapping by themselves (`active`) and only keep the barriers.)
template <int OP>
__device__ __forceinline__ void tile_op_stats_body(const TileOpP& q, const int bx, const int by, const int gy, float* red) {
    const float* __restrict__ a = q.a; const float* __restrict__ b = q.b; float* __restrict__ y = q.y;
    unsigned char* __restrict__ idx = q.idx; float* partial = q.partial;
    const int N = q.N, Ho = q.Ho, Wo = q.Wo, C = q.C, cgs = q.cgs;
    const OutBoundsP tail = q.tail;
    const float* __restrict__ bn_scale = q.bn_scale; const float* __restrict__ bn_shift = q.bn_shift; const int bn_relu = q.bn_relu;
    const int tid = threadIdx.x;
    const int C4 = C >> 2;
    const int rpar = 256 / cgs;
    const int cg_l = tid % cgs, rl = tid / cgs;
    const bool active = rl < rpar;
    const long M = (long)N * Ho * Wo;
    const long row0 = (long)bx * TILE_ROWS;
    const long row1 = row0 + TILE_ROWS < M ? row0 + TILE_ROWS : M;
    float am = 0.f;                                  // max |written value| (tail.amax: fp16x3 bound of a raw consumer)
    float am2 = 0.f;                                 // max |relu?(written value * scale + shift)| (tail.amax_bn)
    const float am2lo = tail.amax_relu ? 0.f : -__builtin_inff();
    for (int cg0 = by * cgs; cg0 < C4; cg0 += cgs * gy) {
        const int cg = cg0 + cg_l;
        float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
        if (active && cg < C4) {
            // batches of UB rows: every load of the batch is issued before the first use (a workgroup of a small level is
            // pure latency), rows are then consumed in the same order as one by one — the sums stay bit-identical to
            // tile_reduce_kernel<0>.  (Built WITHOUT the SLP vectoriser: see build.py.)
#ifndef DSNT_TILE_UB
#define DSNT_TILE_UB 1        // measured: 4 is no faster once small tensors use the 16-lane mapping
#endif
            constexpr int UB = DSNT_TILE_UB;
            float4 bs = make_float4(0.f, 0.f, 0.f, 0.f), bh = bs;
            if (tail.amax_bn) {
                bs = reinterpret_cast<const float4*>(tail.amax_scale)[cg];
                bh = reinterpret_cast<const float4*>(tail.amax_shift)[cg];
            }
            for (long rb = row0 + rl; rb < row1; rb += (long)UB * rpar) {
                float4 in[UB][OP == 0 ? 4 : 2];
                float4 osc = make_float4(0.f, 0.f, 0.f, 0.f), osh = osc;
                if (OP == 2) {
                    osc = reinterpret_cast<const float4*>(bn_scale)[cg];
                    osh = reinterpret_cast<const float4*>(bn_shift)[cg];
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const long r = rb + (long)u * rpar < row1 ? rb + (long)u * rpar : row1 - 1;     // clamped: never used
                    const int ow = (int)(r % Wo);
                    const long t = r / Wo;
                    const int oh = (int)(t % Ho), n = (int)(t / Ho);
                    if (OP == 0) {
                        const int W = Wo * 2;
                        const float4* base = reinterpret_cast<const float4*>(a) + (((long)n * (Ho * 2) + 2 * oh) * W + 2 * ow) * C4 + cg;
                        in[u][0] = base[0]; in[u][1] = base[C4];
                        in[u][OP == 0 ? 2 : 0] = base[(long)W * C4]; in[u][OP == 0 ? 3 : 1] = base[(long)W * C4 + C4];
                    } else if (OP == 1) {
                        in[u][0] = reinterpret_cast<const float4*>(a)[r * C4 + cg];
                        in[u][1] = reinterpret_cast<const float4*>(b)[(((long)n * (Ho >> 1) + (oh >> 1)) * (Wo >> 1) + (ow >> 1)) * C4 + cg];
                    } else {
                        in[u][0] = reinterpret_cast<const float4*>(a)[r * C4 + cg];
                    }
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const long r = rb + (long)u * rpar;
                    if (r >= row1) break;
                    float4 v;
                    if (OP == 0) {
                        const float4 v1 = in[u][1], v2 = in[u][OP == 0 ? 2 : 0], v3 = in[u][OP == 0 ? 3 : 1];
                        v = in[u][0];
                        uchar4 k = make_uchar4(0, 0, 0, 0);
#define POOL_STEP(V, P)                                  \
                        if (V.x > v.x || V.x != V.x) { v.x = V.x; k.x = P; } \
                        if (V.y > v.y || V.y != V.y) { v.y = V.y; k.y = P; } \
                        if (V.z > v.z || V.z != V.z) { v.z = V.z; k.z = P; } \
                        if (V.w > v.w || V.w != V.w) { v.w = V.w; k.w = P; }
                        POOL_STEP(v1, 1) POOL_STEP(v2, 2) POOL_STEP(v3, 3)
#undef POOL_STEP
                        reinterpret_cast<uchar4*>(idx)[r * C4 + cg] = k;
                    } else if (OP == 1) {
                        const float4 uu = in[u][0], l = in[u][1];
                        v = make_float4(uu.x + l.x, uu.y + l.y, uu.z + l.z, uu.w + l.w);
                    } else {
                        const float4 xv = in[u][0];
                        v = make_float4(fmaf(xv.x, osc.x, osh.x), fmaf(xv.y, osc.y, osh.y), fmaf(xv.z, osc.z, osh.z),
                                        fmaf(xv.w, osc.w, osh.w));
                        if (bn_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    }
                    reinterpret_cast<float4*>(y)[r * C4 + cg] = v;
                    am = fmaxf(fmaxf(am, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
                    if (tail.amax_bn)
                        am2 = fmaxf(fmaxf(am2, fabsf(fmaxf(fmaf(v.x, bs.x, bh.x), am2lo))),
                                    fmaxf(fabsf(fmaxf(fmaf(v.y, bs.y, bh.y), am2lo)),
                                          fmaxf(fabsf(fmaxf(fmaf(v.z, bs.z, bh.z), am2lo)), fabsf(fmaxf(fmaf(v.w, bs.w, bh.w), am2lo)))));
                    s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                    s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y);
                    s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
                }
            }
        }
        if (!partial) continue;                       // eval mode: only the operand bound is wanted (uniform)
        __syncthreads();
        if (tid < 256) {                              // (a stage workgroup has 512 threads: the upper half only keeps the barriers)
            float* mine = red + tid * 8;
            mine[0] = s1.x; mine[1] = s1.y; mine[2] = s1.z; mine[3] = s1.w;
            mine[4] = s2.x; mine[5] = s2.y; mine[6] = s2.z; mine[7] = s2.w;
        }
        __syncthreads();
        if (tid < cgs && cg0 + tid < C4) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int j = 0; j < rpar; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += red[(j * cgs + tid) * 8 + e];
            float* p0 = partial + ((size_t)bx * 2 + 0) * C + (size_t)(cg0 + tid) * 4;
            float* p1 = partial + ((size_t)bx * 2 + 1) * C + (size_t)(cg0 + tid) * 4;
            *reinterpret_cast<float4*>(p0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4*>(p1) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
    }
    if (tail.amax) amax_commit(am, tail.amax, 0, bx);
    if (tail.amax_bn) amax_commit(am2, tail.amax_bn, 1, bx);
}

template <int OP>
__global__ __launch_bounds__(256) void tile_op_stats_kernel(TileOpP q) {
    __shared__ __attribute__((aligned(16))) float red[256 * 8];
    tile_op_stats_body<OP>(q, blockIdx.x, blockIdx.y, gridDim.y, red);
}

// FIXED: the grid stride is a multiple of C/4, so a thread stays on ONE channel group — its six per-channel vectors are
// loaded once instead of with every element (they were two thirds of the kernel's load instructions), and two elements are
// in flight per iteration.  Same arithmetic, element for element.
struct BnApplyVec { float4 sc, sh, mu, is, c0, c1; };
__device__ __forceinline__ float4 bn_apply_one(const float4 g, const float4 xv, const BnApplyVec& v, int relu) {
    float4 dz = g;
    if (relu) {
        if (fmaf(xv.x, v.sc.x, v.sh.x) <= 0.f) dz.x = 0.f;
        if (fmaf(xv.y, v.sc.y, v.sh.y) <= 0.f) dz.y = 0.f;
        if (fmaf(xv.z, v.sc.z, v.sh.z) <= 0.f) dz.z = 0.f;
        if (fmaf(xv.w, v.sc.w, v.sh.w) <= 0.f) dz.w = 0.f;
    }
    float4 o;
    o.x = v.sc.x * (dz.x - v.c0.x - (xv.x - v.mu.x) * v.is.x * v.c1.x);
    o.y = v.sc.y * (dz.y - v.c0.y - (xv.y - v.mu.y) * v.is.y * v.c1.y);
    o.z = v.sc.z * (dz.z - v.c0.z - (xv.z - v.mu.z) * v.is.z * v.c1.z);
    o.w = v.sc.w * (dz.w - v.c0.w - (xv.w - v.mu.w) * v.is.w * v.c1.w);
    return o;
}
struct BnApplyP {
    const float4* da; const float4* x; const float4* scale; const float4* shift; const float4* mean; const float4* invstd;
    const float4* coef; int relu; float4* dx; const float4* base; long n4; int C4; unsigned* amax; BnBwdProP pro;
};
// (vb / vgrid: workgroup index and count of the recorded launch — 256 threads each; pro_sh: 512 doubles of LDS; finalise: false when
// this workgroup has already run the prologue for this launch — stage.h)
template <bool FIXED>
__device__ __forceinline__ void bn_act_bwd_apply_body(const BnApplyP& q, const int vb, const int vgrid, double* pro_sh, const bool finalise) {
    const float4* __restrict__ da = q.da; const float4* __restrict__ x = q.x; const float4* __restrict__ scale = q.scale;
    const float4* __restrict__ shift = q.shift; const float4* __restrict__ mean = q.mean; const float4* __restrict__ invstd = q.invstd;
    const float4* coef = q.coef; const int relu = q.relu; float4* dx = q.dx; const float4* base = q.base;
    const long n4 = q.n4; const int C4 = q.C4; unsigned* __restrict__ amax = q.amax; const BnBwdProP pro = q.pro;
    // base: what the result is added to — null (dx = value), dx itself (accumulate in place) or ANOTHER tensor (dx = base + value:
    // the gradient it continues stays intact for a reader that comes later, dsnt_bn_act_bwd_apply_base)
    const bool accumulate = base != nullptr;
    if (pro.partial && finalise) {       // dsnt_bn_act_bwd_apply_pro: coef / dgamma / dbeta from the tile sums, here
        bn_pro_backward<256>(pro, pro_sh, vb == 0);
        __syncthreads();                 // this workgroup's stores to coef are visible to its loads below
    }
    float am = 0.f;
    const long stride = (long)vgrid * 256;
    long i = threadIdx.x < 256 ? (long)vb * 256 + threadIdx.x : n4;       // (threads >= 256 of a stage workgroup: nothing to do)
    auto vec = [&](int cg) {
        BnApplyVec v;
        v.sc = scale[cg]; v.sh = shift[cg]; v.mu = mean[cg]; v.is = invstd[cg]; v.c0 = coef[cg]; v.c1 = coef[C4 + cg];
        return v;
    };
    auto amx = [&](const float4 o) { am = fmaxf(fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))), am); };
    if (FIXED) {
        const BnApplyVec v = vec((int)((i < n4 ? i : 0) % C4));
        for (; i + stride < n4; i += 2 * stride) {
            const float4 g0 = da[i], x0 = x[i], g1 = da[i + stride], x1 = x[i + stride];
            float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0;
            if (accumulate) { p0 = base[i]; p1 = base[i + stride]; }
            float4 o0 = bn_apply_one(g0, x0, v, relu), o1 = bn_apply_one(g1, x1, v, relu);
            if (accumulate) {
                o0.x += p0.x; o0.y += p0.y; o0.z += p0.z; o0.w += p0.w;
                o1.x += p1.x; o1.y += p1.y; o1.z += p1.z; o1.w += p1.w;
            }
            dx[i] = o0; dx[i + stride] = o1;
            amx(o0); amx(o1);
        }
        if (i < n4) {
            float4 o = bn_apply_one(da[i], x[i], v, relu);
            if (accumulate) { const float4 p = base[i]; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
            dx[i] = o;
            amx(o);
        }
    } else {
        for (; i < n4; i += stride) {
            float4 o = bn_apply_one(da[i], x[i], vec((int)(i % C4)), relu);
            if (accumulate) { const float4 p = base[i]; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
            dx[i] = o;
            amx(o);
        }
    }
    if (amax) amax_commit(am, amax, 0, vb);      // max |dx| for the fp16x3 consumers
}

template <bool FIXED>
__global__ void bn_act_bwd_apply_kernel(BnApplyP q) {
    __shared__ double pro_sh[512];
    bn_act_bwd_apply_body<FIXED>(q, blockIdx.x, gridDim.x, pro_sh, true);
}

// (`extra`, optional: a second gradient of x — same layout as dx — added in the same pass: dx (+)= extra + the routed dy)
struct PoolBwdP { const float4* dy; const uchar4* idx; float4* dx; int accumulate; const float4* extra; int N, H, W, C4; unsigned* amax; };
__device__ __forceinline__ void maxpool2_bwd_body(const PoolBwdP& q, const int vb, const int vgrid) {
    const float4* __restrict__ dy = q.dy; const uchar4* __restrict__ idx = q.idx; float4* dx = q.dx; const int accumulate = q.accumulate;
    const float4* __restrict__ extra = q.extra; const int H = q.H, W = q.W, C4 = q.C4, N = q.N; unsigned* amax = q.amax;
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * C4;
    float am = 0.f;
    for (long i = threadIdx.x < 256 ? (long)vb * 256 + threadIdx.x : total; i < total; i += (long)vgrid * 256) {
        const int cg = (int)(i % C4);
        long t = i / C4;
        const int ow = (int)(t % Wo); t /= Wo;
        const int oh = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const float4 g = dy[i];
        const uchar4 k = idx[i];
        float4* base = dx + (((long)n * H + 2 * oh) * W + 2 * ow) * C4 + cg;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float4* q = base + (p >> 1) * (long)W * C4 + (p & 1) * C4;
            float4 o = make_float4(k.x == p ? g.x : 0.f, k.y == p ? g.y : 0.f, k.z == p ? g.z : 0.f,
                                   k.w == p ? g.w : 0.f);
            if (accumulate) { const float4 c = *q; o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
            if (extra) { const float4 e = extra[q - dx]; o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w; }
            *q = o;
            am = fmaxf(fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))), am);
        }
    }
    if (amax) amax_commit(am, amax, 0, vb);
}

struct UpBwdP { const float4* dout; float4* dlow; int accumulate; int N, H, W, C4; unsigned* amax; };
__device__ __forceinline__ void upsample2_bwd_body(const UpBwdP& q, const int vb, const int vgrid) {
    const float4* __restrict__ dout = q.dout; float4* dlow = q.dlow; const int accumulate = q.accumulate;
    const int N = q.N, H = q.H, W = q.W, C4 = q.C4; unsigned* amax = q.amax;
    const int Hl = H >> 1, Wl = W >> 1;
    const long total = (long)N * Hl * Wl * C4;
    float am = 0.f;
    for (long i = threadIdx.x < 256 ? (long)vb * 256 + threadIdx.x : total; i < total; i += (long)vgrid * 256) {
        const int cg = (int)(i % C4);
        long t = i / C4;
        const int w = (int)(t % Wl); t /= Wl;
        const int h = (int)(t % Hl);
        const int n = (int)(t / Hl);
        const float4* b = dout + (((long)n * H + 2 * h) * W + 2 * w) * C4 + cg;
        const float4 v0 = b[0], v1 = b[C4], v2 = b[(long)W * C4], v3 = b[(long)W * C4 + C4];
        float4 o = make_float4((v0.x + v1.x) + (v2.x + v3.x), (v0.y + v1.y) + (v2.y + v3.y),
                               (v0.z + v1.z) + (v2.z + v3.z), (v0.w + v1.w) + (v2.w + v3.w));
        if (accumulate) { const float4 c = dlow[i]; o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
        dlow[i] = o;
        am = fmaxf(fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))), am);
    }
    if (amax) amax_commit(am, amax, 0, vb);
}

// Combine tile partials: 16 channels x 64 tile-lanes per 1024-thread block (the kernel is pure
// latency: many independent loads in flight matter, not bandwidth), fp64 accumulation.
// MODE 0: forward statistics.  MODE 1: backward sums.
#define FIN_T 1024
#define FIN_P (FIN_T / 16)
// MODE 1, optional: the bound of dx = scale (dz - coef0 - xhat coef1) for a consumer that forms dx in registers
// (dsnt_conv1x1_bwd_f16x3): max_c |scale_c| (max|dz| + |coef0_c| + |coef1_c| sqrt(M)), |xhat| <= sqrt(M) for the batch
// statistics of M samples; raised into the 64 slots of `out` like every other bound
struct BnBoundP { const float* scale; const float* dz_amax; float sqrtM; unsigned* out; };
struct BnFinP {
    const float* partial; int ntiles; double invM, unbias; int C;
    const float* gamma; const float* beta; float* running_mean; float* running_var;
    float momentum, eps; int training; float* o0; float* o1; float* o2; float* o3; int accumulate; BnBoundP bp;
};
// The body works on FIN_T VIRTUAL threads: a workgroup of NT threads (1024: the stand-alone launch; 512: a persistent stage,
// stage.h) carries FIN_T / NT of them per thread — virtual thread tid + v NT, i.e. virtual wave (tid >> 6) + v NT / 64 —, and
// every partial sum is formed and combined in the order of the 1024-thread kernel: the results are bit-identical.
// vb: workgroup index (16 channels each).  r0, r1: 256 doubles of LDS each.
template <int MODE, int NT>
__device__ __forceinline__ void bn_finalize_body(const BnFinP& q, const int vb, double* r0, double* r1) {
    constexpr int VP = FIN_T / NT;
    const float* __restrict__ partial = q.partial; const int ntiles = q.ntiles; const double invM = q.invM, unbias = q.unbias;
    const int C = q.C; const float* __restrict__ gamma = q.gamma; const float* __restrict__ beta = q.beta;
    float* running_mean = q.running_mean; float* running_var = q.running_var; const float momentum = q.momentum, eps = q.eps;
    const int training = q.training; float* o0 = q.o0; float* o1 = q.o1; float* o2 = q.o2; float* o3 = q.o3;
    const int accumulate = q.accumulate; const BnBoundP bp = q.bp;
    const int tid = threadIdx.x, cl = tid & 15;
    const int c = vb * 16 + cl;
    // everything the last step needs besides the sums is fetched NOW, under the row loop: the kernel's length is what a BatchNorm
    // costs the dependency chain, and a load issued behind the reduction is a microsecond of it (round 5, box N)
    const bool lead = tid < 16 && c < C;                    // (virtual part 0)
    float pg = 1.f, pb = 0.f, prm = 0.f, prv = 0.f, po0 = 0.f, po1 = 0.f, psc = 0.f, pdz = 0.f;
    if (MODE == 0 && lead) {
        if (gamma) pg = gamma[c];
        if (beta) pb = beta[c];
        if (running_mean) { prm = running_mean[c]; prv = running_var[c]; }
    }
    if (MODE == 1) {
        if (lead && accumulate) { if (o0) po0 = o0[c]; if (o1) po1 = o1[c]; }
        if (bp.out) { pdz = bp.dz_amax[tid & 63]; if (lead) psc = bp.scale[c]; }
    }
    double a0[VP], a1[VP];
#pragma unroll
    for (int v = 0; v < VP; ++v) {
        a0[v] = 0.0; a1[v] = 0.0;
        const int part = (tid + v * NT) >> 4;
        if (c < C && (MODE == 1 || training)) {
#pragma unroll 4
            for (int t = part; t < ntiles; t += FIN_P) {
                a0[v] += (double)partial[((size_t)t * 2 + 0) * C + c];
                a1[v] += (double)partial[((size_t)t * 2 + 1) * C + c];
            }
        }
    }
    // the four tile-lanes of a wave by shuffles, the waves through LDS, summed by the block's first 16 threads in a fixed order
    // (deterministic; ONE barrier — a seven-level tree over 1024 threads cost the chain ~0.5 us per BatchNorm, round 5)
#pragma unroll
    for (int v = 0; v < VP; ++v) {
        a0[v] += __shfl_xor(a0[v], 16, 64); a1[v] += __shfl_xor(a1[v], 16, 64);
        a0[v] += __shfl_xor(a0[v], 32, 64); a1[v] += __shfl_xor(a1[v], 32, 64);
        if ((tid & 63) < 16) { r0[(((tid + v * NT) >> 6)) * 16 + cl] = a0[v]; r1[(((tid + v * NT) >> 6)) * 16 + cl] = a1[v]; }
    }
    __syncthreads();
    double s0 = 0.0, s1 = 0.0;
    if (tid < 16) {
#pragma unroll
        for (int w = 0; w < FIN_T / 64; ++w) { s0 += r0[w * 16 + cl]; s1 += r1[w * 16 + cl]; }
    }
    if (tid < 16 && c < C) {
        if (MODE == 0) {
            double mean, var;
            if (training) {
                mean = s0 * invM;
                var = s1 * invM - mean * mean;
                if (var < 0.0) var = 0.0;
                if (running_mean) {
                    running_mean[c] = (float)((1.0 - momentum) * prm + momentum * mean);
                    running_var[c] = (float)((1.0 - momentum) * prv + momentum * var * unbias);
                }
            } else {
                mean = prm;
                var = prv;
            }
            const float is = (float)(1.0 / sqrt(var + (double)eps));
            const float mu = (float)mean;
            const float sc = gamma ? pg * is : is;
            o0[c] = mu; o1[c] = is; o2[c] = sc;
            o3[c] = (beta ? pb : 0.f) - mu * sc;
        } else {
            // o0 = dgamma, o1 = dbeta, o2 = coef [2][C]
            const float sdz = (float)s0, sdzx = (float)s1;
            if (o0) o0[c] = accumulate ? po0 + sdzx : sdzx;
            if (o1) o1[c] = accumulate ? po1 + sdz : sdz;
            o2[c] = (float)(s0 * invM);
            o2[C + c] = (float)(s1 * invM);
        }
    }
    if (MODE == 1 && bp.out) {
        float dzmax = pdz;                                       // (every thread: the shuffles need whole waves)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dzmax = fmaxf(dzmax, __shfl_xor(dzmax, o, 64));
        float b = 0.f;
        if (tid < 16 && c < C)
            b = fabsf(psc) * (dzmax + fabsf((float)(s0 * invM)) + fabsf((float)(s1 * invM)) * bp.sqrtM);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o, 64));
        if (tid == 0 && b > 0.f) atomicMax(bp.out + (vb & 63), __float_as_uint(b));
    }
}
template <int MODE>
__global__ __launch_bounds__(FIN_T) void bn_finalize_kernel(BnFinP q) {
    __shared__ double r0[256], r1[256];
    bn_finalize_body<MODE, FIN_T>(q, blockIdx.x, r0, r1);
}

