// Forward of the 1x1 convolutions of the full-resolution levels (hourglass.py:20,25: conv1 / conv3 of every Bottleneck;
// :44-48 the projection shortcuts; :146-153 the `fc` convolutions) on the fp16 matrix cores, fp16x3 split, in the shape of
// the one-pass backward (bwd1.hip) — the second streaming 1x1 kernel, after gemm1.hip:
//
//   y[m][n] = sum_k act(x)[m][k] W[n][k] + bias[n] (+ res[m][n]),   act = relu?(x scale + shift) or x itself,
//   + the per-column sums (sum y, sum y^2) of the next BatchNorm's batch statistics, ONE row per workgroup.
//
// gemm1 loads a lane's MFMA operand straight from memory: a lane is a ROW, so one load instruction of a wave touches 32
// rows x 32 bytes — 32 lines per instruction, and the address path, not the bytes, is what it queues on (3.0-4.2 TB/s in the
// step).  Here the activations are loaded the way bwd1 loads dY — whole rows by all threads, 16 bytes per lane, 8 lines per
// instruction, one 32-pixel stage ahead in registers —, transformed and split once, and written pixel-major into a
// double-buffered LDS image that every wave reads as its A operand (ds_read_b128); a wave OWNS 16 CW output columns for all
// pixels of the workgroup, with its slice of W resident (hi plane in registers, lo plane in LDS).  The epilogue works from
// the accumulator layout (column on the lane, 4 pixel rows in registers: 64-byte runs), residuals prefetched a stage ahead.
// v_mfma_f32_16x16x32_f16; one barrier per 32 pixels; 48 MFMAs per wave and stage.
#include "fwd1.h"
#include <stdlib.h>

typedef unsigned f1_u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Fwd1P {
    const float* x; const float* in_scale; const float* in_shift; int in_relu;
    const unsigned short* wq; long wq_stride;           // [Cout][KK] fp16 planes (the forward layout: OHWI with R = S = 1)
    const float* a_bound; const float* w_bound;
    const float* bias; const float* res; float* y; float* stats;
    OutBoundsP ob;
    int M, Cout, nstages, spw;
};

// KK = input channels (the contraction), CW = 16-column tiles per wave, NWV = waves per workgroup (16 CW NWV output columns per
// workgroup; blockIdx.y = column chunk), PRO: BatchNorm(+ReLU) on the operand, RES: one residual addend
template <int KK, int CW, int NWV, bool PRO, bool RES>
__global__ __launch_bounds__(64 * NWV, 2) void fwd1_kernel(Fwd1P p) {
    constexpr int CC = 16 * CW * NWV;
    constexpr int NTHR = 64 * NWV;
    constexpr int KS = KK / 32;
    constexpr int PITCH = KK * 2 + 32;          // bytes per pixel row of one plane (and per column row of the W image)
    constexpr int PL = 32 * PITCH;
    constexpr int IMG = 2 * PL;
    constexpr int WLO = CC * PITCH;
    constexpr int K4 = KK / 4;
    constexpr int U = 32 * K4 / NTHR;
    constexpr int RSTEP = NTHR / K4;
    static_assert(U >= 1 && U * NTHR == 32 * K4, "staging units per thread");
    const unsigned OOB = 0xF0000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char f1_smem[];
    unsigned char* img = f1_smem;                                   // [2 buffers][2 planes][32 pixels][PITCH]
    unsigned char* wlo = f1_smem + 2 * IMG;                         // [CC][PITCH]: lo plane of W
    float* vec = reinterpret_cast<float*>(wlo + WLO);               // PRO: [2][KK] scale, shift (x operand scale)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, lg = lane >> 4;
    const int wg = blockIdx.x;
    const int col0 = blockIdx.y * CC;
    const int ld = p.Cout;
    const int s0 = wg * p.spw;
    const int s1 = min(p.nstages, s0 + p.spw);
    const float sa = pow2_scale(bound64(p.a_bound)), sw = pow2_scale(bound64(p.w_bound));
    const float osc = 1.f / (sa * sw);

    auto stage_off = [&](const int s, const unsigned stage_bytes) {      // (scalar select, laundered: bwd1.hip)
        const unsigned o = s < s1 ? (unsigned)s * stage_bytes : OOB;
        return (unsigned)__builtin_amdgcn_readfirstlane((int)o);
    };
    // ---- activation staging: unit i of this thread = pixel row (tid / K4) + RSTEP i, channels 4 k4 .. 4 k4 + 3
    const int k4 = tid % K4, prow = tid / K4;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.M * KK * 4u), 0x00020000);
    const unsigned xlane = (unsigned)((prow * KK + 4 * k4) * 4);
    f1_u32x4 R[U];
    auto issue = [&](const int s) {
        const unsigned so = stage_off(s, 32u * KK * 4u) + xlane;
#pragma unroll
        for (int i = 0; i < U; ++i) R[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, so + (unsigned)(RSTEP * i * KK * 4), 0, 0);
    };
    const float relu_lo = (PRO && p.in_relu) ? 0.f : -__builtin_inff();
    auto transform = [&](const int buf) {
        float4 sc = make_float4(sa, sa, sa, sa), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (PRO) {
            sc = *reinterpret_cast<const float4*>(vec + 4 * k4);
            sh = *reinterpret_cast<const float4*>(vec + KK + 4 * k4);
        }
        unsigned char* base = img + buf * IMG + prow * PITCH + 8 * k4;
#pragma unroll
        for (int i = 0; i < U; ++i) {
            float4 v = make_float4(__uint_as_float(R[i].x), __uint_as_float(R[i].y), __uint_as_float(R[i].z), __uint_as_float(R[i].w));
            if (PRO) {
                v.x = fmaxf(fmaf(v.x, sc.x, sh.x), relu_lo); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), relu_lo);
                v.z = fmaxf(fmaf(v.z, sc.z, sh.z), relu_lo); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), relu_lo);
            } else {
                v.x *= sa; v.y *= sa; v.z *= sa; v.w *= sa;
            }
            uint2 q1, q2;
            split4h(v, q1, q2);
            *reinterpret_cast<uint2*>(base + RSTEP * i * PITCH) = q1;
            *reinterpret_cast<uint2*>(base + RSTEP * i * PITCH + PL) = q2;
        }
    };
    issue(s0);

    // ---- matrix role: this wave's output columns
    const int c0 = wave * 16 * CW;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.M * ld * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(RES ? p.res : p.y), 0, (int)((size_t)p.M * ld * 4u), 0x00020000);
    const unsigned ylane = (unsigned)((4 * lg * ld + col0 + c0 + lc) * 4);
    const unsigned stage_y = 32u * (unsigned)ld * 4u;
    float rv[RES ? 2 : 1][CW][4];
    auto issue_r = [&](const int rt, const int s) {
        if (!RES) return;
        const unsigned sbase = stage_off(s, stage_y) + (unsigned)(16 * rt * ld * 4) + ylane;
#pragma unroll
        for (int cw = 0; cw < CW; ++cw)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                rv[rt][cw][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, sbase + (unsigned)(cw * 64), (unsigned)(r * ld * 4), 0));
    };
    issue_r(0, s0);
    issue_r(1, s0);

    f16x8 whi[KS][CW];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int cw = 0; cw < CW; ++cw)
            whi[ks][cw] = *reinterpret_cast<const f16x8*>(p.wq + (size_t)(col0 + c0 + 16 * cw + lc) * KK + 32 * ks + 8 * lg);
    {
        constexpr int UN = CC * KK / 8;
        static_assert(UN % NTHR == 0, "weight units per thread");
#pragma unroll
        for (int j = 0; j < UN / NTHR; ++j) {
            const int u = tid + NTHR * j;
            const int c = u / (KK / 8), k8 = u % (KK / 8);
            *reinterpret_cast<uint4*>(wlo + c * PITCH + 16 * k8) =
                *reinterpret_cast<const uint4*>(p.wq + (size_t)p.wq_stride + (size_t)(col0 + c) * KK + 8 * k8);
        }
    }
    if (PRO) {
        for (int k = tid; k < KK; k += NTHR) { vec[k] = p.in_scale[k] * sa; vec[KK + k] = p.in_shift[k] * sa; }
    }
    float cb[CW], asc[CW], ash[CW];
#pragma unroll
    for (int cw = 0; cw < CW; ++cw) {
        const int c = col0 + c0 + 16 * cw + lc;
        cb[cw] = p.bias ? p.bias[c] : 0.f;
        asc[cw] = p.ob.amax_bn ? p.ob.amax_scale[c] : 0.f;
        ash[cw] = p.ob.amax_bn ? p.ob.amax_shift[c] : 0.f;
    }
    const float am2lo = p.ob.amax_relu ? 0.f : -__builtin_inff();
    __syncthreads();
    transform(0);
    issue(s0 + 1);
    __syncthreads();

    float s1a[CW], s2a[CW];
#pragma unroll
    for (int cw = 0; cw < CW; ++cw) { s1a[cw] = 0.f; s2a[cw] = 0.f; }
    float am = 0.f, am2 = 0.f;
    const unsigned a_lane = (unsigned)(lc * PITCH + 16 * lg);
    const unsigned w_lane = (unsigned)(2 * IMG + (c0 + lc) * PITCH + 16 * lg);

    auto stage = [&](const int s) {             // (first stage peeled: exact vmcnt counts across the back edge — bwd1.hip)
        const int buf = (s - s0) & 1;
        transform(buf ^ 1);
        issue(s + 2);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned ib = (unsigned)(buf * IMG);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x4 acc[CW];
#pragma unroll
            for (int cw = 0; cw < CW; ++cw) acc[cw] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const unsigned ao = ib + (unsigned)(rt * 16 * PITCH + 64 * ks) + a_lane;
                const f16x8 d1 = *reinterpret_cast<const f16x8*>(f1_smem + ao);
                const f16x8 d2 = *reinterpret_cast<const f16x8*>(f1_smem + ao + PL);
#pragma unroll
                for (int cw = 0; cw < CW; ++cw) {
                    const f16x8 b2 = *reinterpret_cast<const f16x8*>(f1_smem + w_lane + cw * 16 * PITCH + 64 * ks);
                    acc[cw] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d2, whi[ks][cw], acc[cw], 0, 0, 0);
                    acc[cw] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d1, b2, acc[cw], 0, 0, 0);
                    acc[cw] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d1, whi[ks][cw], acc[cw], 0, 0, 0);
                }
            }
            const unsigned obase = (unsigned)(s * 32 + 16 * rt) * (unsigned)ld * 4u + ylane;
#pragma unroll
            for (int cw = 0; cw < CW; ++cw) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[cw][r] * osc + cb[cw];
                    if (RES) v += rv[rt][cw][r];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, obase + (unsigned)(cw * 64), (unsigned)(r * ld * 4), 0);
                    am = fmaxf(am, fabsf(v));
                    am2 = fmaxf(am2, fabsf(fmaxf(fmaf(v, asc[cw], ash[cw]), am2lo)));      // (unconditional: a uniform branch per element otherwise)
                    s1a[cw] += v;
                    s2a[cw] = fmaf(v, v, s2a[cw]);
                }
            }
            issue_r(rt, s + 1);
        }
        __syncthreads();
    };
    stage(s0);
    for (int s = s0 + 1; s < s1; ++s) stage(s);

    if (p.stats) {
#pragma unroll
        for (int cw = 0; cw < CW; ++cw) {
            float a = s1a[cw], b = s2a[cw];
            a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
            a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
            if (lg == 0) {
                p.stats[((size_t)wg * 2 + 0) * ld + col0 + c0 + 16 * cw + lc] = a;
                p.stats[((size_t)wg * 2 + 1) * ld + col0 + c0 + 16 * cw + lc] = b;
            }
        }
    }
    if (p.ob.amax) amax_commit(am, p.ob.amax);
    if (p.ob.amax_bn) amax_commit(am2, p.ob.amax_bn, 1);
}

// ---------------------------------------------------------------- host side
static int f1_enabled = -1;
struct F1Cfg { int cin, cout, cw, nwv, chunks; };
static const F1Cfg f1_cfgs[] = {
    {256, 128, 1, 8, 1},        // conv1 of a Bottleneck
    {128, 256, 2, 8, 1},        // conv3, the 128 -> 256 projection shortcut
    {128, 128, 1, 8, 1},
    {64, 64, 1, 4, 1},          // the 128 x 128 level
    {64, 128, 1, 8, 1},
    {256, 256, 1, 8, 2},        // the `fc` convolutions: two column chunks
};

Fwd1Plan dsnt_fwd1_plan(const dsnt_conv_geom* g, bool share) {
    Fwd1Plan pl;
    memset(&pl, 0, sizeof(pl));
    if (f1_enabled < 0) f1_enabled = dsnt_kernel_off("fwd1") ? 0 : 1;
    if (!f1_enabled || !g) return pl;
    if (!(g->R == 1 && g->S == 1 && g->stride == 1 && g->pad == 0 && g->Ho == g->H && g->Wo == g->W)) return pl;
    const long M = (long)g->N * g->H * g->W;
    static long min_rows = -1;
    // (16384 when the kernel was written; with the gradient plumbing of the end of round 4 the 16 x 16 level gains too: hg2 batch 32
    // 11.25-11.29 vs 11.30-11.37 ms at 8192 rows, hg8 batch 16 -0.08 ms at 4096 together with DSNT_BF16X6_MIN_ROWS; 2048: +0.07)
    if (min_rows < 0) { const char* e = getenv("DSNT_X_FWD1_MIN_ROWS"); min_rows = e ? atol(e) : 4096; }      // A/B only
    if (M % 32 != 0 || M < min_rows) return pl;
    if ((size_t)M * g->Cin * 4u >= (1ull << 31) || (size_t)M * g->Cout * 4u >= (1ull << 31)) return pl;
    int cfg = -1;
    for (int i = 0; i < (int)(sizeof(f1_cfgs) / sizeof(f1_cfgs[0])); ++i)
        if (g->Cin == f1_cfgs[i].cin && g->Cout == f1_cfgs[i].cout) cfg = i;
    if (cfg < 0) return pl;
    const F1Cfg& c = f1_cfgs[cfg];
    const int cus = dsnt_device_cus();
    const int nstages = (int)(M / 32);
    int nwg = cus * (c.nwv == 4 ? 2 : 1) / c.chunks;
    if (share) nwg = nwg / 2 > 0 ? nwg / 2 : 1;             // DSNT_CONV_SHARE_CHIP: half of the CUs (gemm1.hip, bwd1.hip)
    if (c.chunks > 1) nwg = nwg / 8 * 8 > 0 ? nwg / 8 * 8 : nwg;
    if (nwg > nstages) nwg = nstages;
    const int spw = (nstages + nwg - 1) / nwg;
    nwg = (nstages + spw - 1) / spw;
    const int pitch = c.cin * 2 + 32;
    pl.ok = 1; pl.cfg = cfg; pl.nstages = nstages; pl.spw = spw; pl.nwg = nwg; pl.chunks = c.chunks;
    pl.lds = 2 * 2 * 32 * pitch + 16 * c.cw * c.nwv * pitch + 2 * c.cin * 4;
    return pl;
}

template <int KK, int CW, int NWV, bool PRO, bool RES>
static void f1_launch_k(const Fwd1Plan& pl, const Fwd1P& p, hipStream_t st) {
    DSNT_SET_MAX_LDS((fwd1_kernel<KK, CW, NWV, PRO, RES>), pl.lds);
    DSNT_LAUNCH((fwd1_kernel<KK, CW, NWV, PRO, RES>), dim3(pl.nwg, pl.chunks), dim3(64 * NWV), pl.lds, st, p);
}
template <int KK, int CW, int NWV>
static void f1_launch_cfg(const Fwd1Plan& pl, const Fwd1P& p, hipStream_t st) {
    const bool pro = p.in_scale != nullptr, res = p.res != nullptr;
    if (pro) { if (res) f1_launch_k<KK, CW, NWV, true, true>(pl, p, st); else f1_launch_k<KK, CW, NWV, true, false>(pl, p, st); }
    else { if (res) f1_launch_k<KK, CW, NWV, false, true>(pl, p, st); else f1_launch_k<KK, CW, NWV, false, false>(pl, p, st); }
}

extern "C" int dsnt_conv1x1_fwd_ok(const dsnt_conv_geom* g) { return dsnt_fwd1_plan(g, false).ok; }
extern "C" int dsnt_conv1x1_fwd_stats_rows(const dsnt_conv_geom* g, int in_relu_flags) {
    return dsnt_fwd1_plan(g, (in_relu_flags & DSNT_CONV_SHARE_CHIP) != 0).nwg;
}

extern "C" int dsnt_conv1x1_fwd_f16x3(const float* x, const void* w_planes, int64_t plane_stride, const float* w_bound,
                                      const float* a_bound, const float* bias, float* y, const float* in_scale,
                                      const float* in_shift, int in_relu, const float* res1, float* stats_partial,
                                      const dsnt_conv_geom* g, const dsnt_out_bounds* tail, void* stream) {
    DSNT_REQUIRE(x && w_planes && w_bound && a_bound && y && g, DSNT_ERR_ARG, "dsnt_conv1x1_fwd_f16x3: bad argument");
    DSNT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), DSNT_ERR_ARG,
                 "dsnt_conv1x1_fwd_f16x3: in_scale/in_shift must be given together");
    const Fwd1Plan pl = dsnt_fwd1_plan(g, (in_relu & DSNT_CONV_SHARE_CHIP) != 0);
    DSNT_REQUIRE(pl.ok, DSNT_ERR_SHAPE, "dsnt_conv1x1_fwd_f16x3: geometry not supported (dsnt_conv1x1_fwd_ok)");
    DSNT_REQUIRE(dsnt_aligned16(x) && dsnt_aligned16(w_planes) && dsnt_aligned16(y) && (!res1 || dsnt_aligned16(res1)) &&
                 (!in_scale || (dsnt_aligned16(in_scale) && dsnt_aligned16(in_shift))) && plane_stride % 8 == 0, DSNT_ERR_ALIGN,
                 "dsnt_conv1x1_fwd_f16x3: 16-byte alignment required");
    Fwd1P p;
    memset(&p, 0, sizeof(p));
    if (int e = out_bounds_fill(p.ob, tail, "dsnt_conv1x1_fwd_f16x3")) return e;
    p.x = x; p.in_scale = in_scale; p.in_shift = in_shift; p.in_relu = in_relu & 1;
    p.wq = (const unsigned short*)w_planes; p.wq_stride = plane_stride; p.a_bound = a_bound; p.w_bound = w_bound;
    p.bias = bias; p.res = res1; p.y = y; p.stats = stats_partial;
    p.M = g->N * g->H * g->W; p.Cout = g->Cout; p.nstages = pl.nstages; p.spw = pl.spw;
    hipStream_t st = (hipStream_t)stream;
    switch (pl.cfg) {
    case 0: case 5: f1_launch_cfg<256, 1, 8>(pl, p, st); break;
    case 1: f1_launch_cfg<128, 2, 8>(pl, p, st); break;
    case 2: f1_launch_cfg<128, 1, 8>(pl, p, st); break;
    case 3: f1_launch_cfg<64, 1, 4>(pl, p, st); break;
    default: f1_launch_cfg<64, 1, 8>(pl, p, st); break;
    }
    DSNT_CHECK_LAUNCH("dsnt_conv1x1_fwd_f16x3");
}
