// Implicit-GEMM convolutions on the fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: exact f32, bit-for-bit a k-ordered fmaf chain).
//
// Forward / data-gradient:   Y[M][Cout] = act(im2col(X))[M][K] * W^T[K][Cout] (+bias +res1 +res2)
//   M = N*Ho*Wo, K = R*S*Cin, X is NHWC, W is OHWI (= [Cout][K], K contiguous).
//   The BatchNorm+ReLU that precedes the conv (pre-activation Bottleneck) is applied while the
//   A tile is staged (scale/shift per input channel), zero padding after the activation.
//   Epilogue: bias, up to two residual adds, and per-tile column sums / sums of squares of Y
//   (the next BatchNorm's batch statistics), so no separate pass over Y is needed.
// Weight-gradient:           dW[Cout][K] = sum_m act(im2col(X))[m][k] * dY[m][cout]
//   split over m into slabs (deterministic, no atomics), reduced by a second small kernel
//   that also produces the bias gradient.
//
// Tiling: 256 threads = 4 waves (one per SIMD); every wave owns TM x TN tiles of 32x32
// accumulators; BK = 32.  LDS tiles are k-contiguous with a 36-float pitch so that the
// ds_read_b128 fragment reads (lane (r,h) reads 4 consecutive k at row r, k-offset 4h) are
// bank-conflict free.  One barrier per K-step, global loads for step s+1 in flight during the
// MFMAs of step s (register staging: the A operand needs per-element BN/ReLU/padding).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 32
#define PITCH 36

struct ConvP {
    const float* x; const float* w; const float* bias; float* y;
    const float* in_scale; const float* in_shift;
    const float* res1; const float* res2; float* stats;
    int in_relu;
    int N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil;
    int M, K, mtiles, ntiles;
};

__device__ __forceinline__ void xcd_remap(int bid, int nwg, int& out) {
    // Blocks are dealt round-robin over the 8 XCDs; give every XCD a contiguous run of
    // tiles so neighbouring tiles (shared halo rows, shared A rows across n-tiles) meet in
    // one L2.  Bijective for any nwg (cdna guide §5, "XCD swizzle must be bijective").
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
    out = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

template <int WM, int WN, int TM, int TN, bool PRO>
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvP p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int APASS = BM / 32, BPASS = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BM][PITCH]
    float* Bs = smem + 2 * BM * PITCH;      // [2][BN][PITCH]

    int tile;
    xcd_remap(blockIdx.x, p.mtiles * p.ntiles, tile);
    const int ntile = tile % p.ntiles, mtile = tile / p.ntiles;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int lrow = tid >> 3, kc = tid & 7;

    // per-thread A rows: image base and top-left input coordinate
    int abase[APASS], aih0[APASS], aiw0[APASS];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int m = mtile * BM + lrow + 32 * i;
        if (m < p.M) {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
            abase[i] = n * p.H * p.W * p.Cin;
            aih0[i] = oh * p.stride - p.pad;
            aiw0[i] = ow * p.stride - p.pad;
        } else {
            abase[i] = 0; aih0[i] = -(1 << 28); aiw0[i] = 0;
        }
    }
    int bn_[BPASS];
#pragma unroll
    for (int j = 0; j < BPASS; ++j) bn_[j] = ntile * BN + lrow + 32 * j;

    // Staging registers.  gload() only ISSUES the global loads (unconditional, clamped addresses,
    // so they go out back-to-back and stay in flight during the MFMAs of the current step);
    // the BN+ReLU / zero-padding transform happens in lstore(), after the MFMAs.
    float4 ra[APASS], rb[BPASS];
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned okmask = 0;
    auto gload = [&](int step) {
        const int k0 = step * BK + kc * 4;
        const bool vk = k0 < p.K;
        const int tap = k0 / p.Cin, c = k0 - tap * p.Cin;
        const int r = tap / p.S, s = tap - r * p.S;
        const int dh = r * p.dil, dw = s * p.dil;
        if (PRO) {
            const int cc = vk ? c : 0;
            sc = *reinterpret_cast<const float4*>(p.in_scale + cc);
            sh = *reinterpret_cast<const float4*>(p.in_shift + cc);
        }
        okmask = 0;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int ih = aih0[i] + dh, iw = aiw0[i] + dw;
            const bool ok = vk && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            const size_t off = ok ? (size_t)abase[i] + (size_t)(ih * p.W + iw) * p.Cin + c : 0;
            ra[i] = *reinterpret_cast<const float4*>(p.x + off);
            okmask |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            const bool ok = vk && bn_[j] < p.Cout;
            const size_t off = ok ? (size_t)bn_[j] * p.K + k0 : 0;
            rb[j] = *reinterpret_cast<const float4*>(p.w + off);
            okmask |= (ok ? 1u : 0u) << (8 + j);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            float4 v = ra[i];
            if (PRO) {
                v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
                v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                if (p.in_relu) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
                    v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
            }
            if (!((okmask >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(As + (buf * BM + lrow + 32 * i) * PITCH + kc * 4) = v;
        }
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            float4 v = rb[j];
            if (!((okmask >> (8 + j)) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(Bs + (buf * BN + lrow + 32 * j) * PITCH + kc * 4) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int nsteps = (p.K + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
        const float* Ab = As + (buf * BM + (wm * TM) * 32 + lr) * PITCH + 4 * lh;
        const float* Bb = Bs + (buf * BN + (wn * TN) * 32 + lr) * PITCH + 4 * lh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a)
                fa[a] = *reinterpret_cast<const float4*>(Ab + a * 32 * PITCH + ks * 8);
#pragma unroll
            for (int b = 0; b < TN; ++b)
                fb[b] = *reinterpret_cast<const float4*>(Bb + b * 32 * PITCH + ks * 8);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the vmcnt wait + transform behind the MFMAs
        if (s + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  The accumulators (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) +
    // 4*(lane>>5)) are transposed through LDS into row-major [BM][BN] so that bias / residual /
    // store run as 16-byte row-contiguous accesses, all loads issued before the first use.
    constexpr int CP = BN + 4;                 // C-tile pitch (floats)
    float* Cs = smem;                          // [BM][CP]; the main loop ended with a barrier
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = (wn * TN + b) * 32 + lr;
            const int row0 = (wm * TM + a) * 32 + 4 * lh;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                Cs[(row0 + (e & 3) + 8 * (e >> 2)) * CP + col] = acc[a][b][e];
        }
    __syncthreads();
    constexpr int CH = BN / 4;                 // float4 chunks per row
    constexpr int RPP = 256 / CH;              // rows per pass
    constexpr int NP = BM / RPP;               // passes
    const int ch = tid % CH, r0 = tid / CH;
    const int n0 = ntile * BN + ch * 4;
    const bool vn = n0 < p.Cout;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && vn) bias4 = *reinterpret_cast<const float4*>(p.bias + n0);
    float4 r1[NP], r2[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int m = mtile * BM + r0 + RPP * j;
        const bool ok = vn && m < p.M;
        const size_t o = ok ? (size_t)m * p.Cout + n0 : 0;
        r1[j] = p.res1 ? *reinterpret_cast<const float4*>(p.res1 + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        r2[j] = p.res2 ? *reinterpret_cast<const float4*>(p.res2 + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int row = r0 + RPP * j;
        const int m = mtile * BM + row;
        if (vn && m < p.M) {
            float4 v = *reinterpret_cast<const float4*>(Cs + row * CP + ch * 4);
            v.x += bias4.x + r1[j].x + r2[j].x; v.y += bias4.y + r1[j].y + r2[j].y;
            v.z += bias4.z + r1[j].z + r2[j].z; v.w += bias4.w + r1[j].w + r2[j].w;
            *reinterpret_cast<float4*>(p.y + (size_t)m * p.Cout + n0) = v;
            s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
            s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y);
            s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
        }
    }
    if (p.stats) {
        __syncthreads();                       // every thread has read its part of Cs
        float* red = smem;                     // [RPP][BN][2]
        float* mine = red + ((size_t)r0 * BN + ch * 4) * 2;
        mine[0] = s1.x; mine[1] = s2.x; mine[2] = s1.y; mine[3] = s2.y;
        mine[4] = s1.z; mine[5] = s2.z; mine[6] = s1.w; mine[7] = s2.w;
        __syncthreads();
        if (tid < BN) {
            const int n = ntile * BN + tid;
            if (n < p.Cout) {
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int w = 0; w < RPP; ++w) {
                    a0 += red[((size_t)w * BN + tid) * 2 + 0];
                    a1 += red[((size_t)w * BN + tid) * 2 + 1];
                }
                p.stats[((size_t)mtile * 2 + 0) * p.Cout + n] = a0;
                p.stats[((size_t)mtile * 2 + 1) * p.Cout + n] = a1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// tile configuration choice (shared with the Python side through dsnt_conv_fwd_bm)
static void pick_cfg(const dsnt_conv_geom* g, int& BM, int& BN) {
    const long M = (long)g->N * g->Ho * g->Wo;
    if (g->Cout <= 32) { BM = 128; BN = 32; }
    else if (g->Cout <= 64) { BM = 128; BN = 64; }
    else { BM = 128; BN = 128; }
    // few rows: trade register blocking for more workgroups
    const long tiles = ((M + BM - 1) / BM) * ((g->Cout + BN - 1) / BN);
    if (tiles < 256 && g->Cout >= 128) { BM = 32; BN = 128; }
}

extern "C" int dsnt_conv_fwd_bm(const dsnt_conv_geom* g) {
    int BM, BN;
    pick_cfg(g, BM, BN);
    return BM;
}

template <int WM, int WN, int TM, int TN>
static int launch_fwd(const ConvP& p, bool pro, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const size_t lds = (size_t)2 * (BM + BN) * PITCH * sizeof(float);
    const int grid = p.mtiles * p.ntiles;
    // one-time opt-in to > 64 KiB of dynamic LDS (not a stream operation; safe under capture)
    static bool attr_done = false;
    if (!attr_done && lds > 65536) {
        hipFuncSetAttribute((const void*)conv_fwd_kernel<WM, WN, TM, TN, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)conv_fwd_kernel<WM, WN, TM, TN, false>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    if (pro)
        hipLaunchKernelGGL((conv_fwd_kernel<WM, WN, TM, TN, true>), dim3(grid), dim3(256), lds, st, p);
    else
        hipLaunchKernelGGL((conv_fwd_kernel<WM, WN, TM, TN, false>), dim3(grid), dim3(256), lds, st, p);
    return 0;
}

static int check_geom(const dsnt_conv_geom* g, const char* who) {
    DSNT_REQUIRE(g != nullptr, DSNT_ERR_ARG, "%s: null geometry", who);
    DSNT_REQUIRE(g->N > 0 && g->H > 0 && g->W > 0 && g->Cin > 0 && g->Cout > 0 && g->R > 0 &&
                 g->S > 0 && g->stride > 0 && g->dil > 0 && g->pad >= 0, DSNT_ERR_SHAPE,
                 "%s: non-positive dimension", who);
    DSNT_REQUIRE(g->Cin % 4 == 0, DSNT_ERR_ALIGN, "%s: Cin=%d must be a multiple of 4", who, g->Cin);
    DSNT_REQUIRE(g->Cout % 4 == 0, DSNT_ERR_ALIGN, "%s: Cout=%d must be a multiple of 4", who, g->Cout);
    const int ho = (g->H + 2 * g->pad - g->dil * (g->R - 1) - 1) / g->stride + 1;
    const int wo = (g->W + 2 * g->pad - g->dil * (g->S - 1) - 1) / g->stride + 1;
    DSNT_REQUIRE(ho == g->Ho && wo == g->Wo, DSNT_ERR_SHAPE,
                 "%s: output %dx%d inconsistent with input/filter (expected %dx%d)", who, g->Ho,
                 g->Wo, ho, wo);
    DSNT_REQUIRE((long)g->N * g->H * g->W * g->Cin < (1L << 31) &&
                 (long)g->N * g->Ho * g->Wo * g->Cout < (1L << 31), DSNT_ERR_SHAPE,
                 "%s: tensor exceeds 2^31 elements", who);
    return DSNT_OK;
}

extern "C" int dsnt_conv_fwd(const float* x, const float* w, const float* bias, float* y,
                             const float* in_scale, const float* in_shift, int in_relu,
                             const float* res1, const float* res2, float* stats_partial,
                             const dsnt_conv_geom* g, void* stream) {
    if (int e = check_geom(g, "dsnt_conv_fwd")) return e;
    DSNT_REQUIRE(x && w && y, DSNT_ERR_ARG, "dsnt_conv_fwd: null tensor");
    DSNT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), DSNT_ERR_ARG,
                 "dsnt_conv_fwd: in_scale/in_shift must be given together");
    DSNT_REQUIRE(dsnt_aligned16(x) && dsnt_aligned16(w) && (!in_scale || dsnt_aligned16(in_scale)) &&
                 (!in_shift || dsnt_aligned16(in_shift)), DSNT_ERR_ALIGN,
                 "dsnt_conv_fwd: x/w/scale/shift must be 16-byte aligned");
    ConvP p;
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.in_scale = in_scale; p.in_shift = in_shift;
    p.res1 = res1; p.res2 = res2; p.stats = stats_partial; p.in_relu = in_relu;
    p.N = g->N; p.H = g->H; p.W = g->W; p.Cin = g->Cin; p.Ho = g->Ho; p.Wo = g->Wo;
    p.Cout = g->Cout; p.R = g->R; p.S = g->S; p.stride = g->stride; p.pad = g->pad; p.dil = g->dil;
    p.M = g->N * g->Ho * g->Wo; p.K = g->R * g->S * g->Cin;
    int BM, BN;
    pick_cfg(g, BM, BN);
    p.mtiles = (p.M + BM - 1) / BM; p.ntiles = (p.Cout + BN - 1) / BN;
    hipStream_t st = (hipStream_t)stream;
    const bool pro = in_scale != nullptr;
    if (BM == 128 && BN == 128) launch_fwd<2, 2, 2, 2>(p, pro, st);
    else if (BM == 128 && BN == 64) launch_fwd<2, 2, 2, 1>(p, pro, st);
    else if (BM == 128 && BN == 32) launch_fwd<4, 1, 1, 1>(p, pro, st);
    else launch_fwd<1, 4, 1, 1>(p, pro, st);
    DSNT_CHECK_LAUNCH("dsnt_conv_fwd");
}

// wd[ci][R-1-r][S-1-s][co] = w[co][r][s][ci]
__global__ void pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wd, int Cout,
                                  int R, int S, int Cin) {
    const int total = Cout * R * S * Cin;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % Cout;
        int t = i / Cout;
        const int s = t % S; t /= S;
        const int r = t % R;
        const int ci = t / R;
        wd[i] = w[((co * R + (R - 1 - r)) * S + (S - 1 - s)) * Cin + ci];
    }
}

extern "C" int dsnt_conv_pack_dgrad(const float* w, float* wd, int Cout, int R, int S, int Cin,
                                    void* stream) {
    DSNT_REQUIRE(w && wd && Cout > 0 && R > 0 && S > 0 && Cin > 0, DSNT_ERR_ARG,
                 "dsnt_conv_pack_dgrad: bad argument");
    const int total = Cout * R * S * Cin;
    const int grid = (total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024;
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wd, Cout,
                       R, S, Cin);
    DSNT_CHECK_LAUNCH("dsnt_conv_pack_dgrad");
}

// ------------------------------------------------------------------------------------------
// Weight gradient.  GEMM view: D[k][n] = sum_m A[m][k] * G[m][n], tile 128(k) x 128(n),
// m consumed 32 rows per step.  Both LDS tiles are [32 m][128] row-major; the MFMA operands are
// read with ds_read_b32: lane (i = l&31, mm = l>>5) takes A[m = 2t+mm][k = i] and
// G[m = 2t+mm][n = i] — consecutive lanes, consecutive banks.
struct WgradP {
    const float* x; const float* in_scale; const float* in_shift; const float* dy;
    float* ws;  // [splits][Cout][K] slabs, then [splits][Cout] bias partials
    int in_relu;
    int N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil;
    int M, K, ktiles, ntiles, splits, rows_per_split;
};

#define WPITCH 132

template <bool PRO>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradP p) {
    __shared__ __attribute__((aligned(16))) float As[2][32][WPITCH];
    __shared__ __attribute__((aligned(16))) float Gs[2][32][WPITCH];

    int bid = blockIdx.x;
    const int ktile = bid % p.ktiles; bid /= p.ktiles;
    const int ntile = bid % p.ntiles;
    const int split = bid / p.ntiles;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1;   // wave owns k rows [64wk,64wk+64), n cols [64wn, ..)
    const int li = lane & 31, lm = lane >> 5;
    const int lrow = tid >> 5, cc = tid & 31;  // loader: row lrow + 8i, float4 column cc

    // A loader: this thread's k (4 consecutive) is fixed for the whole kernel
    const int k0 = ktile * 128 + cc * 4;
    const bool vk = k0 < p.K;
    const int tap = k0 / p.Cin, c = k0 - tap * p.Cin;
    const int r = tap / p.S, s = tap - r * p.S;
    const int dh = r * p.dil - p.pad, dw = s * p.dil - p.pad;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PRO && vk) {
        sc = *reinterpret_cast<const float4*>(p.in_scale + c);
        sh = *reinterpret_cast<const float4*>(p.in_shift + c);
    }
    const int n0 = ntile * 128 + cc * 4;
    const bool vn = n0 < p.Cout;   // Cout % 4 == 0 is required by the host wrapper

    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int HoWo = p.Ho * p.Wo;

    float4 ra[4], rg[4];
    unsigned okmask = 0;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    // issue-only loads (clamped addresses); transform / masking in lstore after the MFMAs
    auto gload = [&](int step) {
        okmask = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m_begin + step * 32 + lrow + 8 * i;
            const bool vm = m < m_end;
            const int mc = vm ? m : 0;
            const int n = mc / HoWo, rem = mc - n * HoWo;
            const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
            const int ih = oh * p.stride + dh, iw = ow * p.stride + dw;
            const bool oka = vm && vk && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            const size_t offa = oka ? ((size_t)(n * p.H + ih) * p.W + iw) * p.Cin + c : 0;
            ra[i] = *reinterpret_cast<const float4*>(p.x + offa);
            const bool okg = vm && vn;
            const size_t offg = okg ? (size_t)m * p.Cout + n0 : 0;
            rg[i] = *reinterpret_cast<const float4*>(p.dy + offg);
            okmask |= ((oka ? 1u : 0u) << i) | ((okg ? 1u : 0u) << (8 + i));
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 va = ra[i], vg = rg[i];
            if (PRO) {
                va.x = fmaf(va.x, sc.x, sh.x); va.y = fmaf(va.y, sc.y, sh.y);
                va.z = fmaf(va.z, sc.z, sh.z); va.w = fmaf(va.w, sc.w, sh.w);
                if (p.in_relu) {
                    va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f);
                    va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
                }
            }
            if (!((okmask >> i) & 1u)) va = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!((okmask >> (8 + i)) & 1u)) vg = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&As[buf][lrow + 8 * i][cc * 4]) = va;
            *reinterpret_cast<float4*>(&Gs[buf][lrow + 8 * i][cc * 4]) = vg;
            bsum.x += vg.x; bsum.y += vg.y; bsum.z += vg.z; bsum.w += vg.w;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int nsteps = (m_end - m_begin + 31) / 32;
    if (nsteps > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int buf = st & 1;
        if (st + 1 < nsteps) gload(st + 1);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            float fa[2], fb[2];
            fa[0] = As[buf][2 * t + lm][wk * 64 + li];
            fa[1] = As[buf][2 * t + lm][wk * 64 + 32 + li];
            fb[0] = Gs[buf][2 * t + lm][wn * 64 + li];
            fb[1] = Gs[buf][2 * t + lm][wn * 64 + 32 + li];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }

    // slab store: ws[split][n][k], D row = k (regs, 4 consecutive), D col = n (lane)
    float* slab = p.ws + (size_t)split * p.Cout * p.K;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = ntile * 128 + wn * 64 + b * 32 + li;
        if (n >= p.Cout) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = ktile * 128 + wk * 64 + a * 32 + 8 * q + 4 * lm;
                if (k < p.K) {
                    float4 v = make_float4(acc[a][b][4 * q + 0], acc[a][b][4 * q + 1],
                                           acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
                    *reinterpret_cast<float4*>(slab + (size_t)n * p.K + k) = v;
                }
            }
        }
    }
    // bias partial: column sums of this split's dY rows (only the ktile-0 blocks)
    if (ktile == 0) {
        float* red = &As[0][0][0];  // [8][128] floats
        __syncthreads();
        *reinterpret_cast<float4*>(red + lrow * 128 + cc * 4) = bsum;
        __syncthreads();
        if (tid < 128) {
            const int n = ntile * 128 + tid;
            if (n < p.Cout) {
                float t = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) t += red[j * 128 + tid];
                p.ws[(size_t)p.splits * p.Cout * p.K + (size_t)split * p.Cout + n] = t;
            }
        }
    }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                    float* __restrict__ dbias, int splits, int CK, int Cout,
                                    int accumulate) {
    const int total4 = CK / 4;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total4) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        int s = 0;
        for (; s + 8 <= splits; s += 8) {       // 8 independent 16-byte loads in flight
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = *reinterpret_cast<const float4*>(ws + (size_t)(s + u) * CK + (size_t)i * 4);
#pragma unroll
            for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
        for (; s < splits; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(ws + (size_t)s * CK + (size_t)i * 4);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        float4* o = reinterpret_cast<float4*>(dw) + i;
        if (accumulate) { const float4 c = *o; a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w; }
        *o = a;
    } else if (dbias && i - total4 < Cout) {
        const int n = i - total4;
        const float* b = ws + (size_t)splits * CK;
        float a = 0.f;
        for (int s = 0; s < splits; ++s) a += b[(size_t)s * Cout + n];
        dbias[n] = accumulate ? dbias[n] + a : a;
    }
}

static void wgrad_plan(const dsnt_conv_geom* g, int& ktiles, int& ntiles, int& splits, int& rps) {
    const long M = (long)g->N * g->Ho * g->Wo;
    const int K = g->R * g->S * g->Cin;
    ktiles = (K + 127) / 128;
    ntiles = (g->Cout + 127) / 128;
    long want = 512 / (ktiles * ntiles);      // ~2 workgroups per CU; more only inflates the slabs
    if (want < 1) want = 1;
    long max_splits = (M + 255) / 256;         // at least 8 steps of 32 rows per split
    if (max_splits < 1) max_splits = 1;
    long sp = want < max_splits ? want : max_splits;
    long rows = (M + sp - 1) / sp;
    rows = (rows + 31) / 32 * 32;
    sp = (M + rows - 1) / rows;
    splits = (int)sp;
    rps = (int)rows;
}

extern "C" int64_t dsnt_conv_wgrad_ws_floats(const dsnt_conv_geom* g) {
    if (!g) return 0;
    int kt, nt, sp, rps;
    wgrad_plan(g, kt, nt, sp, rps);
    return (int64_t)sp * g->Cout * (g->R * g->S * g->Cin) + (int64_t)sp * g->Cout;
}

extern "C" int dsnt_conv_wgrad(const float* x, const float* in_scale, const float* in_shift,
                               int in_relu, const float* dy, float* ws, float* dw, float* dbias,
                               int accumulate, const dsnt_conv_geom* g, void* stream) {
    if (int e = check_geom(g, "dsnt_conv_wgrad")) return e;
    DSNT_REQUIRE(x && dy && ws && dw, DSNT_ERR_ARG, "dsnt_conv_wgrad: null tensor");
    DSNT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), DSNT_ERR_ARG,
                 "dsnt_conv_wgrad: in_scale/in_shift must be given together");
    DSNT_REQUIRE(g->Cout % 4 == 0, DSNT_ERR_ALIGN, "dsnt_conv_wgrad: Cout=%d must be a multiple of 4",
                 g->Cout);
    DSNT_REQUIRE(dsnt_aligned16(x) && dsnt_aligned16(dy) && dsnt_aligned16(ws) && dsnt_aligned16(dw),
                 DSNT_ERR_ALIGN, "dsnt_conv_wgrad: tensors must be 16-byte aligned");
    WgradP p;
    p.x = x; p.in_scale = in_scale; p.in_shift = in_shift; p.dy = dy; p.ws = ws; p.in_relu = in_relu;
    p.N = g->N; p.H = g->H; p.W = g->W; p.Cin = g->Cin; p.Ho = g->Ho; p.Wo = g->Wo;
    p.Cout = g->Cout; p.R = g->R; p.S = g->S; p.stride = g->stride; p.pad = g->pad; p.dil = g->dil;
    p.M = g->N * g->Ho * g->Wo; p.K = g->R * g->S * g->Cin;
    wgrad_plan(g, p.ktiles, p.ntiles, p.splits, p.rows_per_split);
    hipStream_t st = (hipStream_t)stream;
    const int grid = p.ktiles * p.ntiles * p.splits;
    if (in_scale)
        hipLaunchKernelGGL(conv_wgrad_kernel<true>, dim3(grid), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL(conv_wgrad_kernel<false>, dim3(grid), dim3(256), 0, st, p);
    const int CK = p.Cout * p.K;
    const int total = CK / 4 + p.Cout;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, ws, dw, dbias,
                       p.splits, CK, p.Cout, accumulate);
    DSNT_CHECK_LAUNCH("dsnt_conv_wgrad");
}
