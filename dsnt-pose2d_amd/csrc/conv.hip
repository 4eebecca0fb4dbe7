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
#include "bn_pro.h"
#include "conv_split.h"
#include "wgrad3.h"
#include "gemm1.h"
#include "conv3s.h"
#include "stage.h"
#include "stem4.h"
#include <string.h>
#include <stdlib.h>


#define BK 32
#define PITCH 36


// Debug timeline (normally null): when set through dsnt_debug_set_timeline, lane 0 of every wave
// of workgroup `g_dbg_block` stamps s_memtime at chosen points: dbg[wave*128 + slot].
__device__ long long* g_dbg = nullptr;
__device__ int g_dbg_block = 0;
#ifdef DSNT_TIMELINE        // DSNT_TIMELINE=1 python build.py --force (tools/timeline*.py)
#define DBG_STAMP(slot)                                                                      \
    do {                                                                                     \
        if (dbg && (slot) < 126) dbg[wave * 128 + (slot)] = __builtin_amdgcn_s_memtime();   \
    } while (0)
// block < 0: every workgroup stamps into its own 8x128 slab; slots 126/127 hold HW_ID / XCC_ID
#define DBG_INIT()                                                                                   \
    long long* dbg = nullptr;                                                                        \
    if (g_dbg && lane == 0 && (g_dbg_block < 0 || (int)blockIdx.x == g_dbg_block)) {                \
        dbg = g_dbg + (g_dbg_block < 0 ? (size_t)blockIdx.x * 1024 : 0);                             \
        dbg[wave * 128 + 126] = __builtin_amdgcn_s_getreg(63492);                                    \
        dbg[wave * 128 + 127] = __builtin_amdgcn_s_getreg(63508);                                    \
    }
#define DBG_WAIT_LDS() do { if (dbg) __builtin_amdgcn_s_waitcnt(0xc07f); } while (0)
#else
#define DBG_STAMP(slot) do { } while (0)
#define DBG_INIT() do { } while (0)
#define DBG_WAIT_LDS() do { } while (0)
#endif

extern "C" int dsnt_debug_set_timeline(long long* buf, int block) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &buf, sizeof(buf)) != hipSuccess) return DSNT_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_block), &block, sizeof(block)) != hipSuccess) return DSNT_ERR_HIP;
    return DSNT_OK;
}


// Shared epilogue of the forward / data-gradient kernels (fp32 and bf16x6 variants).
// HALO: the tile's 128 rows are an 8 x 16 patch of output pixels starting at row `mbase` (row r of the
// tile is output row mbase + (r >> 4) * W + (r & 15)) instead of 128 consecutive output rows.
template <int WM, int WN, int TM, int TN, bool HALO = false, int NT = 512>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x16 (&acc)[TM][TN], float* smem, int mtile,
                                              int ntile, int tid, int wave, int lane, int mbase = 0) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    // (odd patch rows are rotated by two pixels: tile row r holds pixel column (r + 14) & 15 there, which keeps
    // the 18-pixel halo pitch on the conflict-free ds_read_b128 bank pattern — see conv3x3_bf16x6_kernel)
    auto rowmap = [&](int row) {
        return HALO ? mbase + (row >> 4) * p.W + (((row & 15) + ((row >> 4) & 1) * 14) & 15) : mtile * BM + row;
    };
    const int lr = lane & 31, lh = lane >> 5;
    const int cw = wave & 3;
    const int wm = cw / WN, wn = cw % WN;
    // ---- epilogue.  The accumulators (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) +
    // 4*(lane>>5)) are transposed through LDS into row-major [BM][BN] so that bias / residual /
    // store run as 16-byte row-contiguous accesses, all loads issued before the first use.
    constexpr int CP = BN + 4;                 // C-tile pitch (floats)
    float* Cs = smem;                          // [BM][CP]; the main loop ended with a barrier
    if (wave < 4) {
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
    }
    __syncthreads();
    constexpr int CH = BN / 4;                 // float4 chunks per row
    constexpr int RPP = NT / CH;               // rows per pass
    constexpr int NPT = BM / RPP;              // passes
    constexpr int NP = NPT > 8 ? 8 : NPT;      // passes per group (bounds the residual registers)
    constexpr int NG = NPT / NP;
    const int ch = tid % CH, r0 = tid / CH;
    const int n0 = ntile * BN + ch * 4;
    const bool vn = n0 < p.Cout;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && vn) bias4 = *reinterpret_cast<const float4*>(p.bias + n0);
    const bool bnb = p.bnb_scale != nullptr;
    float4 bsc = bias4, bsh = bias4, bmu = bias4, bis = bias4;
    if (bnb && vn) {
        bsc = *reinterpret_cast<const float4*>(p.bnb_scale + n0);
        bsh = *reinterpret_cast<const float4*>(p.bnb_shift + n0);
        bmu = *reinterpret_cast<const float4*>(p.bnb_mean + n0);
        bis = *reinterpret_cast<const float4*>(p.bnb_invstd + n0);
    } else if (p.tail.amax_bn && vn) {         // (the two uses exclude each other: the registers are shared)
        bsc = *reinterpret_cast<const float4*>(p.tail.amax_scale + n0);
        bsh = *reinterpret_cast<const float4*>(p.tail.amax_shift + n0);
    }
    const float am2lo = p.tail.amax_relu ? 0.f : -__builtin_inff();
    float am2 = 0.f;                           // max |relu?(written value * scale + shift)| (p.tail.amax_bn)
    // fp16x3: the accumulators hold (A s_a)(W s_w); both scales are powers of two, the product is undone exactly
    const float osc = p.a_bound ? 1.f / (pow2_scale(bound64(p.a_bound)) * pow2_scale(bound64(p.w_bound))) : 1.f;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    float am = 0.f;                            // max |written value| (p.tail.amax)
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
    float4 r1[NP], r2[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int m = rowmap(r0 + RPP * (gi * NP + j));
        const bool ok = vn && m < p.M;
        const size_t o = ok ? (size_t)m * p.Cout + n0 : 0;
        r1[j] = p.res1 ? *reinterpret_cast<const float4*>(p.res1 + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        r2[j] = p.res2 ? *reinterpret_cast<const float4*>(p.res2 + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int row = r0 + RPP * (gi * NP + j);
        const int m = rowmap(row);
        if (vn && m < p.M) {
            float4 v = *reinterpret_cast<const float4*>(Cs + row * CP + ch * 4);
            v.x *= osc; v.y *= osc; v.z *= osc; v.w *= osc;
            if (bnb) {
                // v = dL/d relu(bn(x)); r1[j] = x: mask by the ReLU, accumulate the BN-backward sums
                const float4 xv = r1[j];
                if (p.bnb_relu) {
                    if (fmaf(xv.x, bsc.x, bsh.x) <= 0.f) v.x = 0.f;
                    if (fmaf(xv.y, bsc.y, bsh.y) <= 0.f) v.y = 0.f;
                    if (fmaf(xv.z, bsc.z, bsh.z) <= 0.f) v.z = 0.f;
                    if (fmaf(xv.w, bsc.w, bsh.w) <= 0.f) v.w = 0.f;
                }
                *reinterpret_cast<float4*>(p.y + (size_t)m * p.Cout + n0) = v;
                s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                s2.x = fmaf(v.x, (xv.x - bmu.x) * bis.x, s2.x); s2.y = fmaf(v.y, (xv.y - bmu.y) * bis.y, s2.y);
                s2.z = fmaf(v.z, (xv.z - bmu.z) * bis.z, s2.z); s2.w = fmaf(v.w, (xv.w - bmu.w) * bis.w, s2.w);
                am = fmaxf(fmaxf(am, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));      // max |dz| (p.tail.amax)
                continue;
            }
            v.x += bias4.x + r1[j].x + r2[j].x; v.y += bias4.y + r1[j].y + r2[j].y;
            v.z += bias4.z + r1[j].z + r2[j].z; v.w += bias4.w + r1[j].w + r2[j].w;
            *reinterpret_cast<float4*>(p.y + (size_t)m * p.Cout + n0) = v;
            am = fmaxf(fmaxf(am, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
            if (p.tail.amax_bn)
                am2 = fmaxf(fmaxf(am2, fabsf(fmaxf(fmaf(v.x, bsc.x, bsh.x), am2lo))),
                            fmaxf(fabsf(fmaxf(fmaf(v.y, bsc.y, bsh.y), am2lo)),
                                  fmaxf(fabsf(fmaxf(fmaf(v.z, bsc.z, bsh.z), am2lo)), fabsf(fmaxf(fmaf(v.w, bsc.w, bsh.w), am2lo)))));
            s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
            s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y);
            s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
        }
    }
    }
    if (p.tail.amax) amax_commit(am, p.tail.amax);
    if (p.tail.amax_bn) amax_commit(am2, p.tail.amax_bn, 1);
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
                tail_store(p.stats + ((size_t)mtile * 2 + 0) * p.Cout + n, a0);
                tail_store(p.stats + ((size_t)mtile * 2 + 1) * p.Cout + n, a1);
            }
        }
    }
}

// Wave-specialised: a workgroup is 8 waves — waves 0..3 only read fragments from LDS and issue
// MFMAs (one per SIMD, 64 MFMAs = 4096 matrix-pipe cycles per K-step), waves 4..7 only move data
// (global loads two K-steps ahead, BN+ReLU / zero-padding transform, LDS stores).  The two roles
// meet at one barrier per K-step, so the matrix pipe never waits on address arithmetic, memory
// latency or the transform.  Two workgroups per CU (LDS-limited) give each SIMD two MFMA waves.
// FAST (Cin % 32 == 0, R*S*APASS <= 64, tensors < 4 GiB): the filter tap of a K-step is
// wave-uniform, so loader addresses are a per-thread constant plus a scalar — range-checked buffer
// loads need one v_add per 16-byte load and no clamping, and zero padding comes from a validity
// bit-mask computed once per thread.  This matters because on gfx950 the fp32 MFMA executes at the
// vector-FP32 rate and loader VALU instructions measurably take matrix-pipe time.
template <int WM, int WN, int TM, int TN, bool PRO, bool FAST>
__global__ __launch_bounds__(512, 4) void conv_fwd_kernel(ConvP p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int APASS = BM / 32, BPASS = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BM][PITCH]
    float* Bs = smem + 2 * BM * PITCH;      // [2][BN][PITCH]

    int tile;
    xcd_remap(blockIdx.x, p.mtiles * p.ntiles, tile);
    const int ntile = tile % p.ntiles, mtile = tile / p.ntiles;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nsteps = (p.K + BK - 1) / BK;
    const int lr = lane & 31, lh = lane >> 5;
    const int cw = wave & 3;
    const int wm = cw / WN, wn = cw % WN;
    DBG_INIT();
    DBG_STAMP(0);
    if (PRO && p.pro.partial) {          // the A operand's BatchNorm is finalised here (bn_pro.h: bn_pro_forward)
        bn_pro_forward<512>(p.pro, reinterpret_cast<double*>(smem), blockIdx.x == 0);
        __syncthreads();                 // this workgroup's stores to in_scale / in_shift are visible to its loads
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    if (FAST && wave >= 4) {
        // ------------------------------------------------------------------ loader waves (fast)
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int ltid = tid - 256;
        const int lrow = ltid >> 3, kc = ltid & 7;
        const int HoWo = p.Ho * p.Wo;
        const int RS = p.R * p.S;
        // per-thread constants: byte offset of (row i, tap (0,0), channel 4*kc) and tap validity
        unsigned apix[APASS];
        unsigned long long vmask = 0ull;            // bit tap*APASS + i
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int m = mtile * BM + lrow + 32 * i;
            const bool vm = m < p.M;
            const int mm = vm ? m : 0;
            const int n = mm / HoWo, rem = mm - n * HoWo;
            const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
            const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
            apix[i] = (unsigned)(((n * p.H + ih0) * p.W + iw0) * p.Cin + kc * 4) * 4u;
            for (int t = 0; t < RS; ++t) {
                const int r = t / p.S, s_ = t - r * p.S;
                const int ih = ih0 + r * p.dil, iw = iw0 + s_ * p.dil;
                if (vm && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                    vmask |= 1ull << (t * APASS + i);
            }
        }
        unsigned bpix[BPASS];
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            const int n = ntile * BN + lrow + 32 * j;
            bpix[j] = n < p.Cout ? (unsigned)(n * p.K + kc * 4) * 4u : 0xF0000000u;   // OOB -> 0
        }
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * p.Cin * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.w), 0, (int)((size_t)p.Cout * p.K * 4u), 0x00020000);
        struct Stage {
            u32x4 ra[APASS], rb[BPASS];
            float4 sc, sh;
            unsigned ok;       // APASS validity bits of this step's tap
        };
        Stage S0, S1;
        auto gload = [&](Stage& st, int step) {
            // all scalar: tap, channel base, byte offset of the tap relative to tap (0,0)
            const int kb = step * BK;
            const int tap = kb / p.Cin, cb = kb - tap * p.Cin;
            const int r = tap / p.S, s_ = tap - r * p.S;
            const unsigned toff = (unsigned)(((r * p.dil) * p.W + s_ * p.dil) * p.Cin + cb) * 4u;
            if (PRO) {
                st.sc = *reinterpret_cast<const float4*>(p.in_scale + cb + kc * 4);
                st.sh = *reinterpret_cast<const float4*>(p.in_shift + cb + kc * 4);
            }
            st.ok = (unsigned)(vmask >> (tap * APASS));
#pragma unroll
            for (int i = 0; i < APASS; ++i)
                st.ra[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, apix[i] + toff, 0, 0);
            const unsigned koff = (unsigned)kb * 4u;
#pragma unroll
            for (int j = 0; j < BPASS; ++j)
                st.rb[j] = __builtin_amdgcn_raw_buffer_load_b128(wr, bpix[j] + koff, 0, 0);
        };
        auto lstore = [&](const Stage& st, int buf) {
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                float4 v = make_float4(__uint_as_float(st.ra[i].x), __uint_as_float(st.ra[i].y),
                                       __uint_as_float(st.ra[i].z), __uint_as_float(st.ra[i].w));
                float* dst = As + (buf * BM + lrow + 32 * i) * PITCH + kc * 4;
                // branch-free on purpose: consuming the loaded registers inside a divergent branch makes
                // hipcc lose track of which loads have completed and drain vmcnt(0) before the next
                // prefetch is issued (seen in the ISA) — the register prefetch pipeline collapses
                if (PRO) {
                    v.x = fmaf(v.x, st.sc.x, st.sh.x); v.y = fmaf(v.y, st.sc.y, st.sh.y);
                    v.z = fmaf(v.z, st.sc.z, st.sh.z); v.w = fmaf(v.w, st.sc.w, st.sh.w);
                    if (p.in_relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                }
                const bool ok = (st.ok >> i) & 1u;
                v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
                *reinterpret_cast<float4*>(dst) = v;
            }
#pragma unroll
            for (int j = 0; j < BPASS; ++j)
                *reinterpret_cast<u32x4*>(Bs + (buf * BN + lrow + 32 * j) * PITCH + kc * 4) = st.rb[j];
        };
        // no conditionals around gload/lstore (see the bf16x6 loader): the tail re-loads the last step
        const int last = nsteps - 1;
        gload(S0, 0);
        gload(S1, min(1, last));
        lstore(S0, 0);
        gload(S0, min(2, last));
        DBG_STAMP(1);
        __syncthreads();
        int s = 0;
        for (; s + 1 < nsteps; s += 2) {
            DBG_STAMP(2 + 3 * s);
            lstore(S1, 1);
            DBG_STAMP(3 + 3 * s);
            gload(S1, min(s + 3, last));
            DBG_STAMP(4 + 3 * s);
            __syncthreads();
            DBG_STAMP(5 + 3 * s);
            lstore(S0, 0);                 // s + 2 == nsteps: refills the idle buffer 0, harmless
            DBG_STAMP(6 + 3 * s);
            gload(S0, min(s + 4, last));
            DBG_STAMP(7 + 3 * s);
            __syncthreads();
        }
        if (s < nsteps) __syncthreads();
    } else if (wave >= 4) {
        // ------------------------------------------------------------------ loader waves (general)
        const int ltid = tid - 256;
        const int lrow = ltid >> 3, kc = ltid & 7;
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
        int boff[BPASS];
        bool bok[BPASS];
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            const int n = ntile * BN + lrow + 32 * j;
            bok[j] = n < p.Cout;
            boff[j] = bok[j] ? n * p.K : 0;
        }
        // Two register sets: the loads of K-step s+2 and s+3 are in flight while s+1 is stored,
        // i.e. every global load has two full K-steps (>= 8k matrix-pipe cycles) to land.
        struct Stage {
            float4 ra[APASS], rb[BPASS];
            float4 sc, sh;
            unsigned okmask;
        };
        Stage S0, S1;
        // issue-only: unconditional loads from clamped addresses, nothing consumed here
        auto gload = [&](Stage& st, int step) {
            const int k0 = step * BK + kc * 4;
            const bool vk = k0 < p.K;
            const int tap = k0 / p.Cin, c = k0 - tap * p.Cin;
            const int r = tap / p.S, s = tap - r * p.S;
            const int dh = r * p.dil, dw = s * p.dil;
            if (PRO) {
                const int cc = vk ? c : 0;
                st.sc = *reinterpret_cast<const float4*>(p.in_scale + cc);
                st.sh = *reinterpret_cast<const float4*>(p.in_shift + cc);
            }
            st.okmask = 0;
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                const int ih = aih0[i] + dh, iw = aiw0[i] + dw;
                const bool ok = vk && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                const int off = ok ? abase[i] + (ih * p.W + iw) * p.Cin + c : 0;
                st.ra[i] = *reinterpret_cast<const float4*>(p.x + off);
                st.okmask |= (ok ? 1u : 0u) << i;
            }
#pragma unroll
            for (int j = 0; j < BPASS; ++j) {
                const bool ok = vk && bok[j];
                st.rb[j] = *reinterpret_cast<const float4*>(p.w + (ok ? boff[j] + k0 : 0));
                st.okmask |= (ok ? 1u : 0u) << (8 + j);
            }
        };
        auto lstore = [&](const Stage& st, int buf) {
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                float4 v = st.ra[i];
                if (PRO) {
                    v.x = fmaf(v.x, st.sc.x, st.sh.x); v.y = fmaf(v.y, st.sc.y, st.sh.y);
                    v.z = fmaf(v.z, st.sc.z, st.sh.z); v.w = fmaf(v.w, st.sc.w, st.sh.w);
                    if (p.in_relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                }
                if (!((st.okmask >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(As + (buf * BM + lrow + 32 * i) * PITCH + kc * 4) = v;
            }
#pragma unroll
            for (int j = 0; j < BPASS; ++j) {
                float4 v = st.rb[j];
                if (!((st.okmask >> (8 + j)) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(Bs + (buf * BN + lrow + 32 * j) * PITCH + kc * 4) = v;
            }
        };
        // no conditionals around gload/lstore (see the bf16x6 loader): the tail re-loads the last step
        const int last = nsteps - 1;
        gload(S0, 0);
        gload(S1, min(1, last));
        lstore(S0, 0);
        gload(S0, min(2, last));
        DBG_STAMP(1);
        __syncthreads();
        // iteration s stores K-step s+1 (held in S1 for even s, S0 for odd s) into the buffer the
        // MFMA waves left in iteration s-1, then refills that register set with K-step s+3
        int s = 0;
        for (; s + 1 < nsteps; s += 2) {
            DBG_STAMP(2 + 3 * s);
            lstore(S1, 1);
            DBG_STAMP(3 + 3 * s);
            gload(S1, min(s + 3, last));
            DBG_STAMP(4 + 3 * s);
            __syncthreads();
            DBG_STAMP(5 + 3 * s);
            lstore(S0, 0);                 // s + 2 == nsteps: refills the idle buffer 0, harmless
            DBG_STAMP(6 + 3 * s);
            gload(S0, min(s + 4, last));
            DBG_STAMP(7 + 3 * s);
            __syncthreads();
        }
        if (s < nsteps) __syncthreads();     // odd step count: the last iteration only synchronises
    } else {
        // ------------------------------------------------------------------ MFMA waves
        __builtin_amdgcn_s_setprio(1);
        // Software-pipelined fragment reads: the ds_reads of k-group g+1 are issued before the 16
        // MFMAs of group g (two fragment register sets), and the barrier of a K-step sits in front
        // of its LAST group, so the first reads of the next step are already in flight while that
        // group's MFMAs run.  The matrix pipe then only idles for the barrier skew.
        struct Frag { float4 a[TM], b[TN]; };
        Frag F0, F1;
        auto rd = [&](Frag& f, int buf, int ks) {
            const float* Ab = As + (buf * BM + (wm * TM) * 32 + lr) * PITCH + 4 * lh + ks * 8;
            const float* Bb = Bs + (buf * BN + (wn * TN) * 32 + lr) * PITCH + 4 * lh + ks * 8;
#pragma unroll
            for (int a = 0; a < TM; ++a) f.a[a] = *reinterpret_cast<const float4*>(Ab + a * 32 * PITCH);
#pragma unroll
            for (int b = 0; b < TN; ++b) f.b[b] = *reinterpret_cast<const float4*>(Bb + b * 32 * PITCH);
        };
        auto mm = [&](const Frag& f) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[a].x, f.b[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[a].y, f.b[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[a].z, f.b[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[a].w, f.b[b].w, acc[a][b], 0, 0, 0);
                }
        };
        __syncthreads();
        rd(F0, 0, 0);
        for (int s = 0; s < nsteps; ++s) {
            const int buf = s & 1;
            DBG_STAMP(1 + 2 * s);
            rd(F1, buf, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(F0);
            __builtin_amdgcn_sched_barrier(0);
            rd(F0, buf, 2);
            __builtin_amdgcn_sched_barrier(0);
            mm(F1);
            __builtin_amdgcn_sched_barrier(0);
            rd(F1, buf, 3);
            __builtin_amdgcn_sched_barrier(0);
            mm(F0);
            __builtin_amdgcn_sched_barrier(0);
            DBG_STAMP(2 + 2 * s);
            __syncthreads();                       // all of this step's LDS reads have landed
            if (s + 1 < nsteps) rd(F0, buf ^ 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mm(F1);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    }
    DBG_STAMP(124);

    conv_epilogue<WM, WN, TM, TN>(p, acc, smem, mtile, ntile, tid, wave, lane);
    DBG_STAMP(125);
}

// ==========================================================================================
// bf16x6: fp32-accurate convolution on the bf16 matrix cores.
// Every fp32 operand is split exactly into three bf16 planes x = x1 + x2 + x3 (8 + 8 + 8 mantissa
// bits); the product keeps the six terms of order <= 2^-16 (x1y1, x1y2, x2y1, x1y3, x2y2, x3y1),
// accumulated in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  The dropped terms are
// <= 2^-23 |xy| — below one fp32 rounding — so results match the fp32 kernel to fp32 accuracy
// (measured 2.4e-7 vs 5.4e-7 for a plain fp32 GEMM, K = 1152), at 6/16 of the fp32-MFMA cost.
// Weights are pre-split once per step (dsnt_split_bf16x3); activations are transformed
// (BN + ReLU, zero padding) and split by the loader waves while staging.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define BK6 16
#define PITCH6 24        // bf16 per LDS row: 16 data + 8 pad = 48 B -> conflict-free ds_read_b128

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// 4 floats -> three planes of 4 bf16 (2 dwords each): exact 3-way split, 5.5 VALU instructions per
// element.  Deliberately NOT on packed fp32 ops: beside MFMAs a v_pk_add_f32 / v_pk_fma_f32 costs more
// issue time than the two plain instructions it replaces (MI355X_MICROARCH.md, cycle constants), and
// these kernels are bound by the SIMD's vector-issue port (conv.hip is built with -fno-slp-vectorize).
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(const float4 v, uint2& p1, uint2& p2, uint2& p3) {
    p1.x = pk_bf16(v.x, v.y); p1.y = pk_bf16(v.z, v.w);
    float4 r;
    r.x = v.x - __uint_as_float(p1.x << 16); r.y = v.y - __uint_as_float(p1.x & 0xffff0000u);
    r.z = v.z - __uint_as_float(p1.y << 16); r.w = v.w - __uint_as_float(p1.y & 0xffff0000u);
    p2.x = pk_bf16(r.x, r.y); p2.y = pk_bf16(r.z, r.w);
    r.x -= __uint_as_float(p2.x << 16); r.y -= __uint_as_float(p2.x & 0xffff0000u);
    r.z -= __uint_as_float(p2.y << 16); r.w -= __uint_as_float(p2.y & 0xffff0000u);
    p3.x = pk_bf16(r.x, r.y); p3.y = pk_bf16(r.z, r.w);
}

// ------------------------------------------------------------------------------------------
// fp16x3: the same idea on TWO fp16 planes after a power-of-two scale: x * s = h1 + h2 with h1 = fp16(x * s),
// h2 = fp16(x * s - h1) keeps 22+ significand bits, the product needs h1 g1 + h1 g2 + h2 g1 = THREE MFMAs (dropped
// term <= 2^-24 |x g|), two thirds of the LDS traffic and about half the split arithmetic of bf16x6.  Error against
// fp64 (K = 1152, tools/split_numerics.py): 7.7e-8 of the output scale — a plain fp32 GEMM has 2.7e-7, bf16x6 5.8e-9.
// The price is fp16's exponent range: s = pow2_scale(bound) keeps |x * s| < 2^14 for any bound >= max|x| (a bound
// 64x too large costs nothing measurable; overflow would be fatal, underflow only costs absolute error
// <= 2^-40 * bound).  Bounds live in device memory: BN+ReLU operands from the BN parameters (|gamma| sqrt(M) + |beta|),
// weights and BN-backward outputs from an amax their producer wrote.
// the MFMAs of one 32x32 accumulator and one 16-wide K step, smallest terms first
template <bool F16>
__device__ __forceinline__ void mma_split(f32x16& acc, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
    if (F16) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, b[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[0]), acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
}

__global__ void split_bf16x3_kernel(const float4* __restrict__ src, uint2* __restrict__ dst, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        uint2 a, b, c;
        split4(src[i], a, b, c);
        dst[i] = a; dst[n4 + i] = b; dst[2 * n4 + i] = c;
    }
}

extern "C" int dsnt_split_bf16x3(const float* src, void* dst, int64_t n, void* stream) {
    DSNT_REQUIRE(src && dst && n > 0 && n % 4 == 0, DSNT_ERR_ARG, "dsnt_split_bf16x3: n must be a positive multiple of 4");
    DSNT_REQUIRE(dsnt_aligned16(src) && (((uintptr_t)dst) & 7u) == 0, DSNT_ERR_ALIGN, "dsnt_split_bf16x3: alignment");
    const long n4 = n / 4;
    long g = (n4 + 255) / 256;
    if (g > 2048) g = 2048;
    DSNT_LAUNCH(split_bf16x3_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)src, (uint2*)dst, n4);
    DSNT_CHECK_LAUNCH("dsnt_split_bf16x3");
}

template <int WM, int WN, int TM, int TN, bool PRO, bool F16 = false, int DA = 4>
__global__ __launch_bounds__(512, 2) void conv_fwd_bf16x6_kernel(ConvP p) {
    constexpr int NPL = F16 ? 2 : 3;            // operand planes (fp16x3: two fp16 planes, three MFMAs)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int APASS = BM / 64;             // loader: 64 rows x 4 float4 chunks per pass
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* A6 = reinterpret_cast<__bf16*>(smem);          // [2][NPL][BM][PITCH6]
    __bf16* B6 = A6 + 2 * NPL * BM * PITCH6;               // [2][NPL][BN][PITCH6]
    float* SS = reinterpret_cast<float*>(B6 + 2 * NPL * BN * PITCH6);   // [2][Cin]: BN scale / shift of the A operand

    int tile;
    xcd_remap(blockIdx.x, p.mtiles * p.ntiles, tile);
    const int ntile = tile % p.ntiles, mtile = tile / p.ntiles;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nsteps = p.K / BK6;
    const int lr = lane & 31, lh = lane >> 5;
    const int cw = wave & 3;
    const int wm = cw / WN, wn = cw % WN;
    DBG_INIT();
    DBG_STAMP(0);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    if (wave >= 4) {
        // ------------------------------------------------------------------ loader waves
        const int ltid = tid - 256;
        // 16 consecutive lanes store 4 float4-chunks of rows {r, r+2, r+4, r+6}: with the 48-byte row
        // pitch those 16 ds_write_b64 hit 16 distinct 8-byte bank groups (rows r..r+3 would 2-way conflict)
        const int kc = ltid & 3;
        const int lrow = ((ltid >> 5) << 3) + (((ltid >> 2) & 3) << 1) + ((ltid >> 4) & 1);
        const int HoWo = p.Ho * p.Wo;
        const int RS = p.R * p.S;
        unsigned apix[APASS];
        unsigned vmask = 0;                          // bit tap*APASS + i
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int m = mtile * BM + lrow + 64 * i;
            const bool vm = m < p.M;
            const int mm = vm ? m : 0;
            const int n = mm / HoWo, rem = mm - n * HoWo;
            const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
            const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
            apix[i] = (unsigned)(((n * p.H + ih0) * p.W + iw0) * p.Cin + kc * 4) * 4u;
            for (int t = 0; t < RS; ++t) {
                const int r = t / p.S, s_ = t - r * p.S;
                const int ih = ih0 + r * p.dil, iw = iw0 + s_ * p.dil;
                if (vm && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) vmask |= 1u << (t * APASS + i);
            }
        }
        // weights: 16-byte chunk `ltid` of each plane's [BN][16] slice: row = ltid>>1, half = ltid&1
        const int bhalf = ltid & 1;     // same idea for the 16-byte weight stores (8-lane groups)
        const int brow = ((ltid >> 4) << 3) + (((ltid >> 1) & 3) << 1) + ((ltid >> 3) & 1);
        const int bn = ntile * BN + brow;
        const bool bvalid = brow < BN && bn < p.Cout;
        unsigned bpix[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j)
            bpix[j] = bvalid ? (unsigned)((size_t)j * p.wq_stride + (size_t)bn * p.K + bhalf * 8) * 2u : 0xF0000000u;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * p.Cin * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned short*>(p.wq), 0, (int)(((size_t)(NPL - 1) * p.wq_stride + (size_t)p.Cout * p.K) * 2u), 0x00020000);
        // fp16x3: operand scale (a power of two) from the bound the producer left in device memory
        const float sa = F16 ? pow2_scale(bound64(p.a_bound)) : 1.f;
        // The A operand is what this kernel waits for: with K = 128 ... 256 a tile is 8 ... 16 K-steps, each a fresh
        // 8 KB slice of activations from HBM, and the chip-wide bytes in flight bound the bandwidth (Little's law: 512
        // resident workgroups x 2 stages x 8 KB = 8 MB gave 3.6 TB/s at ~2.2 us loaded latency).  DA register stages
        // keep DA slices per workgroup in flight; the weights (L2 hits) stay on two stages; the BatchNorm scale / shift
        // vectors live in LDS (copied once, pre-multiplied by the fp16x3 operand scale) instead of riding in every stage.
        struct AStage { u32x4 ra[APASS]; };
        struct BStage { u32x4 rb[NPL]; };
        AStage SA[DA];
        BStage SB[2];
        const float lo_valid = p.in_relu ? 0.f : -__builtin_inff();
        const int last = nsteps - 1;
        // K-step bookkeeping without divisions: the loads (A: DA steps ahead, B: two ahead) and the stores walk the
        // K-steps in order, so each keeps running scalars (channel base, filter tap) advanced branch-free and frozen
        // at the last step (the tail re-loads / re-stores it: never read, or into the idle buffer).  `kb / Cin` and
        // `tap / S` per call were ~100 instructions of emulated integer division per step on the waves the MFMA
        // waves wait for.
        struct Walk { int step, cb, r, s_; };
        Walk wa = {0, 0, 0, 0}, ws = {0, 0, 0, 0};
        int wb_step = 0;
        auto advance = [&](Walk& w) {
            const int adv = w.step < last ? 1 : 0;
            w.step += adv;
            w.cb += adv * BK6;
            const int wrap = w.cb >= p.Cin ? 1 : 0;
            w.cb = wrap ? 0 : w.cb;
            w.s_ += wrap;
            const int wrap2 = w.s_ == p.S ? 1 : 0;
            w.s_ = wrap2 ? 0 : w.s_;
            w.r += wrap2;
        };
        auto gloadA = [&](AStage& st) {
            const unsigned toff = (unsigned)(((wa.r * p.dil) * p.W + wa.s_ * p.dil) * p.Cin + wa.cb) * 4u;
#pragma unroll
            for (int i = 0; i < APASS; ++i)
                st.ra[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, apix[i] + toff, 0, 0);
            advance(wa);
        };
        auto gloadB = [&](BStage& st) {
            const unsigned koff = (unsigned)(wb_step * BK6) * 2u;
#pragma unroll
            for (int j = 0; j < NPL; ++j)
                st.rb[j] = __builtin_amdgcn_raw_buffer_load_b128(wr, bpix[j] + koff, 0, 0);
            wb_step += wb_step < last ? 1 : 0;
        };
        auto lstore = [&](const AStage& sa_, const BStage& sb_, int buf) {
            const int cb = ws.cb;
            const unsigned okm = vmask >> ((ws.r * p.S + ws.s_) * APASS);
            advance(ws);
            float4 sc, sh;
            if (PRO) {
                sc = *reinterpret_cast<const float4*>(SS + cb + kc * 4);
                sh = *reinterpret_cast<const float4*>(SS + p.Cin + cb + kc * 4);
            }
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                float4 v = make_float4(__uint_as_float(sa_.ra[i].x), __uint_as_float(sa_.ra[i].y),
                                       __uint_as_float(sa_.ra[i].z), __uint_as_float(sa_.ra[i].w));
                uint2 q1, q2, q3;
                // branch-free zero padding (a divergent branch around the loaded registers makes hipcc
                // drain vmcnt(0) before the next prefetch: see the fp32 loader)
                const bool ok = (okm >> i) & 1u;
                if (PRO) {
                    // BN FMAs; ReLU and the padding select as ONE median per element:
                    // valid rows clamp to [0 or -inf, +inf), padded rows to [0, 0]
                    const sp_f32x2 a = {fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y)};
                    const sp_f32x2 b = {fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w)};
                    const float lo = ok ? lo_valid : 0.f, hi = ok ? __builtin_inff() : 0.f;
                    v.x = __builtin_amdgcn_fmed3f(a.x, lo, hi); v.y = __builtin_amdgcn_fmed3f(a.y, lo, hi);
                    v.z = __builtin_amdgcn_fmed3f(b.x, lo, hi); v.w = __builtin_amdgcn_fmed3f(b.y, lo, hi);
                } else {
                    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
                }
                __bf16* dst = A6 + ((size_t)(buf * NPL) * BM + lrow + 64 * i) * PITCH6 + kc * 4;
                if (F16) {
                    if (!PRO) { v.x *= sa; v.y *= sa; v.z *= sa; v.w *= sa; }
                    split4h(v, q1, q2);
                    *reinterpret_cast<uint2*>(dst) = q1;
                    *reinterpret_cast<uint2*>(dst + BM * PITCH6) = q2;
                } else {
                    split4(v, q1, q2, q3);
                    *reinterpret_cast<uint2*>(dst) = q1;
                    *reinterpret_cast<uint2*>(dst + BM * PITCH6) = q2;
                    *reinterpret_cast<uint2*>(dst + 2 * BM * PITCH6) = q3;
                }
            }
            if (brow < BN) {
#pragma unroll
                for (int j = 0; j < NPL; ++j)
                    *reinterpret_cast<u32x4*>(B6 + ((size_t)(buf * NPL + j) * BN + brow) * PITCH6 + bhalf * 8) = sb_.rb[j];
            }
        };
        // No conditionals around the loads and stores of the main loop: hipcc's vmcnt bookkeeping is exact only on
        // straight-line code (a guarded prefetch made it wait for vmcnt(0) before every LDS store).  Phase i = 1 .. nsteps
        // stores K-step i into buffer i & 1 (the MFMA waves are on step i - 1), then refills the A stage with step
        // i + DA and the B stage with step i + 2; the loop is unrolled over P = lcm(DA, 2) phases so that stage and
        // buffer indices are compile-time constants.
        constexpr int P = (DA % 2 == 0) ? DA : 2 * DA;
#pragma unroll
        for (int d = 0; d < DA; ++d) gloadA(SA[d]);
        gloadB(SB[0]);
        gloadB(SB[1]);
        if (PRO) {                                   // the BatchNorm vectors -> LDS, once
            for (int c = ltid; c < p.Cin; c += 256) {
                SS[c] = p.in_scale[c] * sa;
                SS[p.Cin + c] = p.in_shift[c] * sa;
            }
        }
        __syncthreads();                             // (all eight waves) SS is in place
        lstore(SA[0], SB[0], 0);
        gloadA(SA[0]);
        gloadB(SB[0]);
        __syncthreads();
        int i = 1;
        for (; i + P - 1 <= nsteps; i += P) {
#pragma unroll
            for (int u = 0; u < P; ++u) {
                DBG_STAMP(1 + 3 * (i + u - 1));
                lstore(SA[(1 + u) % DA], SB[(1 + u) & 1], (1 + u) & 1);
                DBG_STAMP(2 + 3 * (i + u - 1));
                gloadA(SA[(1 + u) % DA]);
                gloadB(SB[(1 + u) & 1]);
                DBG_STAMP(3 + 3 * (i + u - 1));
                __syncthreads();
            }
        }
        // up to P - 1 phases left (uniform branches; nothing is prefetched any more)
#pragma unroll
        for (int u = 0; u < P - 1; ++u)
            if (i + u <= nsteps) {
                lstore(SA[(1 + u) % DA], SB[(1 + u) & 1], (1 + u) & 1);
                __syncthreads();
            }
    } else {
        // ------------------------------------------------------------------ MFMA waves
        struct Frag { bf16x8 a[TM][3], b[TN][3]; };
        Frag F;
        auto rd = [&](Frag& f, int buf) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
                    f.a[a][pl] = *reinterpret_cast<const bf16x8*>(
                        A6 + ((size_t)(buf * NPL + pl) * BM + (wm * TM + a) * 32 + lr) * PITCH6 + 8 * lh);
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    f.b[b][pl] = *reinterpret_cast<const bf16x8*>(
                        B6 + ((size_t)(buf * NPL + pl) * BN + (wn * TN + b) * 32 + lr) * PITCH6 + 8 * lh);
            }
        };
        auto mm = [&](const Frag& f) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) mma_split<F16>(acc[a][b], f.a[a], f.b[b]);
        };
        __syncthreads();                   // the loaders' BatchNorm vectors are in LDS
        __syncthreads();                   // K-step 0 is staged
        for (int s = 0; s < nsteps; ++s) {
            DBG_STAMP(1 + 3 * s);
            rd(F, s & 1);
            DBG_WAIT_LDS();                // timeline builds only: fragment reads landed
            DBG_STAMP(2 + 3 * s);
            mm(F);
            DBG_STAMP(3 + 3 * s);
            __syncthreads();
        }
    }
    DBG_STAMP(124);
    conv_epilogue<WM, WN, TM, TN>(p, acc, smem, mtile, ntile, tid, wave, lane);
    DBG_STAMP(125);
}


extern "C" int dsnt_conv_bf16x6_ok(const dsnt_conv_geom* g) {
    if (!g) return 0;
    return g->Cin % BK6 == 0 && g->Cout % 4 == 0 && g->R * g->S * 2 <= 32 &&
           (size_t)g->N * g->H * g->W * g->Cin * 4u < (1ull << 31) &&
           (size_t)3 * g->Cout * g->R * g->S * g->Cin * 2u < (1ull << 31);
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
    if (lds > 65536) {
        DSNT_SET_MAX_LDS((conv_fwd_kernel<WM, WN, TM, TN, true, true>), lds);
        DSNT_SET_MAX_LDS((conv_fwd_kernel<WM, WN, TM, TN, false, true>), lds);
        DSNT_SET_MAX_LDS((conv_fwd_kernel<WM, WN, TM, TN, true, false>), lds);
        DSNT_SET_MAX_LDS((conv_fwd_kernel<WM, WN, TM, TN, false, false>), lds);
    }
    const bool fast = (p.Cin % BK == 0) && (p.R * p.S * (BM / 32) <= 64) &&
                      ((size_t)p.N * p.H * p.W * p.Cin * 4u < (1ull << 31)) &&
                      ((size_t)p.Cout * p.K * 4u < (1ull << 31));
    dim3 gr(grid), bl(512);
    if (pro && fast) DSNT_LAUNCH((conv_fwd_kernel<WM, WN, TM, TN, true, true>), gr, bl, lds, st, p);
    else if (pro) DSNT_LAUNCH((conv_fwd_kernel<WM, WN, TM, TN, true, false>), gr, bl, lds, st, p);
    else if (fast) DSNT_LAUNCH((conv_fwd_kernel<WM, WN, TM, TN, false, true>), gr, bl, lds, st, p);
    else DSNT_LAUNCH((conv_fwd_kernel<WM, WN, TM, TN, false, false>), gr, bl, lds, st, p);
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

// ------------------------------------------------------------------------------------------
// Few output rows (the 8x8 and 4x4 hourglass levels: M = 2048 / 512 at batch 32): the tiled kernels above
// put one 32 x 32 accumulator per wave behind the WHOLE reduction (3x3 128->128: 576 dependent
// v_mfma_f32_32x32x2_f32 = 37 k cycles = ~18 us whatever the tile shape) on 16 ... 64 workgroups, i.e. most
// SIMDs idle.  Here a 512-thread workgroup owns ONE 32 x 32 output tile and its eight waves split K:
// every wave streams its K slice straight from global memory into MFMA operand registers (lane (i, h) holds
// row i, k = 8 j + 4 h .. + 3 of the A rows and of the weight rows: one 16-byte load each per four MFMAs, no
// LDS staging, no barrier in the loop), the eight partial tiles are summed through LDS in wave order
// (deterministic) and the usual epilogue (bias, residuals, statistics, BN-backward masking) runs on the sum.
// Same contract and statistics layout (32-row tiles) as conv_fwd_kernel<1, 4, 1, 1>.
// CW = k values per chunk and row (8, 16 or 32).  A lane (i, h) owns row i and CW / 2 consecutive k values of a chunk:
// CW / 8 16-byte loads per operand, issued back to back.  With CW = 8 the two lanes of a row use 32 bytes of a
// 128-byte line per load instruction, the next 32 bytes a whole MFMA batch (and 15 other waves' loads) later: the L1
// (32 KB against 16 waves x 64 lines in flight) has dropped the line by then, every chunk re-fetches it from L2, and the
// loads cost L1 fills at 4x the operand bytes on top of 32 tag look-ups per instruction (a lane is a row: the MFMA operand
// layout), which together take about as long as the MFMAs (3x3 256->256 at M = 2048: 49 us for 18 us of matrix pipe;
// deeper prefetch makes it worse: 57 / 70 us with 4 / 6 chunks in flight).  Measured, CW = 8 / 16 / 32: that launch 49 /
// 45 / 49 us, 3x3 512->512 at M = 512 51 / 43 / 41 us, the hourglass's 128-channel forms 18 / 16.5 / 18.8 us (CW = 32
// holds 96 load registers: occupancy 3-4 waves per SIMD); resnet34 batch 8 5.18 / 5.02 / 5.05 ms per step, hg2 batch 32
// 13.58 / 13.57 / 13.74 ms.  CW = 16 ships.  Needs Cin % CW == 0 (a chunk never straddles a filter tap); else CW = 8.
// The kernel's body is a device function of a VIRTUAL workgroup index `vb`: the stand-alone launch passes blockIdx.x, the
// persistent low-resolution stage (stage.h, the end of this file) walks the same indices from a loop — same instructions, same
// summation order, bit-identical results.  `part`: 8 x 32 x 33 floats of LDS; `finalise`: false when THIS workgroup has already run
// the prologue's finalisation for this launch (a stage workgroup that takes a second tile: the vectors are in memory).
template <bool PRO, int CW>
__device__ __forceinline__ void conv_ksplit_body(const ConvP& p, const int vb, float (*part)[32][33], const bool finalise) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int NL = CW / 8;                      // 16-byte loads per operand, lane and chunk
    const int nt32 = (p.Cout + 31) >> 5;
    const int ntile = vb % nt32, mtile = vb / nt32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const unsigned OOB = 0xF0000000u;
    if (PRO && p.pro.partial && finalise) {          // the A operand's BatchNorm is finalised here (bn_pro.h: bn_pro_forward)
        bn_pro_forward<512>(p.pro, reinterpret_cast<double*>(&part[0][0][0]), vb == 0);
        __syncthreads();                 // this workgroup's stores to in_scale / in_shift are visible to its loads
    }
    // the BatchNorm vectors of the A operand live in LDS during the loop (in `part`, which is only written after it; Cin <=
    // 4096): as global loads inside the loop they would queue behind the prefetched chunks (loads return in order)
    float* const ssc = &part[0][0][0];
    if (PRO) {
        for (int c = tid; c < p.Cin; c += 512) { ssc[c] = p.in_scale[c]; ssc[p.Cin + c] = p.in_shift[c]; }
        __syncthreads();
    }
    // A row of this lane
    const int m = mtile * 32 + i;
    const bool vm = m < p.M;
    const int HoWo = p.Ho * p.Wo;
    const int mm = vm ? m : 0;
    const int img = mm / HoWo, rem = mm - img * HoWo;
    const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
    const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
    // weight row of this lane
    const int nb = ntile * 32 + i;
    const int kl = (CW / 2) * h;                    // this lane's k offset inside a chunk
    const unsigned boff = nb < p.Cout ? (unsigned)((size_t)nb * p.K + kl) * 4u : OOB;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * p.Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0, (int)((size_t)p.Cout * p.K * 4u), 0x00020000);
    const int nch = p.K / CW;                       // a chunk never straddles a filter tap
    const int c0 = wave * nch / 8, c1 = (wave + 1) * nch / 8;
    const float lo_valid = p.in_relu ? 0.f : -__builtin_inff();
    struct Frag { u32x4 a[NL], b[NL]; bool ok; int cb; };
    // position of the next chunk to load (scalars; stepped, not divided: the loop's issue slots belong to the MFMAs —
    // fp32 MFMAs and VALU instructions exclude each other on a SIMD, profiles/r03_pmc_ksplit.txt); it stops at the
    // wave's last chunk, which the tail of the loop re-loads (never used) to stay straight-line
    int pos = c0, pr, ps, pcb;
    {
        const int kb = c0 * CW, tap = kb / p.Cin;
        pcb = kb - tap * p.Cin; pr = tap / p.S; ps = tap - pr * p.S;
    }
    auto load_next = [&]() {
        Frag f;
        const int ih = ih0 + pr * p.dil, iw = iw0 + ps * p.dil;
        f.ok = vm && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        f.cb = pcb;
        const unsigned aoff = f.ok ? (unsigned)(((img * p.H + ih) * p.W + iw) * p.Cin + pcb + kl) * 4u : OOB;
        const int kb4 = ((pr * p.S + ps) * p.Cin + pcb) * 4;
#pragma unroll
        for (int q = 0; q < NL; ++q) f.a[q] = __builtin_amdgcn_raw_buffer_load_b128(xr, aoff + 16u * q, 0, 0);
#pragma unroll
        for (int q = 0; q < NL; ++q) f.b[q] = __builtin_amdgcn_raw_buffer_load_b128(wr, boff + 16u * q, kb4, 0);
        if (pos < c1 - 1) {
            ++pos;
            pcb += CW;
            if (pcb == p.Cin) { pcb = 0; if (++ps == p.S) { ps = 0; ++pr; } }
        }
        return f;
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    auto mma = [&](const Frag& f) {
        const float lo = f.ok ? lo_valid : 0.f, hi = f.ok ? __builtin_inff() : 0.f;
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            float4 a = make_float4(__uint_as_float(f.a[q].x), __uint_as_float(f.a[q].y), __uint_as_float(f.a[q].z),
                                   __uint_as_float(f.a[q].w));
            if (PRO) {
                const float4 sc = *reinterpret_cast<const float4*>(ssc + f.cb + kl + 4 * q);
                const float4 sh = *reinterpret_cast<const float4*>(ssc + p.Cin + f.cb + kl + 4 * q);
                a.x = __builtin_amdgcn_fmed3f(fmaf(a.x, sc.x, sh.x), lo, hi);
                a.y = __builtin_amdgcn_fmed3f(fmaf(a.y, sc.y, sh.y), lo, hi);
                a.z = __builtin_amdgcn_fmed3f(fmaf(a.z, sc.z, sh.z), lo, hi);
                a.w = __builtin_amdgcn_fmed3f(fmaf(a.w, sc.w, sh.w), lo, hi);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, __uint_as_float(f.b[q].x), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, __uint_as_float(f.b[q].y), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, __uint_as_float(f.b[q].z), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, __uint_as_float(f.b[q].w), acc, 0, 0, 0);
        }
    };
    // the epilogue's operands (two rows per thread: rg, rg + 16 of column `col`) are fetched NOW, under the K loop: behind the
    // reduction every one of them is a dependent load on a launch whose whole length is the dependency chain's (round 5, box N)
    const int col = tid & 31, rg = tid >> 5;
    const int n = ntile * 32 + col;
    const bool vn = n < p.Cout;
    const bool bnb = p.bnb_scale != nullptr;
    float bias = 0.f, bsc = 0.f, bsh = 0.f, bmu = 0.f, bis = 0.f;
    if (!PRO) {
        bias = (p.bias && vn) ? p.bias[n] : 0.f;
        if (bnb && vn) { bsc = p.bnb_scale[n]; bsh = p.bnb_shift[n]; bmu = p.bnb_mean[n]; bis = p.bnb_invstd[n]; }
    }
    // (not in the variant with a BatchNorm prologue — the forward launches: it sits at 128 registers, and ten more are three waves per
    // SIMD instead of four, one 512-thread workgroup per CU instead of two: hg2 +0.07 ms.  The data-gradient launches — mask operand and
    // four BatchNorm-backward vectors in the epilogue — are the ones without a prologue)
    float r1v[2] = {0.f, 0.f};
    if (!PRO) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int mo = mtile * 32 + rg + 16 * j;
            if (vn && mo < p.M && p.res1) r1v[j] = p.res1[(size_t)mo * p.Cout + n];
        }
    }
    if (c0 < c1) {
        // two chunks in flight beside the one being multiplied; three register sets in rotation (no copies)
        Frag f0 = load_next(), f1 = load_next(), f2;
        for (int ch = c0;;) {
            f2 = load_next(); mma(f0); if (++ch >= c1) break;
            f0 = load_next(); mma(f1); if (++ch >= c1) break;
            f1 = load_next(); mma(f2); if (++ch >= c1) break;
        }
    }
    if (PRO) __syncthreads();                       // every wave is done with the vectors in `part`
    // partial tiles -> LDS (C/D layout: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5))
#pragma unroll
    for (int e = 0; e < 16; ++e) part[wave][(e & 3) + 8 * (e >> 2) + 4 * h][i] = acc[e];
    __syncthreads();
    if (PRO) {
        bias = (p.bias && vn) ? p.bias[n] : 0.f;
        if (bnb && vn) { bsc = p.bnb_scale[n]; bsh = p.bnb_shift[n]; bmu = p.bnb_mean[n]; bis = p.bnb_invstd[n]; }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = rg + 16 * j;
        const int mo = mtile * 32 + row;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += part[w][row][col];
        if (vn && mo < p.M) {
            const size_t o = (size_t)mo * p.Cout + n;
            if (bnb) {
                const float xv = PRO ? p.res1[o] : r1v[j];
                if (p.bnb_relu && fmaf(xv, bsc, bsh) <= 0.f) v = 0.f;
                p.y[o] = v;
                s1 += v;
                s2 = fmaf(v, (xv - bmu) * bis, s2);
            } else {
                v += bias + (PRO ? (p.res1 ? p.res1[o] : 0.f) : r1v[j]) + (p.res2 ? p.res2[o] : 0.f);
                p.y[o] = v;
                s1 += v;
                s2 = fmaf(v, v, s2);
            }
        }
    }
    if (p.stats) {
        __syncthreads();                             // every thread has read its part of `part`
        float* red = &part[0][0][0];                 // [16][32][2]
        red[(rg * 32 + col) * 2 + 0] = s1;
        red[(rg * 32 + col) * 2 + 1] = s2;
        __syncthreads();
        if (tid < 32 && vn) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) { a0 += red[(w * 32 + tid) * 2 + 0]; a1 += red[(w * 32 + tid) * 2 + 1]; }
            tail_store(p.stats + ((size_t)mtile * 2 + 0) * p.Cout + n, a0);
            tail_store(p.stats + ((size_t)mtile * 2 + 1) * p.Cout + n, a1);
        }
    }
}
template <bool PRO, int CW>
__global__ __launch_bounds__(512) void conv_ksplit_kernel(ConvP p) {
    __shared__ __attribute__((aligned(16))) float part[8][32][33];
    conv_ksplit_body<PRO, CW>(p, blockIdx.x, part, true);
}

// rows up to which the K-split kernel replaces the 32 x 128 tiling (measured crossover: 2048)
static long ksplit_rows() {
    static long v = -1;
    if (v < 0) {
        const char* e = getenv("DSNT_X_KSPLIT_ROWS");        // A/B only (tools/ab_env.sh)
        v = e ? atol(e) : 2048;
    }
    return v;
}

// fp32 path only, <= 256 input channels, <= 128 KB of partial sums: what every workgroup re-reads in its prologue
static bool conv_fwd_pro_ok(const dsnt_conv_geom* g, int tiles, int C) {
    return g && C == g->Cin && C <= 256 && C % 4 == 0 && tiles > 0 && (long)tiles * C <= 16384;
}
extern "C" int dsnt_conv_fwd_pro_ok(const dsnt_conv_geom* g, int tiles, int C) { return conv_fwd_pro_ok(g, tiles, C) ? 1 : 0; }

static int conv_fwd_impl(const float* x, const float* w, const float* bias, float* y,
                         const float* in_scale, const float* in_shift, int in_relu,
                         const float* res1, const float* res2, float* stats_partial,
                         const dsnt_conv_geom* g, const dsnt_bn_bwd_epilogue* g_bnb, const dsnt_out_bounds* g_tail,
                         void* stream, const dsnt_bn_prologue* g_pro = nullptr) {
    if (int e = check_geom(g, "dsnt_conv_fwd")) return e;
    DSNT_REQUIRE(!g_bnb || (g_bnb->x && g_bnb->scale && g_bnb->shift && g_bnb->mean && g_bnb->invstd &&
                            stats_partial && !res1 && !res2 && !bias), DSNT_ERR_ARG,
                 "dsnt_conv_fwd_ex: the batch-norm-backward epilogue needs x/scale/shift/mean/invstd and "
                 "stats_partial, and excludes bias/residuals");
    DSNT_REQUIRE(x && w && y, DSNT_ERR_ARG, "dsnt_conv_fwd: null tensor");
    DSNT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), DSNT_ERR_ARG,
                 "dsnt_conv_fwd: in_scale/in_shift must be given together");
    DSNT_REQUIRE(dsnt_aligned16(x) && dsnt_aligned16(w) && (!in_scale || dsnt_aligned16(in_scale)) &&
                 (!in_shift || dsnt_aligned16(in_shift)), DSNT_ERR_ALIGN,
                 "dsnt_conv_fwd: x/w/scale/shift must be 16-byte aligned");
    ConvP p;
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.in_scale = in_scale; p.in_shift = in_shift;
    p.res1 = res1; p.res2 = res2; p.stats = stats_partial; p.in_relu = in_relu; p.wq = nullptr; p.wq_stride = 0;
    p.bnb_scale = p.bnb_shift = p.bnb_mean = p.bnb_invstd = nullptr; p.bnb_relu = 0;
    p.a_bound = p.w_bound = nullptr;
    p.ap_y = p.ap_scale = p.ap_mean = p.ap_invstd = p.ap_coef = nullptr; p.ap_out = nullptr;
    if (g_bnb) {
        p.res1 = g_bnb->x; p.bnb_scale = g_bnb->scale; p.bnb_shift = g_bnb->shift;
        p.bnb_mean = g_bnb->mean; p.bnb_invstd = g_bnb->invstd; p.bnb_relu = g_bnb->relu;
    }
    if (int e = out_bounds_fill(p.tail, g_tail, "dsnt_conv_fwd_ex")) return e;
    memset(&p.pro, 0, sizeof(p.pro));
    if (g_pro) {
        DSNT_REQUIRE(g_pro->partial && g_pro->mean && g_pro->invstd && g_pro->scale && g_pro->shift && g_pro->M > 0 &&
                     conv_fwd_pro_ok(g, g_pro->tiles, g_pro->C), DSNT_ERR_ARG,
                     "dsnt_conv_fwd_pro: incomplete dsnt_bn_prologue, or more than 256 channels / 128 KB of partial sums");
        DSNT_REQUIRE((g_pro->running_mean == nullptr) == (g_pro->running_var == nullptr), DSNT_ERR_ARG,
                     "dsnt_conv_fwd_pro: running_mean/var must be given together");
        DSNT_REQUIRE(dsnt_aligned16(g_pro->scale) && dsnt_aligned16(g_pro->shift), DSNT_ERR_ALIGN, "dsnt_conv_fwd_pro: alignment");
        p.pro.partial = g_pro->partial; p.pro.tiles = g_pro->tiles; p.pro.C = g_pro->C;
        p.pro.invM = 1.0 / (double)g_pro->M;
        p.pro.unbias = g_pro->M > 1 ? (double)g_pro->M / (double)(g_pro->M - 1) : 1.0;
        p.pro.gamma = g_pro->gamma; p.pro.beta = g_pro->beta; p.pro.rmean = g_pro->running_mean; p.pro.rvar = g_pro->running_var;
        p.pro.momentum = g_pro->momentum; p.pro.eps = g_pro->eps;
        p.pro.mean = g_pro->mean; p.pro.invstd = g_pro->invstd; p.pro.scale = g_pro->scale; p.pro.shift = g_pro->shift;
    }
    p.N = g->N; p.H = g->H; p.W = g->W; p.Cin = g->Cin; p.Ho = g->Ho; p.Wo = g->Wo;
    p.Cout = g->Cout; p.R = g->R; p.S = g->S; p.stride = g->stride; p.pad = g->pad; p.dil = g->dil;
    p.M = g->N * g->Ho * g->Wo; p.K = g->R * g->S * g->Cin;
    int BM, BN;
    pick_cfg(g, BM, BN);
    p.mtiles = (p.M + BM - 1) / BM; p.ntiles = (p.Cout + BN - 1) / BN;
    hipStream_t st = (hipStream_t)stream;
    const bool pro = in_scale != nullptr;
    DSNT_REQUIRE(!(p.tail.amax_bn && g_bnb), DSNT_ERR_ARG, "dsnt_conv_fwd_ex: dsnt_out_bounds.amax_bn excludes the batch-norm-backward epilogue");
    if (BM == 32 && p.M <= ksplit_rows() && p.Cin % 8 == 0 && (size_t)p.N * p.H * p.W * p.Cin * 4u < (1ull << 31) &&
        (size_t)p.Cout * p.K * 4u < (1ull << 31) && !p.tail.amax && !p.tail.amax_bn &&        // (the K-split epilogue has no amax)
        (!pro || p.Cin <= 4096)) {
        const int grid = p.mtiles * ((p.Cout + 31) / 32);
        // (KS_CW: experiments only)
#ifndef KS_CW
#define KS_CW 16
#endif
        if (p.Cin % KS_CW == 0) {
            // (these two can join a persistent stage: stage.h)
            if (pro) DSNT_LAUNCH_OP(KS_CW == 16 ? DSNT_ST_KSPLIT_PRO : DSNT_ST_NONE, (conv_ksplit_kernel<true, KS_CW>), dim3(grid), dim3(512), 0, st, p);
            else DSNT_LAUNCH_OP(KS_CW == 16 ? DSNT_ST_KSPLIT : DSNT_ST_NONE, (conv_ksplit_kernel<false, KS_CW>), dim3(grid), dim3(512), 0, st, p);
        } else if (pro) DSNT_LAUNCH((conv_ksplit_kernel<true, 8>), dim3(grid), dim3(512), 0, st, p);
        else DSNT_LAUNCH((conv_ksplit_kernel<false, 8>), dim3(grid), dim3(512), 0, st, p);
    } else if (BM == 128 && BN == 128) launch_fwd<2, 2, 2, 2>(p, pro, st);
    else if (BM == 128 && BN == 64) launch_fwd<2, 2, 2, 1>(p, pro, st);
    else if (BM == 128 && BN == 32) launch_fwd<4, 1, 1, 1>(p, pro, st);
    else launch_fwd<1, 4, 1, 1>(p, pro, st);
    DSNT_CHECK_LAUNCH("dsnt_conv_fwd");
}

extern "C" int dsnt_conv_fwd(const float* x, const float* w, const float* bias, float* y,
                             const float* in_scale, const float* in_shift, int in_relu,
                             const float* res1, const float* res2, float* stats_partial,
                             const dsnt_conv_geom* g, void* stream) {
    return conv_fwd_impl(x, w, bias, y, in_scale, in_shift, in_relu, res1, res2, stats_partial, g, nullptr, nullptr, stream);
}

extern "C" int dsnt_conv_fwd_ex(const float* x, const float* w, const float* bias, float* y,
                                const float* in_scale, const float* in_shift, int in_relu,
                                const float* res1, const float* res2, float* stats_partial,
                                const dsnt_conv_geom* g, const dsnt_bn_bwd_epilogue* bnb, const dsnt_out_bounds* tail,
                                void* stream) {
    return conv_fwd_impl(x, w, bias, y, in_scale, in_shift, in_relu, res1, res2, stats_partial, g, bnb, tail, stream);
}

extern "C" int dsnt_conv_fwd_pro(const float* x, const float* w, const float* bias, float* y, const dsnt_bn_prologue* pro,
                                 int in_relu, const float* res1, const float* res2, float* stats_partial,
                                 const dsnt_conv_geom* g, const dsnt_out_bounds* tail, void* stream) {
    DSNT_REQUIRE(pro, DSNT_ERR_ARG, "dsnt_conv_fwd_pro: null dsnt_bn_prologue");
    return conv_fwd_impl(x, w, bias, y, pro->scale, pro->shift, in_relu, res1, res2, stats_partial, g, nullptr, tail, stream, pro);
}


// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution on bf16x6 with an LDS halo tile.
//
// Why: VALU issue on a SIMD is arbitrated by priority, then age, and a wave issuing MFMAs back to back
// keeps winning: the loader waves' VALU work only runs in the gaps (tools/starve.py: a 128-instruction
// VALU burst next to a saturated matrix pipe takes the whole MFMA phase to finish).  In the implicit-GEMM
// kernel above every filter tap re-loads, re-normalises and re-splits the same input pixels (9x the
// VALU work, 2.4x the HBM traffic of the tensor).  Here a workgroup owns an 8 x 16 patch of output
// pixels: the (8+2) x (16+2) input halo of 16 channels is transformed and split ONCE into LDS, then all
// nine taps run from it with shifted fragment addresses (immediate offsets), while only the pre-split
// weights stream through the double-buffered B tile (a 16-byte copy, no VALU).
// (A variant without the loader/MFMA role split — 256 threads, fragments of step s+1 read during the
// MFMAs of step s — reached 84 % matrix-pipe use inside the K loop but was 8 % slower end to end: with
// 72 K-steps per tile the ~10k-cycle prologue and ~12k-cycle epilogue of two lock-stepped workgroups
// per CU dominate either way.)
//   LDS: A halo [3 planes][192 px][24] bf16 (27.6 KB, single buffer: refilled at chunk boundaries
//   from registers that were loaded nine steps earlier) + B [2][3][BN][24] bf16 (36.9 KB).
// K order is (16-channel chunk, tap) instead of (tap, channel): same products, different fp32
// summation order than the implicit-GEMM kernel (differences at the 1e-7 level).
template <int TN, bool PRO, bool F16 = false>
__global__ __launch_bounds__(512, 2) void conv3x3_bf16x6_kernel(ConvP p) {
    constexpr int NPL = F16 ? 2 : 3;            // operand planes (fp16x3: two fp16 planes, three MFMAs)
    // fp16x3: the two-plane halo is small enough to be DOUBLE-buffered (2 x 18 KB + 24 KB of weights < the 66 KB the
    // epilogue's C tile needs anyway): the next chunk's halo is staged while this chunk's taps run, instead of in an
    // extra barrier-bracketed stage between chunks (which, with half the MFMAs per tap, had become 10 % of the kernel)
    constexpr int ABUF = F16 ? 2 : 1;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int WN = 2, TM = 2, BM = 128, BN = WN * TN * 32;
    constexpr int HWD = 18, HPP = 192;                 // halo row width; halo pixels (180) padded to 192
    constexpr int BROWS = BN * 2 / 256 >= 1 ? 3 : 3;   // three planes per loader thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* A6 = reinterpret_cast<__bf16*>(smem);          // [ABUF][NPL][HPP][PITCH6]
    __bf16* B6 = A6 + ABUF * NPL * HPP * PITCH6;           // [2][NPL][BN][PITCH6]

    int tile;
    xcd_remap(blockIdx.x, p.mtiles * p.ntiles, tile);
    const int ntile = tile % p.ntiles, mtile = tile / p.ntiles;
    const int tws = p.W >> 4, ths = p.H >> 3;
    const int tw = mtile % tws, th = (mtile / tws) % ths, img = mtile / (tws * ths);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = p.Cin >> 4;                    // even (Cin % 32 == 0)
    const int lr = lane & 31, lh = lane >> 5;
    const int cw = wave & 3;
    const int wm = cw >> 1, wn = cw & 1;
    (void)BROWS;
    DBG_INIT();
    DBG_STAMP(0);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    if (wave >= 4) {
        // ------------------------------------------------------------------ loader waves
        const int ltid = tid - 256;
        const int kc = ltid & 3;
        const unsigned OOB = 0xF0000000u;
        unsigned aoffs[3], alds[3], aok = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pix = (ltid >> 2) + 64 * i;      // 0..191, halo pixels 0..179
            const int hy = pix / HWD, hx = pix - hy * HWD;
            const int ih = th * 8 - 1 + hy, iw = tw * 16 - 1 + hx;
            const bool in = pix < 180 && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            aoffs[i] = in ? (unsigned)(((img * p.H + ih) * p.W + iw) * p.Cin + kc * 4) * 4u : OOB;
            alds[i] = (unsigned)(pix * PITCH6 + kc * 4);
            aok |= (in ? 1u : 0u) << i;
        }
        const int bhalf = ltid & 1;
        const int brow = ((ltid >> 4) << 3) + (((ltid >> 1) & 3) << 1) + ((ltid >> 3) & 1);
        const int bn = ntile * BN + brow;
        const bool bvalid = brow < BN && bn < p.Cout;
        unsigned bpix[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j)
            bpix[j] = bvalid ? (unsigned)((size_t)j * p.wq_stride + (size_t)bn * p.K + bhalf * 8) * 2u : OOB;
        const unsigned blds = (unsigned)(brow * PITCH6 + bhalf * 8);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * p.Cin * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned short*>(p.wq), 0, (int)(((size_t)(NPL - 1) * p.wq_stride + (size_t)p.Cout * p.K) * 2u), 0x00020000);
        const float sa = F16 ? pow2_scale(bound64(p.a_bound)) : 1.f;      // fp16x3 operand scale
        u32x4 ra[3], rb[2][NPL];
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        const int lastc = nchunks - 1;
        const float lo_valid = p.in_relu ? 0.f : -__builtin_inff();
        auto gloadA = [&](int c) {
            c = min(c, lastc);
            if (PRO) {
                sc = *reinterpret_cast<const float4*>(p.in_scale + c * 16 + kc * 4);
                sh = *reinterpret_cast<const float4*>(p.in_shift + c * 16 + kc * 4);
                if (F16) {      // the operand scale rides in the BN vectors
                    sc.x *= sa; sc.y *= sa; sc.z *= sa; sc.w *= sa;
                    sh.x *= sa; sh.y *= sa; sh.z *= sa; sh.w *= sa;
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, aoffs[i], c * 64, 0);
        };
        auto storeA = [&](int abuf) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float4 v = make_float4(__uint_as_float(ra[i].x), __uint_as_float(ra[i].y),
                                       __uint_as_float(ra[i].z), __uint_as_float(ra[i].w));
                if (PRO) {
                    // BN FMAs; ReLU + zero padding (applied after BN + ReLU) as one median per element
                    const sp_f32x2 a = {fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y)};
                    const sp_f32x2 b = {fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w)};
                    const bool ok = (aok >> i) & 1u;
                    const float lo = ok ? lo_valid : 0.f, hi = ok ? __builtin_inff() : 0.f;
                    v.x = __builtin_amdgcn_fmed3f(a.x, lo, hi); v.y = __builtin_amdgcn_fmed3f(a.y, lo, hi);
                    v.z = __builtin_amdgcn_fmed3f(b.x, lo, hi); v.w = __builtin_amdgcn_fmed3f(b.y, lo, hi);
                }
                uint2 q1, q2, q3;
                __bf16* dst = A6 + abuf * NPL * HPP * PITCH6 + alds[i];
                if (F16) {
                    if (!PRO) { v.x *= sa; v.y *= sa; v.z *= sa; v.w *= sa; }
                    split4h(v, q1, q2);
                    *reinterpret_cast<uint2*>(dst) = q1;
                    *reinterpret_cast<uint2*>(dst + HPP * PITCH6) = q2;
                } else {
                    split4(v, q1, q2, q3);
                    *reinterpret_cast<uint2*>(dst) = q1;
                    *reinterpret_cast<uint2*>(dst + HPP * PITCH6) = q2;
                    *reinterpret_cast<uint2*>(dst + 2 * HPP * PITCH6) = q3;
                }
            }
        };
        // weights of K-step (chunk c, tap t): OHWI columns t*Cin + c*16 .. +16 of the three planes
        auto gloadB = [&](int stage, int c, int t) {
            c = min(c, lastc);
            const unsigned koff = (unsigned)(t * p.Cin + c * 16) * 2u;
#pragma unroll
            for (int j = 0; j < NPL; ++j) rb[stage][j] = __builtin_amdgcn_raw_buffer_load_b128(wr, bpix[j], koff, 0);
        };
        auto storeB = [&](int stage, int buf) {
            if (brow < BN) {
#pragma unroll
                for (int j = 0; j < NPL; ++j)
                    *reinterpret_cast<u32x4*>(B6 + (size_t)(buf * NPL + j) * BN * PITCH6 + blds) = rb[stage][j];
            }
        };
        // step s = 9 * chunk + tap reads B buffer s & 1; its weights sit in register stage s & 1
        gloadA(0);
        gloadB(0, 0, 0);
        gloadB(1, 0, 1);
        storeA(0);
        storeB(0, 0);
        gloadA(1);
        gloadB(0, 0, 2);
        __syncthreads();
        for (int c2 = 0; c2 < nchunks; c2 += 2) {
#pragma unroll
            for (int j = 0; j < 18; ++j) {
                if (c2 == 0) DBG_STAMP(1 + 3 * j);                 // timeline builds (tools/timeline_halo.py): step start
                // while the MFMA waves work on step j: stage step j+1, fetch step j+3
                storeB((j + 1) & 1, (j + 1) & 1);
                gloadB((j + 1) & 1, c2 + (j + 3) / 9, (j + 3) % 9);
                if (ABUF == 2) {
                    if (j % 9 == 0) {                // the other halo buffer is free since the last barrier
                        storeA(1 - (j / 9));
                        gloadA(c2 + j / 9 + 2);
                    }
                } else if (j % 9 == 8) {
                    __syncthreads();                 // the MFMA waves hold the last fragments of this chunk
                    storeA(0);
                    gloadA(c2 + j / 9 + 2);
                }
                if (c2 == 0) DBG_STAMP(2 + 3 * j);                 // staged, loads issued
                __syncthreads();
                if (c2 == 0) DBG_STAMP(3 + 3 * j);                 // through the barrier
            }
        }
    } else {
        // ------------------------------------------------------------------ MFMA waves
        struct Frag { bf16x8 a[TM][3], b[TN][3]; };
        Frag F;
        int aoff[TM], boff[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
            // lanes 16..31 sit one halo row (18 pixels) further: rotating their pixel column by two restores
            // the 16-pixel period of the conflict-free ds_read_b128 pattern (26 % -> ~0 % bank conflicts)
            aoff[a] = (((wm * TM + a) * 2 + (lr >> 4)) * HWD + ((lr + (lr >> 4) * 14) & 15)) * PITCH6 + 8 * lh;
#pragma unroll
        for (int b = 0; b < TN; ++b) boff[b] = ((wn * TN + b) * 32 + lr) * PITCH6 + 8 * lh;
        __syncthreads();
        for (int c2 = 0; c2 < nchunks; c2 += 2) {
#pragma unroll
            for (int j = 0; j < 18; ++j) {
                if (c2 == 0) DBG_STAMP(1 + 3 * j);
                const int t = j % 9, buf = j & 1;
                const int toff = ((t / 3) * HWD + (t % 3)) * PITCH6 + (ABUF == 2 ? (j / 9) * NPL * HPP * PITCH6 : 0);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                    for (int a = 0; a < TM; ++a)
                        F.a[a][pl] = *reinterpret_cast<const bf16x8*>(A6 + pl * HPP * PITCH6 + aoff[a] + toff);
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        F.b[b][pl] = *reinterpret_cast<const bf16x8*>(B6 + (buf * NPL + pl) * BN * PITCH6 + boff[b]);
                }
                if (ABUF == 1 && t == 8) __syncthreads();   // fragments are in registers: the halo may be refilled
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) mma_split<F16>(acc[a][b], F.a[a], F.b[b]);
                if (c2 == 0) DBG_STAMP(2 + 3 * j);                 // MFMAs issued
                __syncthreads();
                if (c2 == 0) DBG_STAMP(3 + 3 * j);                 // through the barrier
            }
        }
    }
    const int mbase = (img * p.H + th * 8) * p.W + tw * 16;
    DBG_STAMP(124);
    conv_epilogue<2, WN, TM, TN, true>(p, acc, smem, mtile, ntile, tid, wave, lane, mbase);
    DBG_STAMP(125);
}

template <int TN, bool F16 = false>
static void launch_conv3x3_6(const ConvP& p, bool pro, hipStream_t st) {
    constexpr int BN = 64 * TN, NPL = F16 ? 2 : 3;
    size_t lds = (size_t)((F16 ? 2 : 1) * NPL * 192 + 2 * NPL * BN) * PITCH6 * 2;
    const size_t epi = (size_t)128 * (BN + 4) * 4;
    if (epi > lds) lds = epi;
    if (lds > 65536) {
        DSNT_SET_MAX_LDS((conv3x3_bf16x6_kernel<TN, true, F16>), lds);
        DSNT_SET_MAX_LDS((conv3x3_bf16x6_kernel<TN, false, F16>), lds);
    }
    dim3 gr(p.mtiles * p.ntiles), bl(512);
    if (pro) DSNT_LAUNCH((conv3x3_bf16x6_kernel<TN, true, F16>), gr, bl, lds, st, p);
    else DSNT_LAUNCH((conv3x3_bf16x6_kernel<TN, false, F16>), gr, bl, lds, st, p);
}

static bool conv3x3_halo_ok(const dsnt_conv_geom* g) {
    return g->R == 3 && g->S == 3 && g->stride == 1 && g->pad == 1 && g->dil == 1 && g->Ho == g->H && g->Wo == g->W &&
           g->H % 8 == 0 && g->W % 16 == 0 && g->Cin % 32 == 0;
}

template <int WM, int WN, int TM, int TN, bool F16 = false>
static void launch_fwd6(const ConvP& p, bool pro, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    size_t lds = (size_t)2 * (F16 ? 2 : 3) * (BM + BN) * PITCH6 * 2 + (size_t)2 * p.Cin * sizeof(float);   // tiles + BN vectors
    const size_t epi = (size_t)BM * (BN + 4) * 4;          // the epilogue's C tile lives in the same LDS
    if (epi > lds) lds = epi;
    if (lds > 65536) {
        DSNT_SET_MAX_LDS((conv_fwd_bf16x6_kernel<WM, WN, TM, TN, true, F16, 2>), lds);
        DSNT_SET_MAX_LDS((conv_fwd_bf16x6_kernel<WM, WN, TM, TN, false, F16, 2>), lds);
    }
    // two A-operand register stages (DA): four measured 2-5 % slower on every 1x1 shape of the hourglass (round 2)
    dim3 gr(p.mtiles * p.ntiles), bl(512);
    if (pro) DSNT_LAUNCH((conv_fwd_bf16x6_kernel<WM, WN, TM, TN, true, F16, 2>), gr, bl, lds, st, p);
    else DSNT_LAUNCH((conv_fwd_bf16x6_kernel<WM, WN, TM, TN, false, F16, 2>), gr, bl, lds, st, p);
}

#ifndef FWD6_BN64_ROWS_DEFAULT
#define FWD6_BN64_ROWS_DEFAULT 8192
#endif
static bool g_force_gemm6 = false;      // debug/bench: route 3x3 convolutions through the implicit-GEMM kernel
extern "C" int dsnt_debug_force_gemm6(int on) { g_force_gemm6 = on != 0; return DSNT_OK; }

static int conv_fwd6_impl(const float* x, const void* w_planes, int64_t plane_stride, const float* bias, float* y,
                          const float* in_scale, const float* in_shift, int in_relu,
                          const float* res1, const float* res2, float* stats_partial,
                          const dsnt_conv_geom* g, const dsnt_bn_bwd_epilogue* g_bnb, const dsnt_out_bounds* g_tail,
                          void* stream, const float* a_bound = nullptr, const float* w_bound = nullptr,
                          bool stream_w = false, const dsnt_bn_bwd_apply* g_ap = nullptr, float* ap_out = nullptr) {
    if (int e = check_geom(g, "dsnt_conv_fwd_bf16x6")) return e;
    DSNT_REQUIRE(!g_bnb || (g_bnb->x && g_bnb->scale && g_bnb->shift && g_bnb->mean && g_bnb->invstd &&
                            stats_partial && !res1 && !res2 && !bias), DSNT_ERR_ARG,
                 "dsnt_conv_fwd_bf16x6_ex: the batch-norm-backward epilogue needs x/scale/shift/mean/invstd "
                 "and stats_partial, and excludes bias/residuals");
    DSNT_REQUIRE(x && w_planes && y, DSNT_ERR_ARG, "dsnt_conv_fwd_bf16x6: null tensor");
    DSNT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), DSNT_ERR_ARG,
                 "dsnt_conv_fwd_bf16x6: in_scale/in_shift must be given together");
    DSNT_REQUIRE(dsnt_conv_bf16x6_ok(g), DSNT_ERR_SHAPE,
                 "dsnt_conv_fwd_bf16x6: geometry not supported (need Cin %% 16 == 0, <= 16 taps, < 2 GiB)");
    DSNT_REQUIRE(dsnt_aligned16(x) && dsnt_aligned16(w_planes) && (!in_scale || dsnt_aligned16(in_scale)) &&
                 (!in_shift || dsnt_aligned16(in_shift)), DSNT_ERR_ALIGN, "dsnt_conv_fwd_bf16x6: alignment");
    ConvP p;
    DSNT_REQUIRE(plane_stride >= (int64_t)g->Cout * g->R * g->S * g->Cin && plane_stride % 8 == 0 &&
                 (2 * plane_stride + (int64_t)g->Cout * g->R * g->S * g->Cin) * 2 < (1LL << 31), DSNT_ERR_SHAPE,
                 "dsnt_conv_fwd_bf16x6: bad plane stride %lld", (long long)plane_stride);
    memset(&p.pro, 0, sizeof(p.pro));
    p.x = x; p.w = nullptr; p.wq = (const unsigned short*)w_planes; p.wq_stride = plane_stride; p.bias = bias; p.y = y;
    p.in_scale = in_scale; p.in_shift = in_shift; p.res1 = res1; p.res2 = res2; p.stats = stats_partial;
    p.in_relu = in_relu;
    p.bnb_scale = p.bnb_shift = p.bnb_mean = p.bnb_invstd = nullptr; p.bnb_relu = 0;
    p.a_bound = p.w_bound = nullptr;
    p.ap_y = p.ap_scale = p.ap_mean = p.ap_invstd = p.ap_coef = nullptr; p.ap_out = nullptr;
    if (g_bnb) {
        p.res1 = g_bnb->x; p.bnb_scale = g_bnb->scale; p.bnb_shift = g_bnb->shift;
        p.bnb_mean = g_bnb->mean; p.bnb_invstd = g_bnb->invstd; p.bnb_relu = g_bnb->relu;
    }
    if (int e = out_bounds_fill(p.tail, g_tail, "dsnt_conv_fwd_bf16x6_ex")) return e;
    p.N = g->N; p.H = g->H; p.W = g->W; p.Cin = g->Cin; p.Ho = g->Ho; p.Wo = g->Wo;
    p.Cout = g->Cout; p.R = g->R; p.S = g->S; p.stride = g->stride; p.pad = g->pad; p.dil = g->dil;
    p.M = g->N * g->Ho * g->Wo; p.K = g->R * g->S * g->Cin;
    // 64-column tiles also for wider outputs when there are few row tiles (the 16 x 16 level and below: 32-64 workgroups of 128 x 128
    // on 256 CUs, each a prologue + 8-16 K-steps + an epilogue through LDS): twice the workgroups, half the epilogue each.
    // DSNT_X_FWD6_BN64_ROWS: A/B
    static long bn64_rows = -1;
    if (bn64_rows < 0) { const char* e = getenv("DSNT_X_FWD6_BN64_ROWS"); bn64_rows = e ? atol(e) : FWD6_BN64_ROWS_DEFAULT; }
    const int BN = (g->Cout <= 64 || (p.M <= bn64_rows && g->Cout % 64 == 0 && !conv3x3_halo_ok(g))) ? 64 : 128;
    p.mtiles = (p.M + 127) / 128; p.ntiles = (p.Cout + BN - 1) / BN;
    hipStream_t st = (hipStream_t)stream;
    DSNT_REQUIRE(!(p.tail.amax_bn && g_bnb), DSNT_ERR_ARG, "dsnt_conv_fwd_bf16x6_ex: dsnt_out_bounds.amax_bn excludes the batch-norm-backward epilogue");
    p.a_bound = a_bound; p.w_bound = w_bound;
    const bool share_chip = a_bound && (in_relu & DSNT_CONV_SHARE_CHIP) != 0;       // (fp16x3 entry points) leave room beside this launch
    if (a_bound) p.in_relu = in_relu & 1;
    if (g_ap) {
        p.ap_y = g_ap->y; p.ap_scale = g_ap->scale; p.ap_mean = g_ap->mean; p.ap_invstd = g_ap->invstd; p.ap_coef = g_ap->coef;
        p.ap_out = ap_out;
    }
    if (stream_w) {                  // 3x3, weights in the stream layout: the symmetric kernel (conv3s.hip)
        DSNT_REQUIRE(dsnt_conv3s_ok(p), DSNT_ERR_SHAPE, "dsnt_conv_fwd_f16x3_stream: launch not supported (dsnt_conv_fwd_stream_ok; "
                     "no second residual)");
        dsnt_conv3s_launch(p, in_scale != nullptr, st, share_chip);
        DSNT_CHECK_LAUNCH("dsnt_conv_fwd_f16x3_stream");
    }
    if (a_bound) {                   // fp16x3: two fp16 weight planes, operand bounds in device memory
        const int ntw = dsnt_gemm1_cfg(p);      // large 1x1 convolutions: the streaming kernel (gemm1.hip)
        if (ntw > 0) {
            dsnt_gemm1_launch(p, ntw, in_scale != nullptr, st, share_chip);
            DSNT_CHECK_LAUNCH("dsnt_conv_fwd_f16x3");
        }
        if (conv3x3_halo_ok(g) && !g_force_gemm6) {
            if (BN == 128) launch_conv3x3_6<2, true>(p, in_scale != nullptr, st);
            else launch_conv3x3_6<1, true>(p, in_scale != nullptr, st);
        } else if (BN == 128) launch_fwd6<2, 2, 2, 2, true>(p, in_scale != nullptr, st);
        else launch_fwd6<2, 2, 2, 1, true>(p, in_scale != nullptr, st);
    } else if (conv3x3_halo_ok(g) && !g_force_gemm6) {
        if (BN == 128) launch_conv3x3_6<2>(p, in_scale != nullptr, st);
        else launch_conv3x3_6<1>(p, in_scale != nullptr, st);
    } else if (BN == 128) launch_fwd6<2, 2, 2, 2>(p, in_scale != nullptr, st);
    else launch_fwd6<2, 2, 2, 1>(p, in_scale != nullptr, st);
    DSNT_CHECK_LAUNCH("dsnt_conv_fwd_bf16x6");
}

extern "C" int dsnt_conv_fwd_bf16x6(const float* x, const void* w_planes, int64_t plane_stride, const float* bias, float* y,
                                    const float* in_scale, const float* in_shift, int in_relu,
                                    const float* res1, const float* res2, float* stats_partial,
                                    const dsnt_conv_geom* g, void* stream) {
    return conv_fwd6_impl(x, w_planes, plane_stride, bias, y, in_scale, in_shift, in_relu, res1, res2,
                          stats_partial, g, nullptr, nullptr, stream);
}

extern "C" int dsnt_conv_fwd_bf16x6_ex(const float* x, const void* w_planes, int64_t plane_stride, const float* bias,
                                       float* y, const float* in_scale, const float* in_shift, int in_relu,
                                       const float* res1, const float* res2, float* stats_partial,
                                       const dsnt_conv_geom* g, const dsnt_bn_bwd_epilogue* bnb, const dsnt_out_bounds* tail,
                                       void* stream) {
    return conv_fwd6_impl(x, w_planes, plane_stride, bias, y, in_scale, in_shift, in_relu, res1, res2,
                          stats_partial, g, bnb, tail, stream);
}

extern "C" int dsnt_conv_fwd_f16x3_ex(const float* x, const void* w_planes, int64_t plane_stride, const float* w_bound,
                                      const float* a_bound, const float* bias, float* y, const float* in_scale,
                                      const float* in_shift, int in_relu, const float* res1, const float* res2,
                                      float* stats_partial, const dsnt_conv_geom* g, const dsnt_bn_bwd_epilogue* bnb,
                                      const dsnt_out_bounds* tail, void* stream) {
    DSNT_REQUIRE(a_bound && w_bound, DSNT_ERR_ARG, "dsnt_conv_fwd_f16x3_ex: the operand bounds (device scalars) are required");
    return conv_fwd6_impl(x, w_planes, plane_stride, bias, y, in_scale, in_shift, in_relu, res1, res2,
                          stats_partial, g, bnb, tail, stream, a_bound, w_bound);
}

// The same call with the weight planes in the STREAM layout of dsnt_f16_prep_weights (3x3 convolutions the symmetric
// kernel of conv3s.hip runs: dsnt_conv_fwd_stream_ok)
extern "C" int dsnt_conv_fwd_f16x3_stream(const float* x, const void* w_planes, int64_t plane_stride, const float* w_bound,
                                          const float* a_bound, const float* bias, float* y, const float* in_scale,
                                          const float* in_shift, int in_relu, const float* res1, const float* res2,
                                          float* stats_partial, const dsnt_conv_geom* g, const dsnt_bn_bwd_epilogue* bnb,
                                          const dsnt_out_bounds* tail, void* stream) {
    DSNT_REQUIRE(a_bound && w_bound, DSNT_ERR_ARG, "dsnt_conv_fwd_f16x3_stream: the operand bounds (device scalars) are required");
    return conv_fwd6_impl(x, w_planes, plane_stride, bias, y, in_scale, in_shift, in_relu, res1, res2,
                          stats_partial, g, bnb, tail, stream, a_bound, w_bound, true);
}
extern "C" int dsnt_conv_fwd_stream_ok(const dsnt_conv_geom* g) { return dsnt_conv3s_geom_ok(g) ? 1 : 0; }
extern "C" int dsnt_conv_fwd_stream_form(const dsnt_conv_geom* g, int mode) { return dsnt_conv3s_form_of(g, mode); }

// The data gradient of a 3x3 convolution whose OUTPUT feeds a train-mode BatchNorm, with that BatchNorm's backward folded into
// the operand load (conv3s.hip MODE 4): instead of dL/dy the launch reads dz (the ReLU-masked, reduced gradient behind the
// BatchNorm) and the BatchNorm's input ap->y and forms  dy = scale (dz - c0 - (y - mean) invstd c1)  on the fly; it also writes
// dy to dy_out (what the weight gradient of the same convolution reads next).  bnb (required): the BatchNorm-backward epilogue
// of the BatchNorm IN FRONT of the convolution, as in dsnt_conv_fwd_f16x3_stream.  a_bound: a bound of |dy|
// (dsnt_bn_bwd_finalize_bound).
extern "C" int dsnt_conv_dgrad_f16x3_stream_apply(const float* dz, const dsnt_bn_bwd_apply* ap, float* dy_out,
                                                  const void* w_planes, int64_t plane_stride, const float* w_bound,
                                                  const float* a_bound, float* dx_dz, float* stats_partial, int flags,
                                                  const dsnt_conv_geom* g, const dsnt_bn_bwd_epilogue* bnb,
                                                  const dsnt_out_bounds* tail, void* stream) {
    DSNT_REQUIRE(a_bound && w_bound, DSNT_ERR_ARG, "dsnt_conv_dgrad_f16x3_stream_apply: the operand bounds (device scalars) are required");
    DSNT_REQUIRE(ap && ap->y && ap->scale && ap->mean && ap->invstd && ap->coef && dy_out && bnb, DSNT_ERR_ARG,
                 "dsnt_conv_dgrad_f16x3_stream_apply: needs a complete dsnt_bn_bwd_apply, dy_out and the BatchNorm-backward epilogue");
    DSNT_REQUIRE(dsnt_aligned16(ap->y) && dsnt_aligned16(dy_out) && dy_out != dz, DSNT_ERR_ALIGN,
                 "dsnt_conv_dgrad_f16x3_stream_apply: 16-byte alignment; dy_out must not alias dz (halo pixels are re-read by other workgroups)");
    return conv_fwd6_impl(dz, w_planes, plane_stride, nullptr, dx_dz, nullptr, nullptr, flags & DSNT_CONV_SHARE_CHIP, nullptr, nullptr,
                          stats_partial, g, bnb, tail, stream, a_bound, w_bound, true, ap, dy_out);
}

// max |src[i]| -> out[0] (bit pattern of a non-negative float: integer max is float max)
__global__ void amax_kernel(const float4* __restrict__ src, unsigned* __restrict__ out, long n4) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        m = fmaxf(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))), m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out + (blockIdx.x & (DSNT_BOUND_SLOTS - 1)), __float_as_uint(m));
}

extern "C" int dsnt_amax(const float* src, int64_t n, float* out, void* stream) {
    DSNT_REQUIRE(src && out && n > 0 && n % 4 == 0 && dsnt_aligned16(src), DSNT_ERR_ARG,
                 "dsnt_amax: n must be a positive multiple of 4, src 16-byte aligned");
    DSNT_REQUIRE(!dsnt_recording(), DSNT_ERR_ARG, "dsnt_amax: cannot be recorded into a launch list (it clears its output with a memset)");
    if (hipMemsetAsync(out, 0, 4 * DSNT_BOUND_SLOTS, (hipStream_t)stream) != hipSuccess) return dsnt_set_error(DSNT_ERR_HIP, "dsnt_amax: memset");
    long g = (n / 4 + 255) / 256;
    if (g > 1024) g = 1024;
    DSNT_LAUNCH(amax_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float4*)src,
                       (unsigned*)out, (long)(n / 4));
    DSNT_CHECK_LAUNCH("dsnt_amax");
}

__global__ void split_f16x2_kernel(const float4* __restrict__ src, uint2* __restrict__ dst, long n4, long stride4,
                                   const float* __restrict__ bound) {
    const float sc = pow2_scale(bound64(bound));
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = src[i];
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        uint2 a, b;
        split4h(v, a, b);
        dst[i] = a; dst[stride4 + i] = b;
    }
}

extern "C" int dsnt_split_f16x2(const float* src, void* dst, int64_t n, int64_t plane_stride, const float* bound,
                                void* stream) {
    DSNT_REQUIRE(src && dst && bound && n > 0 && n % 4 == 0 && plane_stride >= n && plane_stride % 4 == 0, DSNT_ERR_ARG,
                 "dsnt_split_f16x2: n and plane_stride must be positive multiples of 4, plane_stride >= n");
    DSNT_REQUIRE(dsnt_aligned16(src) && (((uintptr_t)dst) & 7u) == 0, DSNT_ERR_ALIGN, "dsnt_split_f16x2: alignment");
    const long n4 = n / 4;
    long g = (n4 + 255) / 256;
    if (g > 2048) g = 2048;
    DSNT_LAUNCH(split_f16x2_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float4*)src,
                       (uint2*)dst, n4, (long)(plane_stride / 4), bound);
    DSNT_CHECK_LAUNCH("dsnt_split_f16x2");
}

// Per-step preparation of the fp16x3 operands of MANY tensors in one launch each (table rows of int64):
//   dsnt_f16_prep_weights: row {src float*, dst fp16 plane 0*, bound float*, count (multiple of 4), plane stride,
//     stream Cout, stream Cin}: one workgroup per row: bound = max|src|, then dst = two fp16 planes of src * pow2_scale(bound);
//     stream Cout > 0: src is an OHWI 3x3 filter [Cout][3][3][Cin] and the planes are written in STREAM order
//     [Cin / 16][9 taps][Cout][16] — the K-step sequence of conv3s.hip, each step one contiguous block;
//   dsnt_f16_prep_bn_bounds: row {gamma float*, beta float*, out float*, C, float bits of sqrt(M)}: out =
//     max_c(|gamma_c| sqrt(M) + |beta_c|) >= every |relu?(bn(x))| of a train-mode BatchNorm over M samples
//     (|(x - mean) / std| <= sqrt(M - 1) for the biased batch variance).
// (1024 threads and four independent loads per thread and pass: one workgroup walks a whole tensor, so the launch lasts as
// long as its largest row — 130 us for a 3x3 128->128 filter with 256 threads and one load in flight, and proportionally longer
// on hg8, where the stem convolution no longer covers it.)
#define PREP_T 1024
__global__ __launch_bounds__(PREP_T) void f16_prep_weights_kernel(const long long* __restrict__ table, int row_ints) {
    __shared__ float red[PREP_T / 64];
    const long long* t = table + (size_t)blockIdx.x * row_ints;
    const float4* src = reinterpret_cast<const float4*>(t[0]);
    uint2* dst = reinterpret_cast<uint2*>(t[1]);
    float* bound = reinterpret_cast<float*>(t[2]);
    const long n4 = (long)t[3] / 4, stride4 = (long)t[4] / 4;
    // > 0: OHWI [Cout][9][Cin] -> stream order [Cin/16][9][Cout][16] (rows of 5 values: always the plain layout)
    const int s_cout = row_ints >= 7 ? (int)t[5] : 0, s_cin = row_ints >= 7 ? (int)t[6] : 0;
    const int s_k4 = 9 * s_cin / 4, s_cin4 = s_cin / 4;
    float m = 0.f;
    long i = threadIdx.x;
    for (; i + 3 * PREP_T < n4; i += 4 * PREP_T) {
        const float4 v0 = src[i], v1 = src[i + PREP_T], v2 = src[i + 2 * PREP_T], v3 = src[i + 3 * PREP_T];
        m = fmaxf(fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))), m);
        m = fmaxf(fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w))), m);
        m = fmaxf(fmaxf(fmaxf(fabsf(v2.x), fabsf(v2.y)), fmaxf(fabsf(v2.z), fabsf(v2.w))), m);
        m = fmaxf(fmaxf(fmaxf(fabsf(v3.x), fabsf(v3.y)), fmaxf(fabsf(v3.z), fabsf(v3.w))), m);
    }
    for (; i < n4; i += PREP_T) {
        const float4 v = src[i];
        m = fmaxf(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))), m);
    }
    m = block_max(m, red);
    if (threadIdx.x < DSNT_BOUND_SLOTS) bound[threadIdx.x] = m;
    const float sc = pow2_scale(m);
    auto put = [&](long j, float4 v) {
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        uint2 a, b;
        split4h(v, a, b);
        if (s_cout > 0) {
            const int n = (int)(j / s_k4), k4 = (int)(j - (long)n * s_k4);
            const int tap = k4 / s_cin4, c4 = k4 - tap * s_cin4;
            j = ((long)((c4 >> 2) * 9 + tap) * s_cout + n) * 4 + (c4 & 3);
        }
        dst[j] = a; dst[stride4 + j] = b;
    };
    i = threadIdx.x;
    for (; i + 3 * PREP_T < n4; i += 4 * PREP_T) {
        const float4 v0 = src[i], v1 = src[i + PREP_T], v2 = src[i + 2 * PREP_T], v3 = src[i + 3 * PREP_T];
        put(i, v0); put(i + PREP_T, v1); put(i + 2 * PREP_T, v2); put(i + 3 * PREP_T, v3);
    }
    for (; i < n4; i += PREP_T) put(i, src[i]);
}

// (the table's row width is an argument: the first form of this entry point read five values per row, and a caller's
// five-wide table must not be walked with a stride of seven)
extern "C" int dsnt_f16_prep_weights(const int64_t* table, int rows, int row_ints, void* stream) {
    DSNT_REQUIRE(table && rows > 0 && (row_ints == 5 || row_ints == 7), DSNT_ERR_ARG,
                 "dsnt_f16_prep_weights: bad argument (row_ints is 5: plain layout only, or 7: with the stream-layout columns)");
    DSNT_LAUNCH(f16_prep_weights_kernel, dim3(rows), dim3(PREP_T), 0, (hipStream_t)stream, (const long long*)table, row_ints);
    DSNT_CHECK_LAUNCH("dsnt_f16_prep_weights");
}

// The stem's per-step weight preparation in ONE small launch (one 256-thread workgroup; 16 K values): the [Cout][4][4][16]
// space-to-depth form of the OHWI [Cout][7][7][4] filter (dsnt_s2d_weights, back = 0), its maximum, and its fp16 (two) and
// bf16 (three) planes — instead of three launches, one of them a 1024-thread workgroup that waits for a free CU at the head of
// every step.
__global__ __launch_bounds__(256) void s2d_weights_prep_kernel(const float* __restrict__ w, float* __restrict__ w2,
                                                               uint2* __restrict__ p16, uint2* __restrict__ p6,
                                                               float* __restrict__ bound, int Cout) {
    __shared__ float red[4];
    const int n4 = Cout * 64;                     // float4 groups of w2: [Cout][4][4][4 blocks-of-4]
    auto fetch = [&](int q) {                     // group q = ((co*4 + R)*4 + S)*4 + (dy*2+dx): four channels of one tap
        const int d = q & 3, S = (q >> 2) & 3, R = (q >> 4) & 3, co = q >> 6;
        const int r = 2 * R + (d >> 1) - 1, s_ = 2 * S + (d & 1) - 1;
        return (r >= 0 && s_ >= 0) ? *reinterpret_cast<const float4*>(w + ((co * 7 + r) * 7 + s_) * 4)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float m = 0.f;
    auto emit = [&](const int q, const float4 v, const float sc) {
        if (w2) reinterpret_cast<float4*>(w2)[q] = v;
        uint2 a, b, c;
        split4(v, a, b, c);
        p6[q] = a; p6[n4 + q] = b; p6[2 * n4 + q] = c;
        split4h(make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc), a, b);
        p16[q] = a; p16[n4 + q] = b;
    };
    if (n4 <= 16 * 256) {
        // (Cout <= 64, every stem: the 16 groups of a thread are fetched at once and kept — this one-workgroup launch sits on the
        // dependency chain at the head of every step, and 2 x 16 dependent round trips to L2 were 24-55 us of it)
        float4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int q = threadIdx.x + 256 * j;
            v[j] = q < n4 ? fetch(q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
            m = fmaxf(fmaxf(fmaxf(fabsf(v[j].x), fabsf(v[j].y)), fmaxf(fabsf(v[j].z), fabsf(v[j].w))), m);
        m = block_max(m, red);
        if (threadIdx.x < DSNT_BOUND_SLOTS) bound[threadIdx.x] = m;
        const float sc = pow2_scale(m);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int q = threadIdx.x + 256 * j;
            if (q < n4) emit(q, v[j], sc);
        }
        return;
    }
    for (int q = threadIdx.x; q < n4; q += 256) {
        const float4 v = fetch(q);
        m = fmaxf(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))), m);
    }
    m = block_max(m, red);
    if (threadIdx.x < DSNT_BOUND_SLOTS) bound[threadIdx.x] = m;
    const float sc = pow2_scale(m);
    for (int q = threadIdx.x; q < n4; q += 256) emit(q, fetch(q), sc);
}

extern "C" int dsnt_s2d_weights_prep(const float* w, float* w2, void* planes16, void* planes_bf16, float* bound, int Cout,
                                     void* stream) {
    DSNT_REQUIRE(w && planes16 && planes_bf16 && bound && Cout > 0, DSNT_ERR_ARG, "dsnt_s2d_weights_prep: bad argument");
    DSNT_REQUIRE(dsnt_aligned16(w) && (!w2 || dsnt_aligned16(w2)) && (((uintptr_t)planes16) & 7u) == 0 &&
                 (((uintptr_t)planes_bf16) & 7u) == 0, DSNT_ERR_ALIGN, "dsnt_s2d_weights_prep: alignment");
    DSNT_LAUNCH(s2d_weights_prep_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w, w2, (uint2*)planes16, (uint2*)planes_bf16,
                bound, Cout);
    DSNT_CHECK_LAUNCH("dsnt_s2d_weights_prep");
}

__global__ __launch_bounds__(64) void f16_prep_bn_bounds_kernel(const long long* __restrict__ table) {
    const long long* t = table + (size_t)blockIdx.x * 5;
    const float* gamma = reinterpret_cast<const float*>(t[0]);
    const float* beta = reinterpret_cast<const float*>(t[1]);
    float* out = reinterpret_cast<float*>(t[2]);
    const int C = (int)t[3];
    const float sqrtM = __uint_as_float((unsigned)t[4]);
    float m = 0.f;
    for (int c = threadIdx.x; c < C; c += 64) m = fmaxf(m, fmaf(fabsf(gamma[c]), sqrtM, fabsf(beta[c])));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    out[threadIdx.x] = m;                    // all DSNT_BOUND_SLOTS entries
}

extern "C" int dsnt_f16_prep_bn_bounds(const int64_t* table, int rows, void* stream) {
    DSNT_REQUIRE(table && rows > 0, DSNT_ERR_ARG, "dsnt_f16_prep_bn_bounds: bad argument");
    DSNT_LAUNCH(f16_prep_bn_bounds_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, (const long long*)table);
    DSNT_CHECK_LAUNCH("dsnt_f16_prep_bn_bounds");
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
    DSNT_LAUNCH(pack_dgrad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wd, Cout,
                       R, S, Cin);
    DSNT_CHECK_LAUNCH("dsnt_conv_pack_dgrad");
}

// All data-gradient weight packs of a backward pass in ONE launch: for conv c (table row c =
// {src offset, dst offset, Cout, R, S, Cin}) write wd[ci][R-1-r][S-1-s][co] = w[co][r][s][ci] as fp32
// and as three bf16 planes (plane stride = `total` elements).
// One tap of one convolution is a [Cout][Cin] matrix with row pitch R S Cin; it lands transposed, [Cin][Cout] with row pitch
// R S Cout.  32 x 32 tiles through LDS: 128-byte runs along ci on the way in, along co on the way out (element by element the
// reads were 4 bytes per cache line: 360 MB fetched for the 27 MB of hg2's weights, and — 24 000 workgroups for hg8 — a flood that
// kept the dependency chain's 4-workgroup BatchNorm finalise waiting 140-240 us for a slot at the start of every step).
__global__ __launch_bounds__(256) void pack_dgrad_all_kernel(const int* __restrict__ table, const float* __restrict__ params,
                                                            float* __restrict__ out, unsigned short* __restrict__ planes, long total) {
    __shared__ float tl[32][33];
    const int* t = table + blockIdx.y * 6;
    const int src = t[0], dst = t[1], Cout = t[2], R = t[3], S = t[4], Cin = t[5];
    const float* w = params + src;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int cits = (Cin + 31) >> 5, cots = (Cout + 31) >> 5, RS = R * S;
    const int ntiles = RS * cits * cots;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tap = tile % RS;
        int q = tile / RS;
        const int cit = q % cits, cot = q / cits;
        const int r = tap / S, s_ = tap - r * S;            // DESTINATION tap; the source is the flipped one
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = cot * 32 + ty + 8 * i, ci = cit * 32 + tx;
            tl[ty + 8 * i][tx] = (co < Cout && ci < Cin) ? w[((co * R + (R - 1 - r)) * S + (S - 1 - s_)) * Cin + ci] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ci = cit * 32 + ty + 8 * i, co = cot * 32 + tx;
            if (ci < Cin && co < Cout) {
                const float v = tl[tx][ty + 8 * i];
                const long o = (long)dst + ((long)(ci * R + r) * S + s_) * Cout + co;
                out[o] = v;
                // exact 3-way bf16 split (round-to-nearest-even by hand: one scalar at a time)
                float rem = v;
                for (int pl = 0; pl < 3; ++pl) {
                    unsigned u = __float_as_uint(rem);
                    unsigned rb = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
                    planes[(long)pl * total + o] = (unsigned short)(rb >> 16);
                    rem -= __uint_as_float(rb);
                }
            }
        }
        __syncthreads();
    }
}

extern "C" int dsnt_conv_pack_dgrad_all(const int* table, int nconv, const float* params, float* out,
                                        void* planes, int64_t total, void* stream) {
    DSNT_REQUIRE(table && params && out && planes && nconv > 0 && total > 0, DSNT_ERR_ARG,
                 "dsnt_conv_pack_dgrad_all: bad argument");
    DSNT_LAUNCH(pack_dgrad_all_kernel, dim3(16, nconv), dim3(256), 0, (hipStream_t)stream, table, params,
                       out, (unsigned short*)planes, (long)total);
    DSNT_CHECK_LAUNCH("dsnt_conv_pack_dgrad_all");
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
    // fp16x3: bound slots of the A operand (after its BN+ReLU prologue) and of dy; null on the bf16x6 path
    const float* a_bound; const float* g_bound;
};

#define WPITCH 132

// VALU diet (fp32 MFMA and VALU are serialised on gfx950): each thread's filter tap / channel
// chunk is fixed for the whole kernel, rows advance by 32 per step with incremental (n, oh, ow)
// carries instead of divisions, loads are range-checked buffer loads (invalid rows / padded taps
// point out of range and come back as zeros), two register sets keep two steps of loads in flight.
template <bool PRO>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradP p) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float As[2][32][WPITCH];
    __shared__ __attribute__((aligned(16))) float Gs[2][32][WPITCH];

    int bid;
    xcd_remap(blockIdx.x, gridDim.x, bid);
    const int ktile = bid % p.ktiles; bid /= p.ktiles;
    const int ntile = bid % p.ntiles;
    const int split = bid / p.ntiles;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1;   // wave owns k rows [64wk,64wk+64), n cols [64wn, ..)
    const int li = lane & 31, lm = lane >> 5;
    const int lrow = tid >> 5, cc = tid & 31;  // loader: row lrow + 8i, float4 column cc

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
    const bool vn = n0 < p.Cout;

    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int HoWo = p.Ho * p.Wo;
    const int adv_h = 32 / p.Wo, adv_w = 32 - adv_h * p.Wo;     // scalars: 32 rows = adv_h rows + adv_w cols

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * p.Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy), 0, (int)((size_t)p.M * p.Cout * 4u), 0x00020000);
    const unsigned OOB = 0xF0000000u;

    // per-row state of this thread's four rows
    int rn[4], roh[4], row_[4], rm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m_begin + lrow + 8 * i;
        rm[i] = m;
        const int mm = m < p.M ? m : 0;
        rn[i] = mm / HoWo;
        const int rem = mm - rn[i] * HoWo;
        roh[i] = rem / p.Wo;
        row_[i] = rem - roh[i] * p.Wo;
    }
    struct Stage { u32x4 a[4], g[4]; unsigned ok; };
    Stage S0, S1;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    // issue the loads of the current rows, then advance the rows by 32
    auto gload = [&](Stage& st) {
        st.ok = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool vm = rm[i] < m_end;
            const int ih = roh[i] * p.stride + dh, iw = row_[i] * p.stride + dw;
            const bool oka = vm && vk && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            const unsigned offa = oka ? (unsigned)(((rn[i] * p.H + ih) * p.W + iw) * p.Cin + c) * 4u : OOB;
            st.a[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, offa, 0, 0);
            const unsigned offg = (vm && vn) ? (unsigned)(rm[i] * p.Cout + n0) * 4u : OOB;
            st.g[i] = __builtin_amdgcn_raw_buffer_load_b128(gr, offg, 0, 0);
            st.ok |= (oka ? 1u : 0u) << i;
            // advance
            rm[i] += 32;
            row_[i] += adv_w;
            roh[i] += adv_h;
            if (row_[i] >= p.Wo) { row_[i] -= p.Wo; roh[i] += 1; }
            while (roh[i] >= p.Ho) { roh[i] -= p.Ho; rn[i] += 1; }
        }
    };
    auto lstore = [&](const Stage& st, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 va = make_float4(__uint_as_float(st.a[i].x), __uint_as_float(st.a[i].y),
                                    __uint_as_float(st.a[i].z), __uint_as_float(st.a[i].w));
            const float4 vg = make_float4(__uint_as_float(st.g[i].x), __uint_as_float(st.g[i].y),
                                          __uint_as_float(st.g[i].z), __uint_as_float(st.g[i].w));
            if (PRO) {
                va.x = fmaf(va.x, sc.x, sh.x); va.y = fmaf(va.y, sc.y, sh.y);
                va.z = fmaf(va.z, sc.z, sh.z); va.w = fmaf(va.w, sc.w, sh.w);
                if (p.in_relu) {
                    va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f);
                    va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
                }
                const bool ok = (st.ok >> i) & 1u;     // branch-free (see the forward loader)
                va.x = ok ? va.x : 0.f; va.y = ok ? va.y : 0.f; va.z = ok ? va.z : 0.f; va.w = ok ? va.w : 0.f;
            }
            *reinterpret_cast<float4*>(&As[buf][lrow + 8 * i][cc * 4]) = va;   // OOB loads are zeros
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

    auto compute = [&](int buf) {
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
    };

    const int nsteps = (m_end - m_begin + 31) / 32;
    // S0 holds step 0 (then 2, 4, ...), S1 holds step 1 (3, 5, ...): two steps of loads in flight
    // unconditional on purpose (hipcc's vmcnt bookkeeping is exact only on straight-line code, see the
    // forward loader): rows past m_end load nothing (out-of-range buffer offsets) and store zeros
    gload(S0);
    gload(S1);
    lstore(S0, 0);
    gload(S0);
    __syncthreads();
    int st = 0;
    for (; st + 1 < nsteps; st += 2) {
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        lstore(S1, 1);
        gload(S1);
        __syncthreads();
        compute(1);
        __builtin_amdgcn_sched_barrier(0);
        lstore(S0, 0);
        gload(S0);
        __syncthreads();
    }
    if (st < nsteps) compute(0);

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

// bf16x6 weight gradient: D[k][n] = sum_m A[m][k] G[m][n] on the bf16 matrix cores (see the
// forward bf16x6 kernel for the numerics).  The reduction index of the MFMA is m, so both LDS tiles
// are stored transposed ([k][m] and [n][m], m contiguous): a loader thread owns a 4(m) x 4(k or n)
// block, loads four rows, applies BN+ReLU / zero padding (A only), splits and packs pairs of ROWS with
// v_cvt_pk_bf16_f32, i.e. the transpose costs no extra instruction.  128 threads stage A, 128 stage G;
// waves 0..3 run the MFMAs (64 x 64 of the 128 x 128 tile each), one barrier per 16 rows of m.
//
// The kernel is bound by the loaders' VALU work (a SIMD issues either an MFMA or a VALU instruction:
// DESIGN.md "issue starvation"), so the two loader kinds are separate straight-line instantiations:
// the G waves only split (out-of-range buffer loads already return zeros), the A waves fold ReLU and
// the zero-padding select into one v_med3_f32 against per-row bounds, and the split itself runs on
// plain (unpacked) fp32 ops (split4).
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool PRO, bool IS_A, bool F16>
__device__ __forceinline__ void wgrad6_loader(const WgradP& p, __bf16* T, const int ltid, const int ktile,
                                              const int ntile, const int m_begin, const int m_end,
                                              const int nsteps, f32x2& bs0, f32x2& bs1) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int NPL = F16 ? 2 : 3;
    // fp16x3: power-of-two operand scale from the bound slots (A: folded into the BN vectors below; dY: one multiply)
    const float sop = F16 ? pow2_scale(bound64(IS_A ? p.a_bound : p.g_bound)) : 1.f;
    // 4-wide column chunk q, 4-row block mb.  mb varies fastest: a 16-lane store group then spans
    // 4 chunks x 4 blocks (2-way bank conflicts; q fastest would be 8-way with the 48-byte pitch)
    const int mb = ltid & 3, q = (ltid >> 2) & 31;
    const unsigned OOB = 0xF0000000u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(IS_A ? p.x : p.dy), 0,
        IS_A ? (int)((size_t)p.N * p.H * p.W * p.Cin * 4u) : (int)((size_t)p.M * p.Cout * 4u), 0x00020000);
    // A side: fixed tap / channel chunk
    const int k0 = ktile * 128 + q * 4;
    const bool vk = k0 < p.K;
    const int tap = k0 / p.Cin, c = k0 - tap * p.Cin;
    const int r = tap / p.S, s_ = tap - r * p.S;
    const int dh = r * p.dil - p.pad, dw = s_ * p.dil - p.pad;
    f32x2 sc0 = {1.f, 1.f}, sc1 = {1.f, 1.f}, sh0 = {0.f, 0.f}, sh1 = {0.f, 0.f};
    if (PRO && IS_A && vk) {
        const float4 a = *reinterpret_cast<const float4*>(p.in_scale + c);
        const float4 b = *reinterpret_cast<const float4*>(p.in_shift + c);
        sc0 = (f32x2){a.x, a.y}; sc1 = (f32x2){a.z, a.w};
        sh0 = (f32x2){b.x, b.y}; sh1 = (f32x2){b.z, b.w};
        if (F16) { sc0 *= sop; sc1 *= sop; sh0 *= sop; sh1 *= sop; }
    }
    // ReLU and the padding select as one median: valid rows clamp to [lo, +inf) with lo = 0 (ReLU) or
    // -inf (no ReLU), invalid rows to [0, 0]
    const float lo_valid = (PRO && IS_A && p.in_relu) ? 0.f : -__builtin_inff();
    // G side
    const int n0 = ntile * 128 + q * 4;
    const bool vn = n0 < p.Cout;
    // first row of this thread's 4-row block (Wo % 4 == 0: the 4 rows share n and oh)
    const int HoWo = p.Ho * p.Wo;
    int rm = m_begin + mb * 4;
    const int mm0 = rm < p.M ? rm : 0;
    int rn = mm0 / HoWo;
    int roh = (mm0 - rn * HoWo) / p.Wo;
    int row_ = mm0 - rn * HoWo - roh * p.Wo;
    const int adv_h = 16 / p.Wo, adv_w = 16 - adv_h * p.Wo;
    struct Stage { u32x4 v[4]; unsigned ok; };
    Stage S0, S1;
    auto gload = [&](Stage& st) {
        st.ok = 0;
        if (IS_A) {
            const int ih = roh * p.stride + dh;
            const bool vrow = vk && ih >= 0 && ih < p.H;
            const int base = ((rn * p.H + ih) * p.W) * p.Cin + c;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iw = (row_ + j) * p.stride + dw;
                const bool ok = (rm + j) < m_end && vrow && iw >= 0 && iw < p.W;
                const unsigned off = (unsigned)(base + iw * p.Cin) * 4u;
                st.v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : OOB, 0, 0);
                st.ok |= (ok ? 1u : 0u) << j;
            }
            row_ += adv_w; roh += adv_h;
            if (row_ >= p.Wo) { row_ -= p.Wo; roh += 1; }
            // 16 rows cross at most one image boundary when an image has >= 16 pixels (branch-free); tiny maps loop
            if (HoWo >= 16) { const bool wrap = roh >= p.Ho; roh = wrap ? roh - p.Ho : roh; rn = wrap ? rn + 1 : rn; }
            else while (roh >= p.Ho) { roh -= p.Ho; rn += 1; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = (rm + j) < m_end && vn;
                const unsigned off = (unsigned)((rm + j) * p.Cout + n0) * 4u;
                st.v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : OOB, 0, 0);
            }
        }
        rm += 16;
    };
    auto lstore = [&](const Stage& st, int buf) {
        f32x2 v0[4], v1[4];      // (x, y) and (z, w) of the four rows
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v0[j] = (f32x2){__uint_as_float(st.v[j].x), __uint_as_float(st.v[j].y)};
            v1[j] = (f32x2){__uint_as_float(st.v[j].z), __uint_as_float(st.v[j].w)};
            if (PRO && IS_A) {
                v0[j].x = fmaf(v0[j].x, sc0.x, sh0.x); v0[j].y = fmaf(v0[j].y, sc0.y, sh0.y);
                v1[j].x = fmaf(v1[j].x, sc1.x, sh1.x); v1[j].y = fmaf(v1[j].y, sc1.y, sh1.y);
                const bool ok = (st.ok >> j) & 1u;
                const float lo = ok ? lo_valid : 0.f, hi = ok ? __builtin_inff() : 0.f;
                v0[j].x = __builtin_amdgcn_fmed3f(v0[j].x, lo, hi); v0[j].y = __builtin_amdgcn_fmed3f(v0[j].y, lo, hi);
                v1[j].x = __builtin_amdgcn_fmed3f(v1[j].x, lo, hi); v1[j].y = __builtin_amdgcn_fmed3f(v1[j].y, lo, hi);
            }
            if (!IS_A) { bs0.x += v0[j].x; bs0.y += v0[j].y; bs1.x += v1[j].x; bs1.y += v1[j].y; }
        }
        __bf16* base = T + ((size_t)(buf * NPL) * 128 + q * 4) * PITCH6 + mb * 4;
        // component e of the four rows -> LDS row (q*4 + e), columns mb*4 .. mb*4+3, three planes
#define SPLIT_COL(E, V, COMP)                                                                        \
        {                                                                                            \
            uint2 q1, q2, q3;                                                                        \
            float4 c4 = make_float4(V[0].COMP, V[1].COMP, V[2].COMP, V[3].COMP);                     \
            __bf16* d = base + (E) * PITCH6;                                                         \
            if (F16) {                                                                               \
                if (!(PRO && IS_A)) { c4.x *= sop; c4.y *= sop; c4.z *= sop; c4.w *= sop; }          \
                split4h(c4, q1, q2);                                                                 \
                *reinterpret_cast<uint2*>(d) = q1;                                                   \
                *reinterpret_cast<uint2*>(d + 128 * PITCH6) = q2;                                    \
            } else {                                                                                 \
                split4(c4, q1, q2, q3);                                                              \
                *reinterpret_cast<uint2*>(d) = q1;                                                   \
                *reinterpret_cast<uint2*>(d + 128 * PITCH6) = q2;                                    \
                *reinterpret_cast<uint2*>(d + 2 * 128 * PITCH6) = q3;                                \
            }                                                                                        \
        }
        SPLIT_COL(0, v0, x) SPLIT_COL(1, v0, y) SPLIT_COL(2, v1, x) SPLIT_COL(3, v1, y)
#undef SPLIT_COL
    };
    // unconditional (see the forward loader): rows past m_end load nothing and store zeros
    gload(S0);
    gload(S1);
    lstore(S0, 0);
    gload(S0);
    __syncthreads();
    int s = 0;
    for (; s + 1 < nsteps; s += 2) {
        lstore(S1, 1);
        gload(S1);
        __syncthreads();
        lstore(S0, 0);
        gload(S0);
        __syncthreads();
    }
    if (s < nsteps) __syncthreads();
}

template <bool PRO, bool F16 = false>
__device__ __forceinline__ void wgrad6_body(const WgradP& p, int bid, float* smem) {
    constexpr int NPL = F16 ? 2 : 3;
    __bf16* At = reinterpret_cast<__bf16*>(smem);            // [2][NPL][128][PITCH6]
    __bf16* Gt = At + 2 * NPL * 128 * PITCH6;                // [2][NPL][128][PITCH6]

    const int ktile = bid % p.ktiles; bid /= p.ktiles;
    const int ntile = bid % p.ntiles;
    const int split = bid / p.ntiles;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nsteps = (m_end - m_begin + 15) / 16;

    f32x2 bs0 = {0.f, 0.f}, bs1 = {0.f, 0.f};

    if (wave >= 6) {
        wgrad6_loader<PRO, false, F16>(p, Gt, tid - 384, ktile, ntile, m_begin, m_end, nsteps, bs0, bs1);
    } else if (wave >= 4) {
        wgrad6_loader<PRO, true, F16>(p, At, tid - 256, ktile, ntile, m_begin, m_end, nsteps, bs0, bs1);
    } else {
        // ------------------------------------------------------------------ MFMA waves
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
        const int wk = wave >> 1, wn = wave & 1;
        struct Frag { bf16x8 a[2][3], b[2][3]; };
        Frag F;
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const int buf = s & 1;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    F.a[t][pl] = *reinterpret_cast<const bf16x8*>(
                        At + ((size_t)(buf * NPL + pl) * 128 + wk * 64 + t * 32 + lr) * PITCH6 + 8 * lh);
                    F.b[t][pl] = *reinterpret_cast<const bf16x8*>(
                        Gt + ((size_t)(buf * NPL + pl) * 128 + wn * 64 + t * 32 + lr) * PITCH6 + 8 * lh);
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) mma_split<F16>(acc[a][b], F.a[a], F.b[b]);
            __syncthreads();
        }
        // slab store: ws[split][n][k], D row = k (regs, 4 consecutive), D col = n (lane)
        // (fp16x3: the accumulators hold (A s_a)^T (dY s_g); both scales are powers of two, undone exactly)
        const float osc = F16 ? 1.f / (pow2_scale(bound64(p.a_bound)) * pow2_scale(bound64(p.g_bound))) : 1.f;
        float* slab = p.ws + (size_t)split * p.Cout * p.K;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = ntile * 128 + wn * 64 + b * 32 + lr;
            if (n >= p.Cout) continue;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int k = ktile * 128 + wk * 64 + a * 32 + 8 * qq + 4 * lh;
                    if (k < p.K)
                        *reinterpret_cast<float4*>(slab + (size_t)n * p.K + k) =
                            make_float4(acc[a][b][4 * qq + 0] * osc, acc[a][b][4 * qq + 1] * osc,
                                        acc[a][b][4 * qq + 2] * osc, acc[a][b][4 * qq + 3] * osc);
                }
        }
    }
    // bias partial: column sums of this split's dY rows (G loader threads of the ktile-0 blocks)
    if (ktile == 0) {
        float* red = smem;   // [4][128] floats
        __syncthreads();
        if (wave >= 6) {
            const int ltid = tid - 384;
            *reinterpret_cast<float4*>(red + (ltid & 3) * 128 + ((ltid >> 2) & 31) * 4) =
                make_float4(bs0.x, bs0.y, bs1.x, bs1.y);
        }
        __syncthreads();
        if (tid < 128) {
            const int n = ntile * 128 + tid;
            if (n < p.Cout)
                p.ws[(size_t)p.splits * p.Cout * p.K + (size_t)split * p.Cout + n] =
                    red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid];
        }
    }
}

template <bool PRO, bool F16 = false>
__global__ __launch_bounds__(512, 2) void conv_wgrad_bf16x6_kernel(WgradP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // the k-tiles (filter taps) and n-tiles of one split read the same rows of x and dy: keep them in one
    // XCD (one L2) instead of dealing them round-robin over the eight
    int bid;
    xcd_remap(blockIdx.x, gridDim.x, bid);
    wgrad6_body<PRO, F16>(p, bid, smem);
}

// Grouped launch: blockIdx.y picks one of many convolutions from a device table of WgradP descriptors
// (dsnt_conv_wgrad_desc), blockIdx.x is the block within it.  The ~50 weight gradients of the 16x16 ... 4x4
// hourglass levels have 4 ... 288 workgroups each and take 25-55 us apiece as separate launches (a serial
// chain of 16-row steps per workgroup, nothing to overlap with); nothing downstream in backward needs them,
// so the engine defers them to the end of their parameter bucket and runs them side by side in ONE launch.
__global__ __launch_bounds__(512, 2) void conv_wgrad_bf16x6_group_kernel(const WgradP* __restrict__ table) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WgradP p = table[blockIdx.y];
    const int nblk = p.ktiles * p.ntiles * p.splits;
    if ((int)blockIdx.x >= nblk) return;
    if (p.a_bound) wgrad6_body<true, true>(p, blockIdx.x, smem);        // workgroup-uniform: fp16x3 descriptors
    else wgrad6_body<true, false>(p, blockIdx.x, smem);
}

// Slab reduction: 64 float4 columns x 4 split-lanes per block, 8 loads in flight per thread.
// RC float4 columns x SL = 256 / RC slab lanes per 256-thread block: a block reads RC * 16 contiguous bytes of every slab (WRED_RC: A/B)
#ifndef WRED_RC
#define WRED_RC 32
#endif
#ifndef WRED_U
#define WRED_U 8        /* independent 16-byte loads in flight per thread */
#endif
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ ws, float* __restrict__ dw,
                                                  float* __restrict__ dbias, int splits, int CK, int Cout,
                                                  int accumulate, int block) {
    constexpr int RC = WRED_RC, SL = 256 / RC;
    __shared__ float4 red[256];
    const int total4 = CK / 4;
    const int col = threadIdx.x % RC, sl = threadIdx.x / RC;
    const int i = block * RC + col;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < total4) {
        int s = sl;
        for (; s + (WRED_U - 1) * SL < splits; s += WRED_U * SL) {
            float4 v[WRED_U];
#pragma unroll
            for (int u = 0; u < WRED_U; ++u)
                v[u] = *reinterpret_cast<const float4*>(ws + (size_t)(s + SL * u) * CK + (size_t)i * 4);
#pragma unroll
            for (int u = 0; u < WRED_U; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
        for (; s < splits; s += SL) {
            const float4 v = *reinterpret_cast<const float4*>(ws + (size_t)s * CK + (size_t)i * 4);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    } else if (dbias && i - total4 < (Cout + 3) / 4) {
        // bias partials live behind the slabs: [splits][Cout]
        const int n0 = (i - total4) * 4;
        const float* b = ws + (size_t)splits * CK;
        for (int s = sl; s < splits; s += SL) {
            const float* q = b + (size_t)s * Cout + n0;
            a.x += q[0]; if (n0 + 1 < Cout) a.y += q[1]; if (n0 + 2 < Cout) a.z += q[2]; if (n0 + 3 < Cout) a.w += q[3];
        }
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (sl == 0) {
        for (int j = 1; j < SL; ++j) {
            const float4 v = red[j * RC + col];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        if (i < total4) {
            float4* o = reinterpret_cast<float4*>(dw) + i;
            if (accumulate) { const float4 c = *o; a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w; }
            *o = a;
        } else if (dbias && i - total4 < (Cout + 3) / 4) {
            const int n0 = (i - total4) * 4;
            const float v4[4] = {a.x, a.y, a.z, a.w};
            for (int e = 0; e < 4 && n0 + e < Cout; ++e)
                dbias[n0 + e] = accumulate ? dbias[n0 + e] + v4[e] : v4[e];
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                            float* __restrict__ dbias, int splits, int CK, int Cout,
                                                            int accumulate) {
    wgrad_reduce_body(ws, dw, dbias, splits, CK, Cout, accumulate, blockIdx.x);
}

// Deferred slab reduction of many convolutions in one launch: blockIdx.y = table row
// {ws, dw, dbias (0 = none), splits, Cout*K, Cout, accumulate}, blockIdx.x = 64-column block of that row.
__global__ __launch_bounds__(256) void wgrad_reduce_all_kernel(const long long* __restrict__ table) {
    const long long* t = table + (size_t)blockIdx.y * 7;
    const int CK = (int)t[4], Cout = (int)t[5];
    const int total = CK / 4 + (Cout + 3) / 4;
    if ((int)blockIdx.x * WRED_RC >= total) return;
    wgrad_reduce_body(reinterpret_cast<const float*>(t[0]), reinterpret_cast<float*>(t[1]),
                      reinterpret_cast<float*>(t[2]), (int)t[3], CK, Cout, (int)t[6], blockIdx.x);
}

extern "C" int dsnt_wgrad_reduce_all(const int64_t* table, int rows, int max_blocks, void* stream) {
    DSNT_REQUIRE(table && rows > 0 && rows <= 65535 && max_blocks > 0, DSNT_ERR_ARG, "dsnt_wgrad_reduce_all: bad argument");
    // (max_blocks counts 64-column blocks: the interface's unit)
    DSNT_LAUNCH(wgrad_reduce_all_kernel, dim3((max_blocks * 64 + WRED_RC - 1) / WRED_RC, rows), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)table);
    DSNT_CHECK_LAUNCH("dsnt_wgrad_reduce_all");
}

static void wgrad_plan(const dsnt_conv_geom* g, int& ktiles, int& ntiles, int& splits, int& rps) {
    const long M = (long)g->N * g->Ho * g->Wo;
    const int K = g->R * g->S * g->Cin;
    ktiles = (K + 127) / 128;
    ntiles = (g->Cout + 127) / 128;
    // 256 workgroups = one per CU: these launches run at one workgroup per CU beside the dependency chain anyway
    // (DSNT_WGRAD_SHARE_CHIP), and half as many splits are half the slab traffic (512 measured +0.2 ms per hg2 step)
    const long target = 256;
    long want = target / (ktiles * ntiles);
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

static int64_t wgrad_ws_floats_plain(const dsnt_conv_geom* g) {
    int kt, nt, sp, rps;
    wgrad_plan(g, kt, nt, sp, rps);
    return (int64_t)sp * g->Cout * (g->R * g->S * g->Cin) + (int64_t)sp * g->Cout;
}
// Enough for ANY of the weight-gradient entry points on this geometry (the fp16x3 kernels of wgrad3.hip / wgrad1.hip cut
// the pixels into their own, sometimes more, slabs: dsnt_conv_wgrad_f16x3_ws_floats is the exact size of that call)
// `accumulate` flags -> the halo kernel's plan: 0 whole chip, 1 DSNT_WGRAD_SHARE_CHIP, 2 with DSNT_WGRAD_NARROW on top
static inline int wg3_share(int accumulate) {
    return (accumulate & DSNT_WGRAD_SHARE_CHIP) ? ((accumulate & DSNT_WGRAD_NARROW) ? 2 : 1) : 0;
}
extern "C" int64_t dsnt_conv_wgrad_ws_floats(const dsnt_conv_geom* g) {
    if (!g) return 0;
    int64_t n = wgrad_ws_floats_plain(g);
    const int64_t per = (int64_t)g->Cout * (g->R * g->S * g->Cin) + g->Cout;
    if (const int s4 = dsnt_stem4_wgrad_slabs(g)) { if (s4 * per > n) n = s4 * per; }
    for (int share = 0; share < 2; ++share) {
        const Wg3Plan p3 = dsnt_wg3_plan(g, share);
        if (p3.ok && p3.nslabs * per > n) n = p3.nslabs * per;
        const Wg1Plan p1 = dsnt_wg1_plan(g, share != 0);
        if (p1.ok && p1.nsplits * per > n) n = p1.nsplits * per;
    }
    return n;
}

extern "C" int dsnt_conv_wgrad_splits(const dsnt_conv_geom* g) {
    if (!g) return 0;
    int kt, nt, sp, rps;
    wgrad_plan(g, kt, nt, sp, rps);
    return sp;
}

static int conv_wgrad_impl(const float* x, const float* in_scale, const float* in_shift, int in_relu,
                           const float* dy, float* ws, float* dw, float* dbias, int accumulate,
                           const dsnt_conv_geom* g, void* stream, bool bf16x6, const float* a_bound = nullptr,
                           const float* g_bound = nullptr);

extern "C" int dsnt_conv_wgrad(const float* x, const float* in_scale, const float* in_shift,
                               int in_relu, const float* dy, float* ws, float* dw, float* dbias,
                               int accumulate, const dsnt_conv_geom* g, void* stream) {
    return conv_wgrad_impl(x, in_scale, in_shift, in_relu, dy, ws, dw, dbias, accumulate, g, stream, false);
}

extern "C" int dsnt_conv_wgrad_bf16x6_ok(const dsnt_conv_geom* g) {
    if (!g) return 0;
    return g->Cin % 4 == 0 && g->Cout % 4 == 0 && g->Wo % 4 == 0 &&
           (size_t)g->N * g->H * g->W * g->Cin * 4u < (1ull << 31) &&
           (size_t)g->N * g->Ho * g->Wo * g->Cout * 4u < (1ull << 31);
}

extern "C" int dsnt_conv_wgrad_bf16x6(const float* x, const float* in_scale, const float* in_shift,
                                      int in_relu, const float* dy, float* ws, float* dw, float* dbias,
                                      int accumulate, const dsnt_conv_geom* g, void* stream) {
    DSNT_REQUIRE(dsnt_conv_wgrad_bf16x6_ok(g), DSNT_ERR_SHAPE,
                 "dsnt_conv_wgrad_bf16x6: geometry not supported (need Wo %% 4 == 0, tensors < 2 GiB)");
    return conv_wgrad_impl(x, in_scale, in_shift, in_relu, dy, ws, dw, dbias, accumulate, g, stream, true);
}

static void wgrad_fill(WgradP& p, const float* x, const float* in_scale, const float* in_shift, int in_relu,
                       const float* dy, float* ws, const dsnt_conv_geom* g, const float* a_bound = nullptr,
                       const float* g_bound = nullptr) {
    memset(&p, 0, sizeof(p));
    p.a_bound = a_bound; p.g_bound = g_bound;
    p.x = x; p.in_scale = in_scale; p.in_shift = in_shift; p.dy = dy; p.ws = ws; p.in_relu = in_relu;
    p.N = g->N; p.H = g->H; p.W = g->W; p.Cin = g->Cin; p.Ho = g->Ho; p.Wo = g->Wo;
    p.Cout = g->Cout; p.R = g->R; p.S = g->S; p.stride = g->stride; p.pad = g->pad; p.dil = g->dil;
    p.M = g->N * g->Ho * g->Wo; p.K = g->R * g->S * g->Cin;
    wgrad_plan(g, p.ktiles, p.ntiles, p.splits, p.rows_per_split);
}

extern "C" int dsnt_conv_wgrad_desc_bytes(void) { return (int)sizeof(WgradP); }

extern "C" int dsnt_conv_wgrad_desc(const float* x, const float* in_scale, const float* in_shift, int in_relu,
                                    const float* dy, float* ws, const dsnt_conv_geom* g, void* desc_out) {
    if (int e = check_geom(g, "dsnt_conv_wgrad_desc")) return -e;
    if (!x || !dy || !ws || !in_scale || !in_shift || !desc_out || !dsnt_conv_wgrad_bf16x6_ok(g) ||
        !dsnt_aligned16(x) || !dsnt_aligned16(dy) || !dsnt_aligned16(ws)) {
        return -dsnt_set_error(DSNT_ERR_ARG, "dsnt_conv_wgrad_desc: needs x, dy, ws (16-byte aligned), in_scale/in_shift and a "
                                             "geometry dsnt_conv_wgrad_bf16x6_ok accepts");
    }
    WgradP p;
    wgrad_fill(p, x, in_scale, in_shift, in_relu, dy, ws, g);
    memcpy(desc_out, &p, sizeof(p));
    return p.ktiles * p.ntiles * p.splits;
}

extern "C" int dsnt_conv_wgrad_desc_f16x3(const float* x, const float* in_scale, const float* in_shift, int in_relu,
                                          const float* dy, float* ws, const float* a_bound, const float* g_bound,
                                          const dsnt_conv_geom* g, void* desc_out) {
    if (int e = check_geom(g, "dsnt_conv_wgrad_desc_f16x3")) return -e;
    if (!x || !dy || !ws || !in_scale || !in_shift || !desc_out || !a_bound || !g_bound || !dsnt_conv_wgrad_bf16x6_ok(g) ||
        !dsnt_aligned16(x) || !dsnt_aligned16(dy) || !dsnt_aligned16(ws)) {
        return -dsnt_set_error(DSNT_ERR_ARG, "dsnt_conv_wgrad_desc_f16x3: needs x, dy, ws (16-byte aligned), in_scale/in_shift, "
                                             "both operand bounds and a geometry dsnt_conv_wgrad_bf16x6_ok accepts");
    }
    WgradP p;
    wgrad_fill(p, x, in_scale, in_shift, in_relu, dy, ws, g, a_bound, g_bound);
    memcpy(desc_out, &p, sizeof(p));
    return p.ktiles * p.ntiles * p.splits;
}

extern "C" int dsnt_conv_wgrad_group(const void* table, int nconv, int max_blocks, void* stream) {
    DSNT_REQUIRE(table && nconv > 0 && nconv <= 65535 && max_blocks > 0, DSNT_ERR_ARG,
                 "dsnt_conv_wgrad_group: bad argument");
    const int lds = 2 * 2 * 3 * 128 * PITCH6 * 2;
    DSNT_SET_MAX_LDS(conv_wgrad_bf16x6_group_kernel, lds);
    DSNT_LAUNCH(conv_wgrad_bf16x6_group_kernel, dim3(max_blocks, nconv), dim3(512), lds, (hipStream_t)stream,
                       (const WgradP*)table);
    DSNT_CHECK_LAUNCH("dsnt_conv_wgrad_group");
}

extern "C" int dsnt_conv_wgrad_f16x3(const float* x, const float* in_scale, const float* in_shift, int in_relu,
                                     const float* dy, float* ws, float* dw, float* dbias, int accumulate,
                                     const float* a_bound, const float* g_bound, const dsnt_conv_geom* g, void* stream) {
    DSNT_REQUIRE(dsnt_conv_wgrad_bf16x6_ok(g), DSNT_ERR_SHAPE,
                 "dsnt_conv_wgrad_f16x3: geometry not supported (need Wo %% 4 == 0, tensors < 2 GiB)");
    DSNT_REQUIRE(a_bound && g_bound, DSNT_ERR_ARG, "dsnt_conv_wgrad_f16x3: both operand bounds are required");
    // the stem's space-to-depth convolution (4x4, 16 -> 64 channels, raw operand): its own kernel (stem4.hip), one slab per workgroup
    if (const int s4 = dsnt_stem4_wgrad_slabs(g)) {
        if (int e = check_geom(g, "dsnt_conv_wgrad_f16x3")) return e;
        DSNT_REQUIRE(x && dy && ws, DSNT_ERR_ARG, "dsnt_conv_wgrad_f16x3: null tensor");
        DSNT_REQUIRE(dw || !dbias, DSNT_ERR_ARG, "dsnt_conv_wgrad_f16x3: dbias without dw");
        DSNT_REQUIRE(!in_scale && !in_shift, DSNT_ERR_ARG, "dsnt_conv_wgrad_f16x3: the 4x4 / 16 -> 64 stem geometry takes a raw operand only");
        DSNT_REQUIRE(dsnt_aligned16(x) && dsnt_aligned16(dy) && dsnt_aligned16(ws) && (!dw || dsnt_aligned16(dw)),
                     DSNT_ERR_ALIGN, "dsnt_conv_wgrad_f16x3: tensors must be 16-byte aligned");
        hipStream_t st = (hipStream_t)stream;
        dsnt_stem4_wgrad_launch(x, dy, ws, a_bound, g_bound, g, st);
        if (dw) {
            const int CK = g->Cout * 16 * g->Cin;
            const int total = CK / 4 + (g->Cout + 3) / 4;
            DSNT_LAUNCH(wgrad_reduce_kernel, dim3((total + WRED_RC - 1) / WRED_RC), dim3(256), 0, st, ws, dw, dbias, s4, CK,
                        g->Cout, accumulate & 1);
        }
        DSNT_CHECK_LAUNCH("dsnt_conv_wgrad_f16x3");
    }
    // 3x3 / stride 1 convolutions: the halo kernel (wgrad3.hip) — every operand element staged once for all nine taps
    const Wg3Plan pl = dsnt_wg3_plan(g, wg3_share(accumulate));
    if (pl.ok) {
        if (int e = check_geom(g, "dsnt_conv_wgrad_f16x3")) return e;
        DSNT_REQUIRE(x && dy && ws, DSNT_ERR_ARG, "dsnt_conv_wgrad_f16x3: null tensor");
        DSNT_REQUIRE(dw || !dbias, DSNT_ERR_ARG, "dsnt_conv_wgrad_f16x3: dbias without dw");
        DSNT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), DSNT_ERR_ARG,
                     "dsnt_conv_wgrad_f16x3: in_scale/in_shift must be given together");
        DSNT_REQUIRE(dsnt_aligned16(x) && dsnt_aligned16(dy) && dsnt_aligned16(ws) && (!dw || dsnt_aligned16(dw)),
                     DSNT_ERR_ALIGN, "dsnt_conv_wgrad_f16x3: tensors must be 16-byte aligned");
        hipStream_t st = (hipStream_t)stream;
        dsnt_wg3_launch(pl, x, in_scale, in_shift, in_relu, dy, ws, a_bound, g_bound, g, st,
                        wg3_share(accumulate));
        if (dw) {
            const int CK = g->Cout * 9 * g->Cin;
            const int total = CK / 4 + (g->Cout + 3) / 4;
            DSNT_LAUNCH(wgrad_reduce_kernel, dim3((total + WRED_RC - 1) / WRED_RC), dim3(256), 0, st, ws, dw, dbias, pl.nslabs, CK,
                        g->Cout, accumulate & 1);
        }
        DSNT_CHECK_LAUNCH("dsnt_conv_wgrad_f16x3");
    }
    // 1x1 convolutions of >= 16384 rows: the transposition-free four-wave kernel (wgrad1.hip)
    const Wg1Plan p1 = dsnt_wg1_plan(g, (accumulate & DSNT_WGRAD_SHARE_CHIP) != 0);
    if (p1.ok) {
        if (int e = check_geom(g, "dsnt_conv_wgrad_f16x3")) return e;
        DSNT_REQUIRE(x && dy && ws, DSNT_ERR_ARG, "dsnt_conv_wgrad_f16x3: null tensor");
        DSNT_REQUIRE(dw || !dbias, DSNT_ERR_ARG, "dsnt_conv_wgrad_f16x3: dbias without dw");
        DSNT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), DSNT_ERR_ARG,
                     "dsnt_conv_wgrad_f16x3: in_scale/in_shift must be given together");
        DSNT_REQUIRE(dsnt_aligned16(x) && dsnt_aligned16(dy) && dsnt_aligned16(ws) && (!dw || dsnt_aligned16(dw)),
                     DSNT_ERR_ALIGN, "dsnt_conv_wgrad_f16x3: tensors must be 16-byte aligned");
        hipStream_t st = (hipStream_t)stream;
        dsnt_wg1_launch(p1, x, in_scale, in_shift, in_relu, dy, ws, a_bound, g_bound, g, st);
        if (dw) {
            const int CK = g->Cout * g->Cin;
            const int total = CK / 4 + (g->Cout + 3) / 4;
            DSNT_LAUNCH(wgrad_reduce_kernel, dim3((total + WRED_RC - 1) / WRED_RC), dim3(256), 0, st, ws, dw, dbias, p1.nsplits, CK,
                        g->Cout, accumulate & 1);
        }
        DSNT_CHECK_LAUNCH("dsnt_conv_wgrad_f16x3");
    }
    return conv_wgrad_impl(x, in_scale, in_shift, in_relu, dy, ws, dw, dbias, accumulate, g, stream, true, a_bound, g_bound);
}

// Plan of dsnt_conv_wgrad_f16x3 (the halo kernel cuts the pixels into its own slabs): number of slabs to reduce and
// workspace floats for a launch with these `accumulate` flags (DSNT_WGRAD_SHARE_CHIP changes the halo kernel's plan);
// equal to dsnt_conv_wgrad_splits / _ws_floats where the implicit-GEMM kernel runs.
extern "C" int dsnt_conv_wgrad_f16x3_splits(const dsnt_conv_geom* g, int accumulate) {
    if (!g) return 0;
    if (const int s4 = dsnt_stem4_wgrad_slabs(g)) return s4;
    const Wg3Plan pl = dsnt_wg3_plan(g, wg3_share(accumulate));
    if (pl.ok) return pl.nslabs;
    const Wg1Plan p1 = dsnt_wg1_plan(g, (accumulate & DSNT_WGRAD_SHARE_CHIP) != 0);
    return p1.ok ? p1.nsplits : dsnt_conv_wgrad_splits(g);
}
extern "C" int64_t dsnt_conv_wgrad_f16x3_ws_floats(const dsnt_conv_geom* g, int accumulate) {
    if (!g) return 0;
    if (const int s4 = dsnt_stem4_wgrad_slabs(g)) return (int64_t)s4 * g->Cout * (16 * g->Cin) + (int64_t)s4 * g->Cout;
    const Wg3Plan pl = dsnt_wg3_plan(g, wg3_share(accumulate));
    if (pl.ok) return (int64_t)pl.nslabs * g->Cout * (9 * g->Cin) + (int64_t)pl.nslabs * g->Cout;
    const Wg1Plan p1 = dsnt_wg1_plan(g, (accumulate & DSNT_WGRAD_SHARE_CHIP) != 0);
    if (p1.ok) return (int64_t)p1.nsplits * g->Cout * g->Cin + (int64_t)p1.nsplits * g->Cout;
    return wgrad_ws_floats_plain(g);
}
extern "C" int dsnt_conv_wgrad_halo_ok(const dsnt_conv_geom* g) { return g ? dsnt_wg3_plan(g, 0).ok : 0; }

static int conv_wgrad_impl(const float* x, const float* in_scale, const float* in_shift, int in_relu,
                           const float* dy, float* ws, float* dw, float* dbias, int accumulate,
                           const dsnt_conv_geom* g, void* stream, bool bf16x6, const float* a_bound,
                           const float* g_bound) {
    if (int e = check_geom(g, "dsnt_conv_wgrad")) return e;
    DSNT_REQUIRE(x && dy && ws, DSNT_ERR_ARG, "dsnt_conv_wgrad: null tensor");
    DSNT_REQUIRE(dw || !dbias, DSNT_ERR_ARG, "dsnt_conv_wgrad: dbias without dw");
    DSNT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), DSNT_ERR_ARG,
                 "dsnt_conv_wgrad: in_scale/in_shift must be given together");
    DSNT_REQUIRE(g->Cout % 4 == 0, DSNT_ERR_ALIGN, "dsnt_conv_wgrad: Cout=%d must be a multiple of 4",
                 g->Cout);
    DSNT_REQUIRE(dsnt_aligned16(x) && dsnt_aligned16(dy) && dsnt_aligned16(ws) && (!dw || dsnt_aligned16(dw)),
                 DSNT_ERR_ALIGN, "dsnt_conv_wgrad: tensors must be 16-byte aligned");
    WgradP p;
    wgrad_fill(p, x, in_scale, in_shift, in_relu, dy, ws, g, a_bound, g_bound);
    hipStream_t st = (hipStream_t)stream;
    const int grid = p.ktiles * p.ntiles * p.splits;
    // DSNT_WGRAD_SHARE_CHIP (bit 1 of `accumulate`): the launch runs beside a dependency chain on another stream.  Its
    // workgroups live as long as the kernel (one wave of ~2 per CU), and two of them fill a CU's registers: the chain's
    // small kernels (a 16-workgroup BatchNorm finalise) then wait for the whole weight gradient to end — 150 us seen.
    // Asking for more than half of the LDS keeps it to ONE workgroup per CU and the other half of every CU free.
    const bool share = (accumulate & DSNT_WGRAD_SHARE_CHIP) != 0;
    accumulate &= 1;
    const int share_lds = 88 * 1024;
    if (bf16x6) {
        const int lds_full = 2 * 2 * 3 * 128 * PITCH6 * 2;
        const int lds = share ? share_lds : lds_full;
        DSNT_SET_MAX_LDS((conv_wgrad_bf16x6_kernel<true, false>), share_lds);
        DSNT_SET_MAX_LDS((conv_wgrad_bf16x6_kernel<false, false>), share_lds);
        if (a_bound) {              // fp16x3 (role-split kernel; its two fp16 planes need 2/3 of the LDS)
            DSNT_SET_MAX_LDS((conv_wgrad_bf16x6_kernel<true, true>), share_lds);
            DSNT_SET_MAX_LDS((conv_wgrad_bf16x6_kernel<false, true>), share_lds);
            const int lds16 = share ? share_lds : 2 * 2 * 2 * 128 * PITCH6 * 2;
            if (in_scale) DSNT_LAUNCH((conv_wgrad_bf16x6_kernel<true, true>), dim3(grid), dim3(512), lds16, st, p);
            else DSNT_LAUNCH((conv_wgrad_bf16x6_kernel<false, true>), dim3(grid), dim3(512), lds16, st, p);
        } else if (in_scale) DSNT_LAUNCH((conv_wgrad_bf16x6_kernel<true, false>), dim3(grid), dim3(512), lds, st, p);
        else DSNT_LAUNCH((conv_wgrad_bf16x6_kernel<false, false>), dim3(grid), dim3(512), lds, st, p);
    } else if (in_scale)
        DSNT_LAUNCH(conv_wgrad_kernel<true>, dim3(grid), dim3(256), 0, st, p);
    else
        DSNT_LAUNCH(conv_wgrad_kernel<false>, dim3(grid), dim3(256), 0, st, p);
    if (dw) {       // dw == nullptr: slabs only, the caller reduces later (dsnt_wgrad_reduce_all)
        const int CK = p.Cout * p.K;
        const int total = CK / 4 + (p.Cout + 3) / 4;
        DSNT_LAUNCH(wgrad_reduce_kernel, dim3((total + WRED_RC - 1) / WRED_RC), dim3(256), 0, st, ws, dw, dbias,
                           p.splits, CK, p.Cout, accumulate);
    }
    DSNT_CHECK_LAUNCH("dsnt_conv_wgrad");
}

// =====================================================================================================================
// The persistent low-resolution stage (stage.h): one launch for a run of small dependent launches of one lane.
#include "ew_bodies.h"

// Chip-wide barrier between two recorded launches: one lane per workgroup adds to the stage's counter with an agent-scope release
// (the workgroup's stores are in L2 behind the __syncthreads and leave it with the release) and polls it with agent-scope acquire
// loads (s_sleep between polls) until every workgroup has arrived; profiles/r02_grid_barrier.txt: 0.9 / 2.4 / 3.9 us for 16 / 64 /
// 128 workgroups against a ~5 us launch boundary in the traced step.  The spin is BOUNDED: a workgroup that gives up raises the
// stage's error word and every workgroup leaves the kernel — a wrong result the host can see, never a hung device.
__device__ __forceinline__ bool stage_barrier(unsigned* sync, const unsigned want, int* s_abort) {
    // EVERY wave releases at agent scope before the workgroup barrier: __syncthreads() only fences LDS (s_waitcnt lgkmcnt), so
    // without this a wave can sit in the barrier with global stores still in flight, lane 0 of the workgroup then announces the
    // launch as done, and a workgroup on another XCD reads the old bytes — seen as a 3 % error in one layer's gradients of the
    // first hg8 step (tests/test_fallback_gpu.py under DSNT_STAGE=1; a kernel boundary waits for every store, a barrier must too)
    // (workgroup scope is enough per wave — s_waitcnt vmcnt(0): its stores are in the XCD's L2 —; ONE agent-scope release, lane 0's
    // below, then writes that L2 back.  Every wave releasing at agent scope is correct too and costs the step 2.5 ms more.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0), whatever the fence above was lowered to
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned guard = 0;
        // (relaxed polls and ONE acquire fence behind them: an acquire load per poll invalidates this CU's vector cache — and the
        // non-local lines of the XCD's L2 — every time, under the kernels of the other lanes that share them)
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            if ((++guard & 1023u) == 0 &&
                (guard >= (1u << 21) || __hip_atomic_load(sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                __hip_atomic_store(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_abort = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (one invalidate of this CU's vector cache and the XCD's non-local L2 lines: the waves
    }                                                       //  of the workgroup are parked at the barrier below and share that cache)
    __syncthreads();
    __builtin_amdgcn_s_dcache_inv();          // (scalar loads of the next launch's operands must not hit lines read before the barrier)
    return *s_abort == 0;
}

__global__ __launch_bounds__(DSNT_STAGE_NT) void dsnt_stage_kernel(const DsntStageOp* __restrict__ ops, const int nops, unsigned* sync) {
    __shared__ __attribute__((aligned(16))) float part[8][32][33];          // the K-split kernel's tile sums; the head of it serves the others
    __shared__ int s_abort;
    const int G = gridDim.x, wg = blockIdx.x;
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();
    for (int k = 0; k < nops; ++k) {
        const DsntStageOp& op = ops[k];
        const int code = op.code, gx = op.gx, gy = op.gy;
        const int nvb = gx * gy;
        bool first = true;
        for (int vb = wg; vb < nvb; vb += G) {
            __syncthreads();                                              // the previous virtual workgroup is done with the LDS
            switch (code) {
            case DSNT_ST_KSPLIT_PRO:
                conv_ksplit_body<true, 16>(*reinterpret_cast<const ConvP*>(op.params), vb, part, first); break;
            case DSNT_ST_KSPLIT:
                conv_ksplit_body<false, 16>(*reinterpret_cast<const ConvP*>(op.params), vb, part, first); break;
            case DSNT_ST_APPLY_FIXED:
                bn_act_bwd_apply_body<true>(*reinterpret_cast<const BnApplyP*>(op.params), vb, nvb, reinterpret_cast<double*>(&part[0][0][0]), first); break;
            case DSNT_ST_APPLY:
                bn_act_bwd_apply_body<false>(*reinterpret_cast<const BnApplyP*>(op.params), vb, nvb, reinterpret_cast<double*>(&part[0][0][0]), first); break;
            case DSNT_ST_TILE_POOL:
                tile_op_stats_body<0>(*reinterpret_cast<const TileOpP*>(op.params), vb % gx, vb / gx, gy, &part[0][0][0]); break;
            case DSNT_ST_TILE_UPADD:
                tile_op_stats_body<1>(*reinterpret_cast<const TileOpP*>(op.params), vb % gx, vb / gx, gy, &part[0][0][0]); break;
            case DSNT_ST_POOL_BWD:
                maxpool2_bwd_body(*reinterpret_cast<const PoolBwdP*>(op.params), vb, nvb); break;
            case DSNT_ST_UP_BWD:
                upsample2_bwd_body(*reinterpret_cast<const UpBwdP*>(op.params), vb, nvb); break;
            case DSNT_ST_FIN_FWD:
                bn_finalize_body<0, DSNT_STAGE_NT>(*reinterpret_cast<const BnFinP*>(op.params), vb,
                                                   reinterpret_cast<double*>(&part[0][0][0]), reinterpret_cast<double*>(&part[0][0][0]) + 256); break;
            case DSNT_ST_FIN_BWD:
                bn_finalize_body<1, DSNT_STAGE_NT>(*reinterpret_cast<const BnFinP*>(op.params), vb,
                                                   reinterpret_cast<double*>(&part[0][0][0]), reinterpret_cast<double*>(&part[0][0][0]) + 256); break;
            default: break;
            }
            first = false;
        }
        if (k + 1 < nops && !stage_barrier(sync, (unsigned)(k + 1) * (unsigned)G, &s_abort)) break;
    }
    // the last workgroup to leave hands the counters back as it found them (every workgroup is past its last poll by then)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned d = __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (d == (unsigned)G - 1u) {
            __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

void dsnt_stage_launch(const DsntStageOp* ops_dev, int nops, unsigned* sync_dev, int grid, hipStream_t st) {
    hipLaunchKernelGGL(dsnt_stage_kernel, dim3(grid), dim3(DSNT_STAGE_NT), 0, st, ops_dev, nops, sync_dev);
}
