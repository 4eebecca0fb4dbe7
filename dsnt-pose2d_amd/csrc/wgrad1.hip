// Weight gradient of the 1x1 convolutions (hourglass.py:20-25: conv1 / conv3 of every Bottleneck; :104-148 the lin
// convolutions) on the fp16 matrix cores, fp16x3 split:   dW[n][c] = sum over pixels  A[pixel][c] * dY[pixel][n].
//
// HBM-bound by a wide margin (128 -> 256 @64x64, batch 32: 201 MB per launch against 8.6 GFLOP), yet the implicit-GEMM
// kernel (conv.hip) runs these launches on the SIMD's issue port: 12 VALU instructions per MFMA, because the MFMA
// contracts over PIXELS while both tensors are channel-major — every element is transformed, split and transposed by
// VALU work, and a 128 x 128 output tile re-stages its operands for every other tile of the same rows.  Here, with the
// machinery of the 3x3 halo kernel (wgrad3.hip):
//   * a workgroup owns (up to) the WHOLE weight matrix — 128 x 256 or 256 x 128 channels — for its range of pixels: each
//     element of A and dY is staged exactly once per launch;
//   * staging is transposition-free: pixel-major [pixel][32 channels] fp16 planes in LDS (a float4 of four channels ->
//     one 8-byte store per plane), the MFMA operands are read with ds_read_b64_tr_b16;
//   * FOUR waves (2 x 2 over the tile, 8 accumulator tiles each) and 50 KB of LDS: one workgroup per CU fills the HBM
//     pipe (three 16-pixel stages of loads in flight per thread) and leaves 40 % of every CU's registers and most of its
//     LDS to the dependency chain's kernels on the other streams — these launches run beside it (DSNT_WGRAD_SHARE_CHIP).
// Slabs ws[split][Cout][K] (+ [split][Cout] bias partials) as every other weight-gradient kernel writes them.
#include "wgrad3.h"
#include "conv_split.h"
#include <stdlib.h>

typedef short w1_s16x4 __attribute__((ext_vector_type(4)));
typedef short w1_s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned w1_u32x4 __attribute__((ext_vector_type(4)));
#define W1_LDS __attribute__((address_space(3)))

struct Wg1P {
    const float* x; const float* in_scale; const float* in_shift; const float* dy;
    float* ws;
    const float* a_bound; const float* g_bound;
    int in_relu;
    int M, Cin, Cout;
    int kchunks, nchunks, nsplits, rows_per_split;
};

__device__ __forceinline__ f16x8 w1_tr_frag(W1_LDS unsigned char* base, int off) {
    const w1_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((W1_LDS w1_s16x4*)(base + off));
    const w1_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((W1_LDS w1_s16x4*)(base + off + 4 * 64));
    const w1_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f16x8, v);
}

// wave tile = KTW x NTW tiles of 32 x 32 (input channels x output channels); the workgroup's four waves sit 2 x 2
template <int KTW, int NTW>
__global__ __launch_bounds__(256, 1) void wgrad1_kernel(Wg1P p) {
    constexpr int CK = 64 * KTW, CN = 64 * NTW;            // channels of A / dY per workgroup
    constexpr int SUB = 16 * 64;                           // bytes of one [16 pixels][32 channels] fp16 sub-tile
    constexpr int A_PL = (CK / 32) * SUB, G_PL = (CN / 32) * SUB;      // one plane of one stage
    constexpr int STAGE = 2 * (A_PL + G_PL);               // two planes of both operands
    constexpr int NA = CK / 64, NG = CN / 64;              // float4 units per thread and stage (256 threads x 16 pixels)
    constexpr int DEPTH = 2;                               // stages of loads in flight beyond the one being stored
    extern __shared__ __attribute__((aligned(16))) unsigned char w1_smem[];       // [2 buffers][STAGE]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave & 1, wn = wave >> 1;
    int bid;
    xcd_remap(blockIdx.x, gridDim.x, bid);
    const int kch = bid % p.kchunks; bid /= p.kchunks;
    const int nch = bid % p.nchunks; bid /= p.nchunks;
    const int split = bid;
    const int c0 = kch * CK, n0 = nch * CN;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nstages = (m_end - m_begin) / 16;            // rows_per_split and M are multiples of 16

    const float sa = pow2_scale(bound64(p.a_bound)), sg = pow2_scale(bound64(p.g_bound));
    const float relu_lo = p.in_relu ? 0.f : -__builtin_inff();
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.M * p.Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy), 0, (int)((size_t)p.M * p.Cout * 4u), 0x00020000);

    // ---- staging role: unit u = tid + 256 i -> (32-channel sub-tile, pixel, 4-channel chunk)
    const int ch4 = tid & 7, px = (tid >> 3) & 15, sub0 = tid >> 7;        // sub-tile = sub0 + 2 i
    unsigned aoff[NA], goff[NG];
    // BatchNorm scale / shift of the workgroup's channels (x operand scale) live in LDS behind the two stage buffers:
    // 8 (16) float4 per thread would otherwise sit in registers for the whole kernel
    float* SS = reinterpret_cast<float*>(w1_smem + 2 * STAGE);          // [2][CK]
    for (int c = tid; c < CK; c += 256) {
        SS[c] = p.in_scale ? p.in_scale[c0 + c] * sa : sa;
        SS[CK + c] = p.in_scale ? p.in_shift[c0 + c] * sa : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NA; ++i) aoff[i] = (unsigned)(px * p.Cin + c0 + 32 * (sub0 + 2 * i) + 4 * ch4) * 4u;
#pragma unroll
    for (int i = 0; i < NG; ++i) goff[i] = (unsigned)(px * p.Cout + n0 + 32 * (sub0 + 2 * i) + 4 * ch4) * 4u;
    const unsigned lds_unit = (unsigned)(sub0 * SUB + px * 64 + ch4 * 8);     // + 2 i SUB; dY region behind A's two planes

    struct Raw { w1_u32x4 a[NA], g[NG]; };
    auto issue = [&](Raw& R, const int s) {                 // loads of stage s (past the end: nothing, zeros)
        const bool ok = s < nstages;
        const unsigned ra = (unsigned)(m_begin + 16 * s) * (unsigned)p.Cin * 4u;
        const unsigned rg = (unsigned)(m_begin + 16 * s) * (unsigned)p.Cout * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i) R.a[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? aoff[i] + ra : 0xF0000000u, 0, 0);
#pragma unroll
        for (int i = 0; i < NG; ++i) R.g[i] = __builtin_amdgcn_raw_buffer_load_b128(gr, ok ? goff[i] + rg : 0xF0000000u, 0, 0);
    };
    float4 bs[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) bs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto store = [&](const Raw& R, const int buf) {         // BN + ReLU (A) / scale (dY), exact split, 8-byte stores
        unsigned char* base = w1_smem + buf * STAGE + lds_unit;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            float4 v = make_float4(__uint_as_float(R.a[i].x), __uint_as_float(R.a[i].y), __uint_as_float(R.a[i].z), __uint_as_float(R.a[i].w));
            const float4 sc = *reinterpret_cast<const float4*>(SS + 32 * (sub0 + 2 * i) + 4 * ch4);
            const float4 sh = *reinterpret_cast<const float4*>(SS + CK + 32 * (sub0 + 2 * i) + 4 * ch4);
            v.x = fmaxf(fmaf(v.x, sc.x, sh.x), relu_lo); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), relu_lo);
            v.z = fmaxf(fmaf(v.z, sc.z, sh.z), relu_lo); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), relu_lo);
            uint2 q1, q2;
            split4h(v, q1, q2);
            *reinterpret_cast<uint2*>(base + 2 * i * SUB) = q1;
            *reinterpret_cast<uint2*>(base + 2 * i * SUB + A_PL) = q2;
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const float4 r = make_float4(__uint_as_float(R.g[i].x), __uint_as_float(R.g[i].y), __uint_as_float(R.g[i].z), __uint_as_float(R.g[i].w));
            bs[i].x += r.x; bs[i].y += r.y; bs[i].z += r.z; bs[i].w += r.w;          // bias partial (column sums of dY)
            uint2 q1, q2;
            split4h(make_float4(r.x * sg, r.y * sg, r.z * sg, r.w * sg), q1, q2);
            *reinterpret_cast<uint2*>(base + 2 * A_PL + 2 * i * SUB) = q1;
            *reinterpret_cast<uint2*>(base + 2 * A_PL + 2 * i * SUB + G_PL) = q2;
        }
    };

    // ---- matrix role: transposed-read address of this lane inside a sub-tile (cdna guide T10; wgrad3.hip)
    const int lh = lane >> 5, cb = (lane >> 4) & 1, q4 = (lane >> 2) & 3, pp = lane & 3;
    const unsigned lane_t = (unsigned)((8 * lh + q4) * 64 + (16 * cb + 4 * pp) * 2);
    W1_LDS unsigned char* lds0 = (W1_LDS unsigned char*)w1_smem;

    f32x16 acc[KTW][NTW];
#pragma unroll
    for (int a = 0; a < KTW; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // ---- pipeline: stage s is consumed from buffer s & 1 while stage s+1 is transformed into the other buffer and the
    // loads of stages s+2 .. s+1+DEPTH travel
    Raw R[DEPTH + 1];
    issue(R[0], 0);
#pragma unroll
    for (int d = 1; d <= DEPTH; ++d) issue(R[d], d);
    store(R[0], 0);
    issue(R[0], DEPTH + 1);
    __syncthreads();
    // (R[(s+1) % (DEPTH+1)] holds stage s+1 at the top of iteration s; the slot freed by the store takes stage s+2+DEPTH)
    auto step = [&](const int s, Raw& next, const int buf) {
        if (s + 1 < nstages) store(next, buf ^ 1);
        issue(next, s + 2 + DEPTH);
        __builtin_amdgcn_sched_barrier(0);
        W1_LDS unsigned char* ab = lds0 + (unsigned)(buf * STAGE) + lane_t;
        // the operand with fewer tiles per wave is held in registers, the other streams past it
        if (NTW <= KTW) {
            f16x8 g1[NTW], g2[NTW];
#pragma unroll
            for (int b = 0; b < NTW; ++b) {
                const int off = 2 * A_PL + (wn * NTW + b) * SUB;
                g1[b] = w1_tr_frag(ab, off);
                g2[b] = w1_tr_frag(ab, off + G_PL);
            }
#pragma unroll
            for (int a = 0; a < KTW; ++a) {
                const int off = (wk * KTW + a) * SUB;
                const f16x8 a1 = w1_tr_frag(ab, off), a2 = w1_tr_frag(ab, off + A_PL);
#pragma unroll
                for (int b = 0; b < NTW; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, g1[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, g2[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, g1[b], acc[a][b], 0, 0, 0);
                }
            }
        } else {
            f16x8 a1[KTW], a2[KTW];
#pragma unroll
            for (int a = 0; a < KTW; ++a) {
                const int off = (wk * KTW + a) * SUB;
                a1[a] = w1_tr_frag(ab, off);
                a2[a] = w1_tr_frag(ab, off + A_PL);
            }
#pragma unroll
            for (int b = 0; b < NTW; ++b) {
                const int off = 2 * A_PL + (wn * NTW + b) * SUB;
                const f16x8 g1 = w1_tr_frag(ab, off), g2 = w1_tr_frag(ab, off + G_PL);
#pragma unroll
                for (int a = 0; a < KTW; ++a) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[a], g1, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[a], g2, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[a], g1, acc[a][b], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    };
    int s = 0;
    for (; s + DEPTH < nstages; s += DEPTH + 1) {           // unrolled over the register slots (no dynamic indexing)
        step(s + 0, R[1], 0 ^ (s & 1));
        step(s + 1, R[2], 1 ^ (s & 1));
        step(s + 2, R[0], 0 ^ (s & 1));
    }
    static_assert(DEPTH == 2, "the loop above is written for three register slots");
    if (s < nstages) { step(s, R[1], s & 1); ++s; }
    if (s < nstages) { step(s, R[2], s & 1); ++s; }

    // ---- slab store: ws[split][n][c], D row = input channel (registers, 4 consecutive), D column = n (lane)
    const float osc = 1.f / (sa * sg);
    const int lr = lane & 31;
    float* slab = p.ws + (size_t)split * p.Cout * p.Cin;
#pragma unroll
    for (int b = 0; b < NTW; ++b) {
        const int n = n0 + (wn * NTW + b) * 32 + lr;
#pragma unroll
        for (int a = 0; a < KTW; ++a) {
            float* o = slab + (size_t)n * p.Cin + c0 + (wk * KTW + a) * 32 + 4 * lh;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
                *reinterpret_cast<float4*>(o + 8 * qq) =
                    make_float4(acc[a][b][4 * qq + 0] * osc, acc[a][b][4 * qq + 1] * osc, acc[a][b][4 * qq + 2] * osc,
                                acc[a][b][4 * qq + 3] * osc);
        }
    }
    // bias partial of this split: the 16 pixel-threads of a (sub-tile, chunk) add up through LDS in pixel order
    if (kch == 0) {
        float4* red = reinterpret_cast<float4*>(w1_smem);          // [16 pixels][CN / 4] float4 (the loop ended on a barrier)
#pragma unroll
        for (int i = 0; i < NG; ++i) red[px * (CN / 4) + (sub0 + 2 * i) * 8 + ch4] = bs[i];
        __syncthreads();
        if (tid < CN / 4) {
            float4 t = red[tid];
            for (int j = 1; j < 16; ++j) {
                const float4 v = red[j * (CN / 4) + tid];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            *reinterpret_cast<float4*>(p.ws + (size_t)p.nsplits * p.Cout * p.Cin + (size_t)split * p.Cout + n0 + tid * 4) = t;
        }
    }
}

static int w1_enabled = -1;

Wg1Plan dsnt_wg1_plan(const dsnt_conv_geom* g, bool share) {
    Wg1Plan pl;
    memset(&pl, 0, sizeof(pl));
    if (w1_enabled < 0) {
        w1_enabled = dsnt_kernel_off("wgrad1") ? 0 : 1;
    }
    if (!w1_enabled || !g) return pl;
    if (!(g->R == 1 && g->S == 1 && g->stride == 1 && g->pad == 0 && g->Ho == g->H && g->Wo == g->W)) return pl;
    const long M = (long)g->N * g->H * g->W;
    if (M < 16384 || M % 16 != 0) return pl;                       // smaller ones: the grouped launch (engine)
    if ((size_t)M * g->Cin * 4u >= (1ull << 31) || (size_t)M * g->Cout * 4u >= (1ull << 31)) return pl;
    // workgroup tile (input x output channels): the whole matrix where it fits 8 accumulator tiles per wave
    int ck, cn;
    if (g->Cin % 256 == 0 && g->Cout % 128 == 0 && g->Cout % 256 != 0) { ck = 256; cn = 128; }
    else if (g->Cin % 128 == 0 && g->Cout % 256 == 0) { ck = 128; cn = 256; }
    else if (g->Cin % 128 == 0 && g->Cout % 128 == 0) { ck = 128; cn = 128; }
    else if (g->Cin % 64 == 0 && g->Cout % 128 == 0) { ck = 64; cn = 128; }
    else if (g->Cin % 128 == 0 && g->Cout % 64 == 0) { ck = 128; cn = 64; }
    else if (g->Cin % 64 == 0 && g->Cout % 64 == 0) { ck = 64; cn = 64; }
    else return pl;
    pl.ck = ck; pl.cn = cn;
    pl.kchunks = g->Cin / ck;
    pl.nchunks = g->Cout / cn;
    // one workgroup per CU; at least 8 stages of 16 pixels per workgroup.  A launch that shares the chip with the
    // dependency chain takes HALF the CUs: slab bytes (written here, flushed at the kernel boundary, read by the reduction)
    // cost the step more than this launch's own duration, which the weight-gradient lane has slack for
    // (measured, hg2 batch 32: 256 slabs per 1x1 convolution +0.15 ms/step over 128)
    long sp = (share ? 128 : 256) / (pl.kchunks * pl.nchunks);
    if (sp < 1) sp = 1;
    const long max_sp = M / 128;
    if (sp > max_sp) sp = max_sp;
    long rows = (M + sp - 1) / sp;
    rows = (rows + 15) / 16 * 16;
    pl.nsplits = (int)((M + rows - 1) / rows);
    pl.rows_per_split = (int)rows;
    pl.blocks = pl.kchunks * pl.nchunks * pl.nsplits;
    pl.lds = 2 * 2 * (ck + cn) * 16 * 2 + 2 * ck * 4;        // two stage buffers + the BatchNorm vectors
    pl.ok = 1;
    return pl;
}

template <int KTW, int NTW>
static void w1_launch_cfg(const Wg1Plan& pl, const Wg1P& p, hipStream_t st) {
    DSNT_SET_MAX_LDS((wgrad1_kernel<KTW, NTW>), pl.lds);
    DSNT_LAUNCH((wgrad1_kernel<KTW, NTW>), dim3(pl.blocks), dim3(256), pl.lds, st, p);
}

void dsnt_wg1_launch(const Wg1Plan& pl, const float* x, const float* in_scale, const float* in_shift, int in_relu,
                     const float* dy, float* ws, const float* a_bound, const float* g_bound, const dsnt_conv_geom* g,
                     hipStream_t st) {
    Wg1P p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.in_scale = in_scale; p.in_shift = in_shift; p.dy = dy; p.ws = ws;
    p.a_bound = a_bound; p.g_bound = g_bound; p.in_relu = in_scale ? in_relu : 0;
    p.M = g->N * g->H * g->W; p.Cin = g->Cin; p.Cout = g->Cout;
    p.kchunks = pl.kchunks; p.nchunks = pl.nchunks; p.nsplits = pl.nsplits; p.rows_per_split = pl.rows_per_split;
    const int kt = pl.ck / 64, nt = pl.cn / 64;
    if (kt == 4 && nt == 2) w1_launch_cfg<4, 2>(pl, p, st);
    else if (kt == 2 && nt == 4) w1_launch_cfg<2, 4>(pl, p, st);
    else if (kt == 2 && nt == 2) w1_launch_cfg<2, 2>(pl, p, st);
    else if (kt == 1 && nt == 2) w1_launch_cfg<1, 2>(pl, p, st);
    else if (kt == 2 && nt == 1) w1_launch_cfg<2, 1>(pl, p, st);
    else w1_launch_cfg<1, 1>(pl, p, st);
}
