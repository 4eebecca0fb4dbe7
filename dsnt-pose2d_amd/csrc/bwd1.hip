// The whole backward of a 1x1 convolution of a pre-activation Bottleneck in ONE pass over its tensors
// (hourglass.py:20-25,33-43: conv1 / conv3 with the BatchNorm + ReLU in front of them; what autograd runs as cuDNN
// backward-data + backward-filter + the BatchNorm backward of the layer behind):
//
//     dX[m][c]  = sum_n dY[m][n] W[n][c]          -> masked by the ReLU of bn(x), written as dz, with the two per-channel
//                                                    sums of the BatchNorm backward of bn(x)
//     dW[n][c]  = sum_m dY[m][n] act(x)[m][c]     -> one slab per workgroup (+ the bias partial sum_m dY[m][n])
//
// and, when the convolution's output y feeds a BatchNorm whose backward has been reduced already (coef), dY itself is
// formed in registers from dz and y:   dY = scale (dz - coef0 - (y - mean) invstd coef1)   — the bn_act_bwd_apply pass
// of that layer and its 12 bytes per element never happen.
//
// Until round 4 these were three launches (apply, gemm1 data gradient, wgrad1) that read dY three times and x twice:
// 737 MB for a 256 -> 128 convolution at 64 x 64, batch 32; here 402 MB, every tensor once.
//
// Shape of the kernel.  The two products contract over different indices (n for dX, pixels for dW), so one of them needs
// dY transposed; the other operand of the weight gradient, act(x), is needed in the SAME layout in which x is needed
// for the ReLU mask of dX.  So:
//   * a wave OWNS 16 CW output columns c of dX (8 waves x 16 CW = Cin) for ALL pixels of the workgroup: its slice of W
//     stays resident (hi plane in registers, lo plane in LDS), its slice of dW (all n x its columns) stays in
//     accumulators for the whole launch, and x is loaded exactly once, in the C layout of the dX tile (column on the
//     lane, 4 pixel rows in registers: 64-byte runs) — where the mask needs it and where, after BN + ReLU + split, it
//     IS the B operand of the dW product (an accumulator-layout tile is a valid MFMA operand when the contraction
//     runs over its rows: cdna guide §3);
//   * dY is what all waves share: stages of 32 pixels are loaded by all 512 threads (16-byte loads, one stage in
//     registers while the previous one is consumed), transformed (the folded BatchNorm backward), split into two fp16
//     planes and written pixel-major into a double-buffered LDS image; every wave reads it by rows (ds_read_b128) as the
//     A operand of dX and transposed (ds_read_b64_tr_b16) as the A operand of dW.  The pitch (2 NN + 32 bytes) makes
//     both conflict-free.
// v_mfma_f32_16x16x32_f16 throughout: 16-column ownership keeps W + dW at 96-128 registers per lane for both shapes
// (n = 128, c = 256 and n = 256, c = 128).  One barrier per 32 pixels.  HBM-bound: 96 MFMAs per wave and stage are a
// third of the stage's memory time.
// Variants of the same body: four-wave workgroups for 64 input channels (the 128 x 128 level), two column chunks on
// gridDim.y for 256 -> 256 (both read dY, the second from L2: workgroups x and x + gridDim.x share an XCD when gridDim.x
// is a multiple of 8), and RAW = a convolution whose input is not seen through a BatchNorm (the projection shortcuts
// hourglass.py:44-48 and the `fc` convolutions :146-153): act(x) = x, dX is written (or accumulated) as it is.
#include "bwd1.h"
#include <stdlib.h>

typedef short b1_s16x4 __attribute__((ext_vector_type(4)));
typedef short b1_s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned b1_u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define B1_LDS __attribute__((address_space(3)))

struct Bwd1P {
    const float* dz;            // MODE 0: dz of the BatchNorm behind y; MODE 1: dL/dy itself
    const float* yo;            // MODE 0: y
    const float* y_scale; const float* y_mean; const float* y_invstd; const float* y_coef;      // MODE 0
    const float* g_bound;       // >= max |dY|
    const float* x; const float* in_scale; const float* in_shift; const float* in_mean; const float* in_invstd;
    int in_relu;
    const float* a_bound;       // >= max |act(x)|
    const unsigned short* wq; long wq_stride; const float* w_bound;     // data-gradient weights [c][n], two fp16 planes
    float* dzx;                 // [M][Cin]: dz (PRO) or dL/dx (RAW; += with `accumulate`)
    float* stats;               // [workgroups][2][Cin] (PRO)
    float* ws;                  // [workgroups][NN][Cin] slabs, then [workgroups][NN] bias partials
    unsigned* dzx_amax;         // optional bound slot of |dzx|
    int M, Cin, nstages, spw, nwg;   // 32-pixel stages in all / per workgroup; nwg = gridDim.x = slabs
    int accumulate;             // RAW: dzx += (the tensor already holds other gradient contributions)
};

__device__ __forceinline__ f16x8 b1_tr_frag(B1_LDS unsigned char* a0, B1_LDS unsigned char* a1) {
    const b1_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((B1_LDS b1_s16x4*)a0);
    const b1_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((B1_LDS b1_s16x4*)a1);
    const b1_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f16x8, v);
}

// NN = output channels of the convolution (n: the contraction of dX), CW = 16-column tiles per wave, NWV = waves per workgroup
// (16 CW NWV columns of Cin per workgroup; blockIdx.y = column chunk), MODE 0: dY from (dz, y) through the folded BatchNorm
// backward, 1: dY given; RAW: x is the operand itself (no BatchNorm in front of the convolution)
template <int NN, int CW, int NWV, int MODE, bool RAW>
__global__ __launch_bounds__(64 * NWV, 2) void bwd1_kernel(Bwd1P p) {
    constexpr int CC = 16 * CW * NWV;           // columns of this workgroup
    constexpr int NTHR = 64 * NWV;
    constexpr int KS = NN / 32;                 // K-steps of dX
    constexpr int NT = NN / 16;                 // n-tiles of dW
    constexpr int PITCH = NN * 2 + 32;          // bytes per pixel row of one plane (and per column row of the W image)
    constexpr int PL = 32 * PITCH;
    constexpr int IMG = 2 * PL;
    constexpr int WLO = CC * PITCH;
    constexpr int N4 = NN / 4;                  // float4 units per pixel row
    constexpr int U = 32 * N4 / NTHR;           // units per thread and stage
    constexpr int RSTEP = NTHR / N4;            // pixel rows between a thread's units
    static_assert(U >= 1 && U * NTHR == 32 * N4, "staging units per thread");
    const unsigned OOB = 0xF0000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char b1_smem[];
    unsigned char* img = b1_smem;                                   // [2 buffers][2 planes][32 pixels][PITCH]
    unsigned char* wlo = b1_smem + 2 * IMG;                         // [CC][PITCH]: lo plane of W
    float* vec = reinterpret_cast<float*>(wlo + WLO);               // MODE 0: [5][NN] scale, mean, invstd, coef0, coef1

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, lg = lane >> 4;
    const int wg = blockIdx.x;
    const int col0 = blockIdx.y * CC;           // first column of this workgroup's chunk
    const int ld = p.Cin;                       // row pitch of x / dzx / the slabs
    const int s0 = wg * p.spw;
    const int s1 = min(p.nstages, s0 + p.spw);

    const float sa = pow2_scale(bound64(p.a_bound)), sw = pow2_scale(bound64(p.w_bound)), sg = pow2_scale(bound64(p.g_bound));
    const float osc_x = 1.f / (sg * sw), osc_w = 1.f / (sg * sa);

    // ---- dY staging role: unit i of this thread = pixel row (tid / N4) + RSTEP i, channels 4 n4 .. 4 n4 + 3
    const int n4 = tid % N4, prow = tid / N4;
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dz), 0, (int)((size_t)p.M * NN * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(MODE == 0 ? p.yo : p.dz), 0, (int)((size_t)p.M * NN * 4u), 0x00020000);
    struct Raw { b1_u32x4 z[U]; b1_u32x4 y[MODE == 0 ? U : 1]; };
    // byte offset of stage s of a tensor with `stage_bytes` per stage — or, past this workgroup's range, an offset the
    // buffer's range check refuses (the load returns zeros).  A SCALAR select, laundered: as a select of the lane offsets
    // hipcc turns it into control flow around the loads and drains vmcnt(0) at the join
    auto stage_off = [&](const int s, const unsigned stage_bytes) {
        unsigned o = s < s1 ? (unsigned)s * stage_bytes : OOB;
        asm volatile("" : "+s"(o));
        return o;
    };
    const unsigned zlane = (unsigned)((prow * NN + 4 * n4) * 4);
    auto issue = [&](Raw& R, const int s) {
        const unsigned so = stage_off(s, 32u * NN * 4u) + zlane;
#pragma unroll
        for (int i = 0; i < U; ++i) {
            R.z[i] = __builtin_amdgcn_raw_buffer_load_b128(zr, so + (unsigned)(RSTEP * i * NN * 4), 0, 0);
            if (MODE == 0) R.y[i] = __builtin_amdgcn_raw_buffer_load_b128(yr, so + (unsigned)(RSTEP * i * NN * 4), 0, 0);
        }
    };
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);            // bias partial: column sums of dY over this thread's pixels
    auto transform = [&](const Raw& R, const int buf, const float okf) {
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 vs = make_float4(1.f, 1.f, 1.f, 1.f), vm = zero4, vi = zero4, v0 = zero4, v1 = zero4;
        if (MODE == 0) {
            vs = *reinterpret_cast<const float4*>(vec + 4 * n4);
            vm = *reinterpret_cast<const float4*>(vec + NN + 4 * n4);
            vi = *reinterpret_cast<const float4*>(vec + 2 * NN + 4 * n4);
            v0 = *reinterpret_cast<const float4*>(vec + 3 * NN + 4 * n4);
            v1 = *reinterpret_cast<const float4*>(vec + 4 * NN + 4 * n4);
        }
        unsigned char* base = img + buf * IMG + prow * PITCH + 8 * n4;
#pragma unroll
        for (int i = 0; i < U; ++i) {
            float4 d = make_float4(__uint_as_float(R.z[i].x), __uint_as_float(R.z[i].y), __uint_as_float(R.z[i].z), __uint_as_float(R.z[i].w));
            if (MODE == 0) {
                const float4 y = make_float4(__uint_as_float(R.y[i].x), __uint_as_float(R.y[i].y), __uint_as_float(R.y[i].z), __uint_as_float(R.y[i].w));
                // the arithmetic of bn_act_bwd_apply (elementwise.hip), element for element
                d.x = vs.x * (d.x - v0.x - (y.x - vm.x) * vi.x * v1.x);
                d.y = vs.y * (d.y - v0.y - (y.y - vm.y) * vi.y * v1.y);
                d.z = vs.z * (d.z - v0.z - (y.z - vm.z) * vi.z * v1.z);
                d.w = vs.w * (d.w - v0.w - (y.w - vm.w) * vi.w * v1.w);
            }
            bs.x = fmaf(d.x, okf, bs.x); bs.y = fmaf(d.y, okf, bs.y); bs.z = fmaf(d.z, okf, bs.z); bs.w = fmaf(d.w, okf, bs.w);
            uint2 q1, q2;
            split4h(make_float4(d.x * sg, d.y * sg, d.z * sg, d.w * sg), q1, q2);
            *reinterpret_cast<uint2*>(base + RSTEP * i * PITCH) = q1;
            *reinterpret_cast<uint2*>(base + RSTEP * i * PITCH + PL) = q2;
        }
    };

    Raw R;
    issue(R, s0);

    // ---- matrix role: this wave's columns
    const int c0 = wave * 16 * CW;              // (within the chunk)
    // x in the C layout of a 16 x 16 tile: lane = column lc, register r = pixel row 4 lg + r
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.M * ld * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t or_ = __builtin_amdgcn_make_buffer_rsrc(p.dzx, 0, (int)((size_t)p.M * ld * 4u), 0x00020000);
    const unsigned xlane = (unsigned)((4 * lg * ld + col0 + c0 + lc) * 4);
    const unsigned stage_x = 32u * (unsigned)ld * 4u;
    float xv[2][CW][4];
    float pv[RAW ? 2 : 1][CW][4];               // RAW + accumulate: what the gradient tensor holds already
    auto issue_x = [&](const int rt, const int s) {
        const unsigned sbase = stage_off(s, stage_x) + (unsigned)(16 * rt * ld * 4) + xlane;
#pragma unroll
        for (int cw = 0; cw < CW; ++cw)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xv[rt][cw][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, sbase + (unsigned)(cw * 64), (unsigned)(r * ld * 4), 0));
                if (RAW) pv[rt][cw][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                    or_, p.accumulate ? sbase + (unsigned)(cw * 64) : OOB, (unsigned)(r * ld * 4), 0));
            }
    };
    issue_x(0, s0);
    issue_x(1, s0);

    // W: hi plane of this wave's columns in registers, lo plane of the chunk's columns in LDS
    f16x8 whi[KS][CW];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int cw = 0; cw < CW; ++cw)
            whi[ks][cw] = *reinterpret_cast<const f16x8*>(p.wq + (size_t)(col0 + c0 + 16 * cw + lc) * NN + 32 * ks + 8 * lg);
    {
        constexpr int UN = CC * NN / 8;             // 16-byte units of the lo plane
        static_assert(UN % NTHR == 0, "weight units per thread");
#pragma unroll
        for (int j = 0; j < UN / NTHR; ++j) {
            const int u = tid + NTHR * j;
            const int c = u / (NN / 8), k8 = u % (NN / 8);
            *reinterpret_cast<uint4*>(wlo + c * PITCH + 16 * k8) =
                *reinterpret_cast<const uint4*>(p.wq + (size_t)p.wq_stride + (size_t)(col0 + c) * NN + 8 * k8);
        }
    }
    if (MODE == 0) {
        for (int k = tid; k < NN; k += NTHR) {
            vec[k] = p.y_scale[k]; vec[NN + k] = p.y_mean[k]; vec[2 * NN + k] = p.y_invstd[k];
            vec[3 * NN + k] = p.y_coef[k]; vec[4 * NN + k] = p.y_coef[NN + k];
        }
    }
    // BatchNorm vectors of x for this lane's columns
    float xsc[CW], xsh[CW], xmu[CW], xis[CW];
#pragma unroll
    for (int cw = 0; cw < CW; ++cw) {
        const int c = col0 + c0 + 16 * cw + lc;
        xsc[cw] = RAW ? 1.f : p.in_scale[c]; xsh[cw] = RAW ? 0.f : p.in_shift[c];
        xmu[cw] = RAW ? 0.f : p.in_mean[c]; xis[cw] = RAW ? 0.f : p.in_invstd[c];
    }
    const float relu_lo = (!RAW && p.in_relu) ? 0.f : -__builtin_inff();
    __syncthreads();                                // vec is complete
    transform(R, 0, s0 < s1 ? 1.f : 0.f);
    issue(R, s0 + 1);
    __syncthreads();

    f32x4 dw[NT][CW];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int cw = 0; cw < CW; ++cw) dw[nt][cw] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float s1a[CW], s2a[CW];
#pragma unroll
    for (int cw = 0; cw < CW; ++cw) { s1a[cw] = 0.f; s2a[cw] = 0.f; }
    float am = 0.f;

    B1_LDS unsigned char* lds0 = (B1_LDS unsigned char*)b1_smem;
    const unsigned a_lane = (unsigned)(lc * PITCH + 16 * lg);                        // row read: pixel lc, k = 8 lg ..
    const unsigned t_lane = (unsigned)((4 * lg + ((lane >> 2) & 3)) * PITCH + 8 * (lane & 3));   // transposed read
    const unsigned w_lane = (unsigned)(2 * IMG + (c0 + lc) * PITCH + 16 * lg);

    // (a lambda, run once before the loop: with the first stage peeled, the loop header is entered from two edges with the
    // SAME order of outstanding memory operations, and hipcc's s_waitcnt vmcnt counts stay exact across the back edge —
    // entered straight from the prologue it waits for the x loads of a whole stage at the top of every iteration)
    auto stage = [&](const int s) {
        const int buf = (s - s0) & 1;
        // the next stage: registers -> the other buffer (nobody reads it before the barrier below); then its successor's loads
        transform(R, buf ^ 1, s + 1 < s1 ? 1.f : 0.f);
        issue(R, s + 2);
        __builtin_amdgcn_sched_barrier(0);

        const unsigned ib = (unsigned)(buf * IMG);
        f16x8 a1[CW], a2[CW];                       // act(x) planes of this stage: the B operand of dW (k = pixel)
        unsigned a1u[CW][4], a2u[CW][4];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            // ---- dX tile: 16 pixels x 16 CW columns
            f32x4 acc[CW];
#pragma unroll
            for (int cw = 0; cw < CW; ++cw) acc[cw] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const unsigned ao = ib + (unsigned)(rt * 16 * PITCH + 64 * ks) + a_lane;
                const f16x8 d1 = *reinterpret_cast<const f16x8*>(b1_smem + ao);
                const f16x8 d2 = *reinterpret_cast<const f16x8*>(b1_smem + ao + PL);
#pragma unroll
                for (int cw = 0; cw < CW; ++cw) {
                    const f16x8 b2 = *reinterpret_cast<const f16x8*>(b1_smem + w_lane + cw * 16 * PITCH + 64 * ks);
                    acc[cw] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d2, whi[ks][cw], acc[cw], 0, 0, 0);
                    acc[cw] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d1, b2, acc[cw], 0, 0, 0);
                    acc[cw] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d1, whi[ks][cw], acc[cw], 0, 0, 0);
                }
            }
            // ---- its epilogue: ReLU mask of bn(x), dz out, the BatchNorm-backward sums; act(x) planes for dW
            const unsigned obase = (unsigned)(s * 32 + 16 * rt) * (unsigned)ld * 4u + xlane;
#pragma unroll
            for (int cw = 0; cw < CW; ++cw) {
                float av[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xx = xv[rt][cw][r];
                    float v = acc[cw][r] * osc_x;
                    if (RAW) {
                        v += pv[rt][cw][r];               // (zeros unless accumulate)
                        av[r] = xx * sa;
                    } else {
                        const float z = fmaf(xx, xsc[cw], xsh[cw]);
                        if (p.in_relu && z <= 0.f) v = 0.f;
                        s1a[cw] += v;
                        s2a[cw] = fmaf(v, (xx - xmu[cw]) * xis[cw], s2a[cw]);
                        av[r] = fmaxf(z, relu_lo) * sa;
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), or_, obase + (unsigned)(cw * 64), (unsigned)(r * ld * 4), 0);
                    am = fmaxf(am, fabsf(v));
                }
                uint2 q1, q2;
                split4h(make_float4(av[0], av[1], av[2], av[3]), q1, q2);
                a1u[cw][2 * rt] = q1.x; a1u[cw][2 * rt + 1] = q1.y;
                a2u[cw][2 * rt] = q2.x; a2u[cw][2 * rt + 1] = q2.y;
            }
            issue_x(rt, s + 1);                     // the next stage's x of this row tile: in flight for a whole stage
        }
#pragma unroll
        for (int cw = 0; cw < CW; ++cw) {
            a1[cw] = __builtin_bit_cast(f16x8, (b1_u32x4){a1u[cw][0], a1u[cw][1], a1u[cw][2], a1u[cw][3]});
            a2[cw] = __builtin_bit_cast(f16x8, (b1_u32x4){a2u[cw][0], a2u[cw][1], a2u[cw][2], a2u[cw][3]});
        }
        // ---- dW += dY^T act(x) over the stage's 32 pixels (element j of a lane: pixel 4 lg + j, then 16 + 4 lg + j - 4)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            B1_LDS unsigned char* tb = lds0 + ib + t_lane + 32 * nt;
            const f16x8 t1 = b1_tr_frag(tb, tb + 16 * PITCH);
            const f16x8 t2 = b1_tr_frag(tb + PL, tb + PL + 16 * PITCH);
#pragma unroll
            for (int cw = 0; cw < CW; ++cw) {
                dw[nt][cw] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t2, a1[cw], dw[nt][cw], 0, 0, 0);
                dw[nt][cw] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t1, a2[cw], dw[nt][cw], 0, 0, 0);
                dw[nt][cw] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t1, a1[cw], dw[nt][cw], 0, 0, 0);
            }
        }
        __syncthreads();
    };
    stage(s0);
    for (int s = s0 + 1; s < s1; ++s) stage(s);

    // ---- slab of this workgroup's pixels: ws[wg][n][c] (D row = n: 4 lg + r of the tile, D column = c: the lane); the column
    // chunks of one pixel range fill one slab
    float* slab = p.ws + (size_t)wg * NN * ld;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int cw = 0; cw < CW; ++cw)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                slab[(size_t)(16 * nt + 4 * lg + r) * ld + col0 + c0 + 16 * cw + lc] = dw[nt][cw][r] * osc_w;
    // BatchNorm-backward sums of bn(x): one row per workgroup
    if (!RAW) {
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
    // bias partial: the RSTEP pixel-threads of a channel group add up through LDS in pixel order (the loop ended on a barrier)
    if (blockIdx.y == 0) {
        float4* red = reinterpret_cast<float4*>(b1_smem);
        red[prow * N4 + n4] = bs;
        __syncthreads();
        if (tid < N4) {
            float4 t = red[tid];
            for (int j = 1; j < RSTEP; ++j) {
                const float4 v = red[j * N4 + tid];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            *reinterpret_cast<float4*>(p.ws + (size_t)p.nwg * NN * ld + (size_t)wg * NN + 4 * tid) = t;
        }
    }
    if (p.dzx_amax) amax_commit(am, p.dzx_amax);
}

// ---------------------------------------------------------------- host side
static int b1_enabled = -1;

struct B1Cfg { int cout, cin, nn, cw, nwv, chunks; };
static const B1Cfg b1_cfgs[] = {
    {128, 256, 128, 2, 8, 1},       // conv1 of a Bottleneck
    {256, 128, 256, 1, 8, 1},       // conv3, the 128 -> 256 projection shortcut
    {128, 128, 128, 1, 8, 1},       // conv1 of a 128-wide Bottleneck
    {64, 64, 64, 1, 4, 1},          // the 128 x 128 level: conv1 ...
    {128, 64, 128, 1, 4, 1},        // ... conv3 and the projection shortcut
    {256, 256, 256, 1, 8, 2},       // the `fc` convolutions: two column chunks
};

Bwd1Plan dsnt_bwd1_plan(const dsnt_conv_geom* g, bool share) {
    Bwd1Plan pl;
    memset(&pl, 0, sizeof(pl));
    if (b1_enabled < 0) b1_enabled = dsnt_kernel_off("bwd1") ? 0 : 1;
    if (!b1_enabled || !g) return pl;
    if (!(g->R == 1 && g->S == 1 && g->stride == 1 && g->pad == 0 && g->Ho == g->H && g->Wo == g->W)) return pl;
    const long M = (long)g->N * g->H * g->W;
    static long min_rows = -1;
    if (min_rows < 0) { const char* e = getenv("DSNT_X_BWD1_MIN_ROWS"); min_rows = e ? atol(e) : 16384; }      // A/B only
    if (M % 32 != 0 || M < min_rows) return pl;
    if ((size_t)M * g->Cin * 4u >= (1ull << 31) || (size_t)M * g->Cout * 4u >= (1ull << 31)) return pl;
    int cfg = -1;
    for (int i = 0; i < (int)(sizeof(b1_cfgs) / sizeof(b1_cfgs[0])); ++i)
        if (g->Cout == b1_cfgs[i].cout && g->Cin == b1_cfgs[i].cin) cfg = i;
    if (cfg < 0) return pl;
    const B1Cfg& c = b1_cfgs[cfg];
    const int cus = dsnt_device_cus();
    const int nstages = (int)(M / 32);
    // workgroups that fit the chip at once: one eight-wave workgroup per CU, two four-wave ones; column chunks side by side.
    // share (DSNT_CONV_SHARE_CHIP): the launch runs on a lane beside the dependency chain; a workgroup holds up to 140 KB of a
    // CU's LDS for the whole launch, so it keeps to half of the CUs (as the streaming 1x1 kernel does: gemm1.hip)
    int nwg = cus * (c.nwv == 4 ? 2 : 1) / c.chunks;
    if (share) nwg = nwg / 2 > 0 ? nwg / 2 : 1;
    if (c.chunks > 1) nwg = nwg / 8 * 8 > 0 ? nwg / 8 * 8 : nwg;      // chunk partners on one XCD (speed only)
    if (nwg > nstages) nwg = nstages;
    const int spw = (nstages + nwg - 1) / nwg;
    nwg = (nstages + spw - 1) / spw;                        // no workgroup without a stage
    const int pitch = c.nn * 2 + 32;
    pl.ok = 1; pl.cfg = cfg; pl.nstages = nstages; pl.spw = spw; pl.nwg = nwg; pl.chunks = c.chunks;
    pl.lds = 2 * 2 * 32 * pitch + 16 * c.cw * c.nwv * pitch + 5 * c.nn * 4;
    if (pl.lds < 64 * c.nwv * 16) pl.lds = 64 * c.nwv * 16; // the bias reduction's scratch
    return pl;
}

template <int NN, int CW, int NWV, int MODE, bool RAW>
static void b1_launch_k(const Bwd1Plan& pl, const Bwd1P& p, hipStream_t st) {
    DSNT_SET_MAX_LDS((bwd1_kernel<NN, CW, NWV, MODE, RAW>), pl.lds);
    DSNT_LAUNCH((bwd1_kernel<NN, CW, NWV, MODE, RAW>), dim3(pl.nwg, pl.chunks), dim3(64 * NWV), pl.lds, st, p);
}

template <int NN, int CW, int NWV>
static void b1_launch_cfg(const Bwd1Plan& pl, const Bwd1P& p, bool apply, bool raw, hipStream_t st) {
    if (raw) b1_launch_k<NN, CW, NWV, 1, true>(pl, p, st);
    else if (apply) b1_launch_k<NN, CW, NWV, 0, false>(pl, p, st);
    else b1_launch_k<NN, CW, NWV, 1, false>(pl, p, st);
}

void dsnt_bwd1_launch(const Bwd1Plan& pl, const dsnt_bn_bwd_epilogue* xs, const float* dy, const dsnt_bn_bwd_apply* ap,
                      const void* wd_planes, int64_t plane_stride, const float* w_bound, const float* a_bound,
                      const float* g_bound, float* dz_out, float* stats, float* ws, float* dz_amax, int accumulate,
                      const dsnt_conv_geom* g, hipStream_t st) {
    Bwd1P p;
    memset(&p, 0, sizeof(p));
    p.dz = dy;
    if (ap) { p.yo = ap->y; p.y_scale = ap->scale; p.y_mean = ap->mean; p.y_invstd = ap->invstd; p.y_coef = ap->coef; }
    p.g_bound = g_bound;
    const bool raw = xs->scale == nullptr;
    p.x = xs->x; p.in_scale = xs->scale; p.in_shift = xs->shift; p.in_mean = xs->mean; p.in_invstd = xs->invstd;
    p.in_relu = raw ? 0 : xs->relu;
    p.a_bound = a_bound;
    p.wq = (const unsigned short*)wd_planes; p.wq_stride = plane_stride; p.w_bound = w_bound;
    p.dzx = dz_out; p.stats = stats; p.ws = ws; p.dzx_amax = (unsigned*)dz_amax;
    p.M = g->N * g->H * g->W; p.Cin = g->Cin; p.nstages = pl.nstages; p.spw = pl.spw; p.nwg = pl.nwg;
    p.accumulate = raw ? accumulate : 0;
    switch (pl.cfg) {
    case 0: b1_launch_cfg<128, 2, 8>(pl, p, ap != nullptr, raw, st); break;
    case 1: case 5: b1_launch_cfg<256, 1, 8>(pl, p, ap != nullptr, raw, st); break;
    case 2: b1_launch_cfg<128, 1, 8>(pl, p, ap != nullptr, raw, st); break;
    case 3: b1_launch_cfg<64, 1, 4>(pl, p, ap != nullptr, raw, st); break;
    default: b1_launch_cfg<128, 1, 4>(pl, p, ap != nullptr, raw, st); break;
    }
}

extern "C" int dsnt_conv1x1_bwd_ok(const dsnt_conv_geom* g) { return dsnt_bwd1_plan(g, false).ok; }
extern "C" int dsnt_conv1x1_bwd_splits(const dsnt_conv_geom* g, int flags) { return dsnt_bwd1_plan(g, (flags & DSNT_CONV_SHARE_CHIP) != 0).nwg; }
extern "C" int64_t dsnt_conv1x1_bwd_ws_floats(const dsnt_conv_geom* g, int flags) {
    const Bwd1Plan pl = dsnt_bwd1_plan(g, (flags & DSNT_CONV_SHARE_CHIP) != 0);
    return pl.ok ? (int64_t)pl.nwg * g->Cout * (g->Cin + 1) : 0;
}

extern "C" int dsnt_conv1x1_bwd_f16x3(const dsnt_bn_bwd_epilogue* xs, const float* dy, const dsnt_bn_bwd_apply* ap,
                                      const void* wd_planes, int64_t plane_stride, const float* w_bound,
                                      const float* a_bound, const float* g_bound, float* dz_out, float* stats_partial,
                                      float* ws, float* dz_amax, int flags, const dsnt_conv_geom* g, void* stream) {
    DSNT_REQUIRE(xs && xs->x && dy && wd_planes && w_bound && a_bound && g_bound && dz_out && ws && g, DSNT_ERR_ARG,
                 "dsnt_conv1x1_bwd_f16x3: bad argument");
    const bool raw = xs->scale == nullptr;
    DSNT_REQUIRE(raw ? (!xs->shift && !xs->mean && !xs->invstd && !ap)
                     : (xs->shift && xs->mean && xs->invstd && stats_partial != nullptr), DSNT_ERR_ARG,
                 "dsnt_conv1x1_bwd_f16x3: a BatchNorm in front of the convolution needs scale/shift/mean/invstd and stats_partial; "
                 "without one (scale == NULL) none of them, and no dsnt_bn_bwd_apply");
    DSNT_REQUIRE(!ap || (ap->y && ap->scale && ap->mean && ap->invstd && ap->coef), DSNT_ERR_ARG,
                 "dsnt_conv1x1_bwd_f16x3: incomplete dsnt_bn_bwd_apply");
    const Bwd1Plan pl = dsnt_bwd1_plan(g, (flags & DSNT_CONV_SHARE_CHIP) != 0);
    DSNT_REQUIRE(pl.ok, DSNT_ERR_SHAPE, "dsnt_conv1x1_bwd_f16x3: geometry not supported (dsnt_conv1x1_bwd_ok)");
    DSNT_REQUIRE(dsnt_aligned16(xs->x) && dsnt_aligned16(dy) && dsnt_aligned16(wd_planes) && dsnt_aligned16(dz_out) &&
                 dsnt_aligned16(ws) && (!ap || dsnt_aligned16(ap->y)) && plane_stride % 8 == 0, DSNT_ERR_ALIGN,
                 "dsnt_conv1x1_bwd_f16x3: 16-byte alignment required");
    dsnt_bwd1_launch(pl, xs, dy, ap, wd_planes, plane_stride, w_bound, a_bound, g_bound, dz_out, stats_partial, ws,
                     dz_amax, (flags & 1) != 0, g, (hipStream_t)stream);
    DSNT_CHECK_LAUNCH("dsnt_conv1x1_bwd_f16x3");
}
