// Skinny 1x1 convolutions on the vector ALU, in fp32, at the streaming rate of their wide side (skinny.h).
//
// K = 16 (`score_` forward 16 -> 256 with two residual inputs and the next BatchNorm's statistics; the data gradient of `score`,
// 16 -> 256 with the BatchNorm-backward epilogue): a 256-thread workgroup owns 128 rows; a thread owns FOUR output channels and
// every fourth row — its 4 x 16 weights stay in registers (rebuilt in fp32 from the split planes: h1 + h2 over the power-of-two
// scale, or b1 + b2 + b3), a row's 16 inputs are wave-uniform (scalar loads), a wave reads and writes whole 1 KB rows.  Sixty-four
// FMAs per 16 output bytes: the ALU is a third of the HBM time.  fp32 operands, fp32 products: at least as accurate as the
// split-precision kernels it replaces; the statistics rows have the 128-row format of those kernels (another summation order).
#include "skinny.h"
#include <stdlib.h>

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float half_to_f32(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// w[k] (k < 16) of output channel n from the planes
template <bool F16>
__device__ __forceinline__ void skinny_w16(const ConvP& p, const int n, const float inv_sw, float (&w)[16]) {
    constexpr int NPL = F16 ? 2 : 3;
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = 0.f;
#pragma unroll
    for (int pl = NPL - 1; pl >= 0; --pl) {                   // small planes first
        const unsigned short* row = p.wq + (size_t)pl * p.wq_stride + (size_t)n * 16;
        const u32x4s lo = *reinterpret_cast<const u32x4s*>(row), hi = *reinterpret_cast<const u32x4s*>(row + 8);
        const unsigned q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned short e0 = (unsigned short)(q[j] & 0xffffu), e1 = (unsigned short)(q[j] >> 16);
            w[2 * j] += F16 ? half_to_f32(e0) : bf16_to_f32(e0);
            w[2 * j + 1] += F16 ? half_to_f32(e1) : bf16_to_f32(e1);
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] *= inv_sw;
}

template <bool F16, bool BNB>
__global__ __launch_bounds__(256) void skinny_k16_kernel(ConvP p) {
    __shared__ __attribute__((aligned(16))) float red[4][64][8];
    const int tid = threadIdx.x, cg = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C4 = p.Cout >> 2;                                  // 64
    const float inv_sw = F16 ? 1.f / pow2_scale(bound64(p.w_bound)) : 1.f;
    float w[4][16];
#pragma unroll
    for (int c = 0; c < 4; ++c) skinny_w16<F16>(p, 4 * cg + c, inv_sw, w[c]);
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!BNB && p.bias) bias = reinterpret_cast<const float4*>(p.bias)[cg];
    float4 bsc = bias, bsh = bias, bmu = bias, bis = bias;       // (BNB: the BatchNorm-backward vectors of this thread's channels)
    if (BNB) {
        bsc = reinterpret_cast<const float4*>(p.bnb_scale)[cg]; bsh = reinterpret_cast<const float4*>(p.bnb_shift)[cg];
        bmu = reinterpret_cast<const float4*>(p.bnb_mean)[cg]; bis = reinterpret_cast<const float4*>(p.bnb_invstd)[cg];
    }
    float4 ts = make_float4(0.f, 0.f, 0.f, 0.f), th = ts;        // tail.amax_bn: the consumer's BatchNorm vectors
    if (p.tail.amax_bn) { ts = reinterpret_cast<const float4*>(p.tail.amax_scale)[cg]; th = reinterpret_cast<const float4*>(p.tail.amax_shift)[cg]; }
    const float am2lo = p.tail.amax_relu ? 0.f : -__builtin_inff();
    const long m0 = (long)blockIdx.x * 128;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    float am = 0.f, am2 = 0.f;
    constexpr int UB = 4;                                        // rows in flight per thread
    for (int j0 = 0; j0 < 32; j0 += UB) {
        float4 r1[UB], r2[UB];
        float a[UB][16];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const long m = m0 + wave + 4 * (j0 + u);             // wave-uniform
            const float4* ar = reinterpret_cast<const float4*>(p.x + m * 16);
            const float4 a0 = ar[0], a1 = ar[1], a2 = ar[2], a3 = ar[3];
            a[u][0] = a0.x; a[u][1] = a0.y; a[u][2] = a0.z; a[u][3] = a0.w; a[u][4] = a1.x; a[u][5] = a1.y; a[u][6] = a1.z; a[u][7] = a1.w;
            a[u][8] = a2.x; a[u][9] = a2.y; a[u][10] = a2.z; a[u][11] = a2.w; a[u][12] = a3.x; a[u][13] = a3.y; a[u][14] = a3.z; a[u][15] = a3.w;
            r1[u] = make_float4(0.f, 0.f, 0.f, 0.f); r2[u] = r1[u];
            if (p.res1) r1[u] = reinterpret_cast<const float4*>(p.res1 + m * p.Cout)[cg];
            if (!BNB && p.res2) r2[u] = reinterpret_cast<const float4*>(p.res2 + m * p.Cout)[cg];
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const long m = m0 + wave + 4 * (j0 + u);
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) acc = fmaf(a[u][k], w[c][k], acc);
                v[c] = acc;
            }
            float4 o;
            if (BNB) {
                // y = dz = acc * [bn(x) > 0]; statistics (sum dz, sum dz * xhat); x arrives through res1
                const float4 xv = r1[u];
                o = make_float4(v[0], v[1], v[2], v[3]);
                if (p.bnb_relu) {
                    if (fmaf(xv.x, bsc.x, bsh.x) <= 0.f) o.x = 0.f;
                    if (fmaf(xv.y, bsc.y, bsh.y) <= 0.f) o.y = 0.f;
                    if (fmaf(xv.z, bsc.z, bsh.z) <= 0.f) o.z = 0.f;
                    if (fmaf(xv.w, bsc.w, bsh.w) <= 0.f) o.w = 0.f;
                }
                s1.x += o.x; s1.y += o.y; s1.z += o.z; s1.w += o.w;
                s2.x = fmaf(o.x, (xv.x - bmu.x) * bis.x, s2.x); s2.y = fmaf(o.y, (xv.y - bmu.y) * bis.y, s2.y);
                s2.z = fmaf(o.z, (xv.z - bmu.z) * bis.z, s2.z); s2.w = fmaf(o.w, (xv.w - bmu.w) * bis.w, s2.w);
            } else {
                o = make_float4(v[0] + bias.x + r1[u].x + r2[u].x, v[1] + bias.y + r1[u].y + r2[u].y,
                                v[2] + bias.z + r1[u].z + r2[u].z, v[3] + bias.w + r1[u].w + r2[u].w);
                s1.x += o.x; s1.y += o.y; s1.z += o.z; s1.w += o.w;
                s2.x = fmaf(o.x, o.x, s2.x); s2.y = fmaf(o.y, o.y, s2.y); s2.z = fmaf(o.z, o.z, s2.z); s2.w = fmaf(o.w, o.w, s2.w);
            }
            reinterpret_cast<float4*>(p.y + m * p.Cout)[cg] = o;
            am = fmaxf(fmaxf(am, fabsf(o.x)), fmaxf(fabsf(o.y), fmaxf(fabsf(o.z), fabsf(o.w))));
            if (p.tail.amax_bn)
                am2 = fmaxf(fmaxf(am2, fabsf(fmaxf(fmaf(o.x, ts.x, th.x), am2lo))),
                            fmaxf(fabsf(fmaxf(fmaf(o.y, ts.y, th.y), am2lo)),
                                  fmaxf(fabsf(fmaxf(fmaf(o.z, ts.z, th.z), am2lo)), fabsf(fmaxf(fmaf(o.w, ts.w, th.w), am2lo)))));
        }
    }
    if (p.stats) {
        // the four waves' sums in wave order (fixed: bit-reproducible), one statistics row per 128-row tile
        float* mine = &red[wave][cg][0];
        mine[0] = s1.x; mine[1] = s1.y; mine[2] = s1.z; mine[3] = s1.w; mine[4] = s2.x; mine[5] = s2.y; mine[6] = s2.z; mine[7] = s2.w;
        __syncthreads();
        if (tid < 64) {
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = red[0][tid][e];
#pragma unroll
            for (int wv = 1; wv < 4; ++wv)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += red[wv][tid][e];
            float* q0 = p.stats + ((size_t)blockIdx.x * 2 + 0) * p.Cout + 4 * tid;
            float* q1 = p.stats + ((size_t)blockIdx.x * 2 + 1) * p.Cout + 4 * tid;
            *reinterpret_cast<float4*>(q0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4*>(q1) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
    }
    (void)C4;
    if (p.tail.amax) amax_commit(am, p.tail.amax);
    if (p.tail.amax_bn) amax_commit(am2, p.tail.amax_bn, 1);
}

static int skinny_on = -1;

bool dsnt_skinny_launch(const ConvP& p, bool f16, hipStream_t st) {
    if (skinny_on < 0) skinny_on = dsnt_kernel_off("skinny") ? 0 : 1;
    if (!skinny_on) return false;
    if (!(p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0 && p.Ho == p.H && p.Wo == p.W)) return false;
    if (p.M % 128 != 0 || p.M < 4096) return false;
    if (p.in_scale || p.ap_y) return false;
    const bool bnb = p.bnb_scale != nullptr;
    if (p.Cin == 16 && p.Cout == 256) {
        // K = 16: forward with bias / residuals / statistics, or the data gradient with the BatchNorm-backward epilogue
        if (bnb && !(p.res1 && p.stats && !p.bias && !p.res2)) return false;
        if (!dsnt_aligned16(p.x) || !dsnt_aligned16(p.y) || !dsnt_aligned16(p.wq) || (p.wq_stride % 8) != 0) return false;
        if ((p.res1 && !dsnt_aligned16(p.res1)) || (p.res2 && !dsnt_aligned16(p.res2)) || (p.stats && !dsnt_aligned16(p.stats)) ||
            (p.bias && !dsnt_aligned16(p.bias)))
            return false;
        if (bnb && !(dsnt_aligned16(p.bnb_scale) && dsnt_aligned16(p.bnb_shift) && dsnt_aligned16(p.bnb_mean) && dsnt_aligned16(p.bnb_invstd)))
            return false;
        const dim3 grid((unsigned)(p.M / 128)), block(256);
        if (f16) {
            if (bnb) DSNT_LAUNCH((skinny_k16_kernel<true, true>), grid, block, 0, st, p);
            else DSNT_LAUNCH((skinny_k16_kernel<true, false>), grid, block, 0, st, p);
        } else {
            if (bnb) DSNT_LAUNCH((skinny_k16_kernel<false, true>), grid, block, 0, st, p);
            else DSNT_LAUNCH((skinny_k16_kernel<false, false>), grid, block, 0, st, p);
        }
        return true;
    }
    return false;
}
