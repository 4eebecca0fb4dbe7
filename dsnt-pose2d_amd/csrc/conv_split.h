// Helpers shared by the split-precision convolution kernels (conv.hip, wgrad3.hip): the XCD-aware block remap, the
// fp16x3 operand scale / 64-slot bounds and the exact two-plane fp16 split.  Device code only (gfx950).
#pragma once
#include "common.h"
#include "bn_pro.h"
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void xcd_remap(int bid, int nwg, int& out) {
    // Blocks are dealt round-robin over the 8 XCDs; give every XCD a contiguous run of
    // tiles so neighbouring tiles (shared halo rows, shared A rows across n-tiles) meet in
    // one L2.  Bijective for any nwg (cdna guide §5, "XCD swizzle must be bijective").
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
    out = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

// fp16x3 operand scale (see the fp16x3 notes further down):
// 2^k with bound * 2^k in [2^13, 2^14)  (bound = m 2^E, 1 <= m < 2  ->  k = 13 - E); zero / tiny bounds are clamped
__device__ __host__ __forceinline__ float pow2_scale(float bound) {
    unsigned bits;
    memcpy(&bits, &bound, 4);
    int E = (int)((bits & 0x7fffffffu) >> 23) - 127;
    E = E < -100 ? -100 : (E > 100 ? 100 : E);
    const unsigned sb = (unsigned)(127 + 13 - E) << 23;
    float sc;
    memcpy(&sc, &sb, 4);
    return sc;
}
// A bound lives in DSNT_BOUND_SLOTS floats; its value is their maximum.  Producers that find it with atomics (the
// BN-backward apply kernel: thousands of workgroups) spread them over the slots by workgroup index: one hot address
// serialised the read-modify-writes and cost the apply kernel 30 %.
#define DSNT_BOUND_SLOTS 64
__device__ __forceinline__ float bound64(const float* __restrict__ p) {
    float b = p[threadIdx.x & 63];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o, 64));
    return b;
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_f16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// 4 (already scaled) floats -> two planes of 4 fp16: 3 VALU instructions per element
__device__ __forceinline__ void split4h(const float4 v, uint2& p1, uint2& p2) {
    p1.x = pk_f16(v.x, v.y); p1.y = pk_f16(v.z, v.w);
    const f16x2v a0 = __builtin_bit_cast(f16x2v, p1.x), a1 = __builtin_bit_cast(f16x2v, p1.y);
    p2.x = pk_f16(v.x - (float)a0.x, v.y - (float)a0.y);
    p2.y = pk_f16(v.z - (float)a1.x, v.w - (float)a1.y);
}

// Kernel arguments of the forward / data-gradient convolution kernels (conv.hip, gemm1.hip)
struct ConvP {
    const float* x; const float* w; const float* bias; float* y;
    const float* in_scale; const float* in_shift;
    const float* res1; const float* res2; float* stats;
    const unsigned short* wq;      // bf16x6 path: plane 0 of the weights; planes are `wq_stride` elements apart
    long wq_stride;
    // optional batch-norm-backward epilogue (data-gradient launches): y = dz = acc * [bn(x) > 0],
    // stats = per-tile (sum dz, sum dz*xhat); the BN input x is passed through res1
    const float* bnb_scale; const float* bnb_shift; const float* bnb_mean; const float* bnb_invstd;
    int bnb_relu;
    int in_relu;
    // fp16x3 path: device scalars >= max|A operand| and max|weights| (null on the other paths)
    const float* a_bound; const float* w_bound;
    int N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil;
    int M, K, mtiles, ntiles;
    // optional (conv3s.hip MODE 4, data-gradient launches with the BatchNorm-backward epilogue): the A operand is formed while it is
    // staged as  scale (x - c0 - (ap_y - mean) invstd c1)  — x = dL/dz of the BatchNorm behind this convolution, ap_y its input,
    // ap_coef = [c0 | c1] — and the patch's own pixels of it are written to ap_out (the materialised dL/dy)
    const float* ap_y; const float* ap_scale; const float* ap_mean; const float* ap_invstd; const float* ap_coef; float* ap_out;
    // optional: the operand bounds this launch leaves for the consumers of its output (bn_pro.h)
    OutBoundsP tail;
    // optional (pro.partial != null): the BatchNorm of the A operand is finalised in this launch's prologue — every
    // workgroup writes in_scale / in_shift (= pro.scale / pro.shift) itself before it reads them (bn_pro.h)
    BnProP pro;
};
