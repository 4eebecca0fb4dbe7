// Calibration kernels (not on the product path): sustained fp32-MFMA rate of this device.
#include "common.h"
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Each wave runs `iters` x 16 back-to-back v_mfma_f32_32x32x2_f32.  DEP = 1: four independent
// accumulators used round-robin; DEP = 4: chains of 4 dependent MFMAs on one accumulator.
template <int DEP>
__global__ void mfma_peak_kernel(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = DEP == 1 ? i : r;
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
            }
        }
        a += 1e-6f;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int dsnt_debug_mfma_peak(float* out, int blocks, int threads, int iters, int dep, void* stream) {
    DSNT_REQUIRE(out && blocks > 0 && threads > 0 && iters > 0, DSNT_ERR_ARG, "dsnt_debug_mfma_peak: bad argument");
    if (dep == 4)
        hipLaunchKernelGGL(mfma_peak_kernel<4>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, iters, 0.5f, 0.25f);
    else
        hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, iters, 0.5f, 0.25f);
    DSNT_CHECK_LAUNCH("dsnt_debug_mfma_peak");
}

// MFMA/VALU co-execution probe: waves 0..3 run `mfma_iters` x 16 fp32 MFMAs, waves 4..7 run
// `valu_iters` x 64 dependent-free v_fma_f32.  Compare the time of (both) with (each alone).
__global__ __launch_bounds__(512) void coexec_kernel(float* out, int mfma_iters, int valu_iters, float a0) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float s = 0.f;
    if (wave < 4) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        float a = a0 + threadIdx.x * 1e-3f, b = a0 - threadIdx.x * 1e-3f;
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r & 3], 0, 0, 0);
            a += 1e-6f;
        }
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 16; ++e) s += acc[i][e];
    } else {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = a0 + i + threadIdx.x;
        const float m = 1.0001f, c = 1e-4f;
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], m, c);
        }
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int dsnt_debug_coexec(float* out, int blocks, int mfma_iters, int valu_iters, void* stream) {
    DSNT_REQUIRE(out && blocks > 0, DSNT_ERR_ARG, "dsnt_debug_coexec: bad argument");
    hipLaunchKernelGGL(coexec_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, out, mfma_iters, valu_iters, 0.5f);
    DSNT_CHECK_LAUNCH("dsnt_debug_coexec");
}

// bf16 MFMA calibration: waves 0..3 run `mfma_iters` x 16 v_mfma_f32_32x32x16_bf16; waves 4..7 (if the
// block has 512 threads) run `valu_iters` x 64 v_fma_f32 — do bf16 MFMA and VALU co-execute?
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void bf16_peak_kernel(float* out, int mfma_iters, int valu_iters, float a0) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float s = 0.f;
    if (wave < 4) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + threadIdx.x * 1e-3f + e); b[e] = (__bf16)(a0 - e * 0.1f); }
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[r & 3], 0, 0, 0);
            a[0] = (__bf16)((float)a[0] + 1.0f);
        }
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 16; ++e) s += acc[i][e];
    } else {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = a0 + i + threadIdx.x;
        const float m = 1.0001f, c = 1e-4f;
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], m, c);
        }
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int dsnt_debug_bf16_peak(float* out, int blocks, int threads, int mfma_iters, int valu_iters, void* stream) {
    DSNT_REQUIRE(out && blocks > 0 && (threads == 256 || threads == 512), DSNT_ERR_ARG, "dsnt_debug_bf16_peak: bad argument");
    hipLaunchKernelGGL(bf16_peak_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, mfma_iters, valu_iters, 0.5f);
    DSNT_CHECK_LAUNCH("dsnt_debug_bf16_peak");
}
