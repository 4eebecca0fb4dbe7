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

// Issue-starvation probe: waves 0..3 keep the matrix pipe saturated with bf16 MFMAs for `mfma_iters` x 16
// instructions while waves 4..7 time (s_memtime) a burst of `valu_n` x 16 v_fma_f32 (16 independent chains)
// issued in the middle of it.  out_cycles[block*4 + w] = cycles of the burst.  prio: s_setprio of the VALU waves.
template <int PAD>     // PAD: s_nop cycles (x16) after each MFMA: 0 = back-to-back
__global__ __launch_bounds__(512) void starve_kernel(float* out, long long* out_cycles, int mfma_iters, int valu_n,
                                                     int prio, float a0) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float s = 0.f;
    if (wave < 4) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + threadIdx.x * 1e-3f + e); b[e] = (__bf16)(a0 - e * 0.1f); }
        const long long m0 = __builtin_amdgcn_s_memtime();
        constexpr int pad = PAD;            // 0: back-to-back; 1..6: s_nop 1/3/5/6/7/9 after each MFMA; 7: s_sleep 1
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[r & 3], 0, 0, 0);
                if (pad >= 1) __builtin_amdgcn_sched_barrier(0);
                if (pad == 1) asm volatile("s_nop 1");
                if (pad == 2) asm volatile("s_nop 3");
                if (pad == 3) asm volatile("s_nop 5");
                if (pad == 4) asm volatile("s_nop 6");
                if (pad == 5) asm volatile("s_nop 7");
                if (pad == 6) asm volatile("s_nop 9");
                if (pad == 7) asm volatile("s_sleep 1");
                if (pad >= 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        const long long m1 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0 && mfma_iters > 0) out_cycles[gridDim.x * 4 + blockIdx.x * 4 + wave] = m1 - m0;
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 16; ++e) s += acc[i][e];
    } else {
        prio &= 15;
        if (prio == 1) __builtin_amdgcn_s_setprio(1);
        if (prio == 3) __builtin_amdgcn_s_setprio(3);
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = a0 + i + threadIdx.x;
        const float m = 1.0001f, c = 1e-4f;
        __builtin_amdgcn_s_sleep(20);           // let the MFMA waves get going
        const long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < valu_n; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], m, c);
        }
        for (int i = 0; i < 16; ++i) s += v[i];
        asm volatile("" :: "v"(s));
        const long long t1 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) out_cycles[blockIdx.x * 4 + wave - 4] = t1 - t0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int dsnt_debug_starve(float* out, long long* out_cycles, int blocks, int mfma_iters, int valu_n, int prio,
                                 void* stream) {
    DSNT_REQUIRE(out && out_cycles && blocks > 0, DSNT_ERR_ARG, "dsnt_debug_starve: bad argument");
    const int pad = prio >> 4;
#define LAUNCH_STARVE(P) hipLaunchKernelGGL(starve_kernel<P>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, out, \
                                            out_cycles, mfma_iters, valu_n, prio, 0.5f)
    switch (pad) {
        case 0: LAUNCH_STARVE(0); break; case 1: LAUNCH_STARVE(1); break; case 2: LAUNCH_STARVE(2); break;
        case 3: LAUNCH_STARVE(3); break; case 4: LAUNCH_STARVE(4); break; case 5: LAUNCH_STARVE(5); break;
        case 6: LAUNCH_STARVE(6); break; default: LAUNCH_STARVE(7); break;
    }
#undef LAUNCH_STARVE
    DSNT_CHECK_LAUNCH("dsnt_debug_starve");
}

// How much does a chip-wide grid barrier cost?  (fusing the sub-32x32 hourglass levels into persistent
// launches replaces ~5 us kernel boundaries by grid barriers — worth it only if a barrier is much cheaper.)
// `blocks` co-resident workgroups (the caller keeps blocks <= CUs) run `iters` barriers: one lane per workgroup adds to a
// counter with an agent-scope release, then polls it with agent-scope acquire loads until every workgroup of the round has
// arrived; a __syncthreads on both sides makes it a workgroup-wide barrier.  A bounded spin (2^16 polls) turns a lost
// workgroup into a wrong result instead of a hang.  out[0] = polls of workgroup 0 (diagnostic).
__global__ void grid_barrier_kernel(unsigned* counter, int iters, float* out) {
    unsigned polls = 0;
    for (int it = 1; it <= iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)it * gridDim.x;
            int guard = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want && ++guard < (1 << 16)) {
                ++polls;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && out) out[0] = (float)polls;
}

// Two-level form: one counter per XCD (workgroups are dealt round-robin to the 8 XCDs, so blockIdx.x & 7 is the XCD of a
// launch that starts on an idle chip — a grouping key, not a placement guarantee), the last arriver of a group adds to the top
// counter, everybody polls the top counter.  counter[0] = top, counter[16 * (1 + g)] = group g (own cache lines).
__global__ void grid_barrier2_kernel(unsigned* counter, int iters, float* out) {
    unsigned polls = 0;
    const int g = blockIdx.x & 7;
    const unsigned gsize = (gridDim.x >> 3) + ((unsigned)g < (gridDim.x & 7u) ? 1u : 0u);
    const unsigned ngroups = gridDim.x < 8 ? gridDim.x : 8;
    for (int it = 1; it <= iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(counter + 16 * (1 + g), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1 == (unsigned)it * gsize)
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)it * ngroups;
            int guard = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want && ++guard < (1 << 16)) {
                ++polls;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && out) out[0] = (float)polls;
}

extern "C" int dsnt_debug_grid_barrier2(unsigned* counter, int blocks, int threads, int iters, float* out, void* stream) {
    DSNT_REQUIRE(counter && blocks > 0 && blocks <= 256 && threads > 0 && threads <= 1024 && iters > 0, DSNT_ERR_ARG,
                 "dsnt_debug_grid_barrier2: bad argument (counter: 16 * 9 zeroed uint32; at most 256 workgroups)");
    DSNT_LAUNCH(grid_barrier2_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, counter, iters, out);
    DSNT_CHECK_LAUNCH("dsnt_debug_grid_barrier2");
}

extern "C" int dsnt_debug_grid_barrier(unsigned* counter, int blocks, int threads, int iters, float* out, void* stream) {
    DSNT_REQUIRE(counter && blocks > 0 && blocks <= 256 && threads > 0 && threads <= 1024 && iters > 0, DSNT_ERR_ARG,
                 "dsnt_debug_grid_barrier: bad argument (at most 256 workgroups: they must be co-resident)");
    DSNT_LAUNCH(grid_barrier_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, counter, iters, out);
    DSNT_CHECK_LAUNCH("dsnt_debug_grid_barrier");
}

__global__ void empty_kernel(float* out) { if (out && threadIdx.x == 0 && blockIdx.x == 0) out[1] = 1.f; }
extern "C" int dsnt_debug_empty(int blocks, int threads, float* out, void* stream) {
    DSNT_REQUIRE(blocks > 0 && threads > 0 && threads <= 1024, DSNT_ERR_ARG, "dsnt_debug_empty: bad argument");
    DSNT_LAUNCH(empty_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out);
    DSNT_CHECK_LAUNCH("dsnt_debug_empty");
}
