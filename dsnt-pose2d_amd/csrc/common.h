// Shared host/device helpers for libdsnt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dsnt_hip.h"
#include "../../include/dsnt_hip_debug.h"

#define DSNT_WAVE 64

int dsnt_set_error(int code, const char* fmt, ...);

#define DSNT_REQUIRE(cond, code, ...)                         \
    do {                                                      \
        if (!(cond)) return dsnt_set_error(code, __VA_ARGS__); \
    } while (0)

struct dsnt_list;
dsnt_list* dsnt_recording(void);
#define DSNT_CHECK_LAUNCH(name)                                                   \
    do {                                                                          \
        if (dsnt_recording()) return DSNT_OK;      /* recorded, nothing launched */ \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess)                                                     \
            return dsnt_set_error(DSNT_ERR_HIP, "%s: %s", name, hipGetErrorString(e_)); \
        return DSNT_OK;                                                           \
    } while (0)

static inline bool dsnt_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// Per-DEVICE host caches (one process may drive several devices; entry points may be called from several threads — autograd's
// backward thread).  dsnt_device_id(): the current device, clamped to the cache size.  dsnt_device_cus(): its CU count.
#define DSNT_MAX_DEVICES 16
int dsnt_device_id(void);
int dsnt_device_cus(void);
// One-time (per kernel instantiation and device) opt-in to `bytes` of dynamic LDS; a failure is left in dsnt_last_error() and the
// launch that follows reports it.  Not a stream operation: safe while a launch list is recording.
#ifdef __HIPCC__
#include <atomic>
#define DSNT_SET_MAX_LDS(kernel, bytes)                                                                              \
    do {                                                                                                             \
        static std::atomic<int> done_[DSNT_MAX_DEVICES];                                                             \
        const int d_ = dsnt_device_id();                                                                             \
        if (done_[d_].load(std::memory_order_acquire) < (int)(bytes)) {                                              \
            hipError_t e_ = hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            if (e_ != hipSuccess)                                                                                    \
                dsnt_set_error(DSNT_ERR_HIP, "hipFuncSetAttribute(%s, %d bytes of LDS): %s", #kernel, (int)(bytes),  \
                               hipGetErrorString(e_));                                                               \
            else                                                                                                     \
                done_[d_].store((int)(bytes), std::memory_order_release);                                            \
        }                                                                                                            \
    } while (0)
#endif

// ---- launch lists (include/dsnt_hip.h: dsnt_list_*).  While a list is recording on this thread, every kernel launch
// of the entry points is captured (arguments by value) instead of enqueued, and the entry point's `stream` argument is a
// LANE index; dsnt_list_replay then issues the captured launches from C with no per-launch host work in the caller's
// language (a train step is ~650 launches: 5.5 ms of Python + ctypes per step otherwise).
#ifdef __HIPCC__
#include <functional>
void dsnt_record_launch(dsnt_list* l, int lane, std::function<void(hipStream_t)>&& fn);
// (code, gx, gy, nt, params: what a launch that can join a persistent stage leaves besides the closure — stage.h; code 0: nothing)
void dsnt_record_launch_op(dsnt_list* l, int lane, std::function<void(hipStream_t)>&& fn, int code, int gx, int gy, int nt,
                           const void* params, size_t bytes);
// a kernel whose ONLY argument is the parameter struct `p` (<= DSNT_STAGE_PARAM_BYTES, trivially copyable), 1-D or 2-D grid
#define DSNT_LAUNCH_OP(code, kernel, grid, block, lds, stream, p)                                            \
    do {                                                                                                     \
        static_assert(sizeof(p) <= 496, "stage op parameter block");                                         \
        const dim3 g_ = (grid), b_ = (block);                                                                \
        if (dsnt_list* rec_ = dsnt_recording())                                                              \
            dsnt_record_launch_op(rec_, (int)(intptr_t)(stream), [=](hipStream_t s_) {                       \
                hipLaunchKernelGGL(kernel, g_, b_, lds, s_, p);                                              \
            }, (code), (int)g_.x, (int)g_.y, (int)b_.x, &(p), sizeof(p));                                    \
        else                                                                                                 \
            hipLaunchKernelGGL(kernel, g_, b_, lds, (hipStream_t)(stream), p);                               \
    } while (0)
#define DSNT_LAUNCH(kernel, grid, block, lds, stream, ...)                                                   \
    do {                                                                                                     \
        if (dsnt_list* rec_ = dsnt_recording())                                                              \
            dsnt_record_launch(rec_, (int)(intptr_t)(stream), [=](hipStream_t s_) {                          \
                hipLaunchKernelGGL(kernel, grid, block, lds, s_, __VA_ARGS__);                               \
            });                                                                                              \
        else                                                                                                 \
            hipLaunchKernelGGL(kernel, grid, block, lds, (hipStream_t)(stream), __VA_ARGS__);                \
    } while (0)
#endif

#ifdef __HIPCC__
// Sum across the 64 lanes of a wave; every lane gets the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide reductions for 256-thread blocks (4 waves).  `red` is >= 4*NV floats of LDS.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();  // protect `red` from a previous use
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += red[w * NV + i];
        v[i] = s;
    }
}
// fp16x3 operand bounds: raise the 64-slot bound `amax` (see bound64 in conv.hip) to this 256-thread workgroup's
// max |value|: wave shuffles, four LDS floats, ONE fire-and-forget integer atomic on the float bits per workgroup,
// slot = workgroup index mod 64 (order-independent: the result is bit-reproducible).
// (`which` = 0 / 1: two commits in a row use separate LDS floats, so the second needs no barrier against the first)
// (`wg`: the workgroup index that picks the slot — blockIdx.x unless a persistent stage passes the recorded launch's, stage.h)
__device__ __forceinline__ void amax_commit(float am, unsigned* amax, int which = 0, int wg = -1) {
    __shared__ float amax_w2[32];
    float* amax_w = amax_w2 + 16 * which;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    if ((threadIdx.x & 63) == 0) amax_w[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 1; w < nw; ++w) am = fmaxf(am, amax_w[w]);
        if (am > 0.f) atomicMax(amax + ((wg < 0 ? (int)blockIdx.x : wg) & 63), __float_as_uint(am));
    }
}
__device__ __forceinline__ float block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float m = red[0];
    for (int w = 1; w < nw; ++w) m = fmaxf(m, red[w]);
    return m;
}
#endif

// DSNT_OFF=<name>[,<name>...] (',' or '+' between names): kernels of round 3 to leave out (conv3s, gemm1, wgrad3, wgrad1, dgrad_up) — the launch then takes the
// round-2 kernel of the same contract.  Host side; read at the first launch that asks.
#include <stdlib.h>
#include <string.h>
static inline bool dsnt_kernel_off(const char* name) {
    const char* e = getenv("DSNT_OFF");
    if (!e) return false;
    const size_t n = strlen(name);
    for (const char* q = e; (q = strstr(q, name)) != nullptr; q += n)
        if ((q == e || q[-1] == ',' || q[-1] == '+') && (q[n] == 0 || q[n] == ',' || q[n] == '+')) return true;
    return false;
}
