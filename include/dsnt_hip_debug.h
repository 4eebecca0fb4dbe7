/*
 * dsnt_hip_debug.h — calibration probes and kernel-timeline switches of libdsnt_hip.so.
 *
 * NOT part of the product ABI (include/dsnt_hip.h): nothing on the dsnt-pose2d hot path calls these; they exist
 * for tools/ (peak-rate calibration, issue-starvation probe, in-kernel s_memtime timelines) and, unlike the
 * product entry points, dsnt_debug_set_timeline / dsnt_debug_force_gemm6 flip process-wide switches.
 */
#ifndef DSNT_HIP_DEBUG_H
#define DSNT_HIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ calibration
 * sustained v_mfma_f32_32x32x2_f32 rate of this device
 * (blocks x threads, `iters` x 16 MFMAs per wave; dep = 1 independent / 4 dependent chains). */
int dsnt_debug_mfma_peak(float* out, int blocks, int threads, int iters, int dep, void* stream);
/* Debug timeline of the conv kernel: lane 0 of every wave of workgroup `block` stamps s_memtime
 * into buf[wave*128 + slot] (buf = 8*128 int64 on the device; NULL switches it off). */
int dsnt_debug_set_timeline(long long* buf, int block);
/* MFMA / VALU co-execution probe (512-thread blocks: 4 MFMA waves + 4 v_fma waves). */
int dsnt_debug_coexec(float* out, int blocks, int mfma_iters, int valu_iters, void* stream);
/* bf16 MFMA rate (v_mfma_f32_32x32x16_bf16) and its co-execution with VALU (threads 256 or 512). */
int dsnt_debug_bf16_peak(float* out, int blocks, int threads, int mfma_iters, int valu_iters, void* stream);
/* Debug/bench switch: route 3x3 convolutions of the bf16x6 path through the implicit-GEMM kernel instead of
 * the LDS halo-tile kernel (process-wide; not for production use). */
int dsnt_debug_force_gemm6(int on);
/* Issue-starvation probe: cycles a burst of valu_n x 16 v_fma_f32 takes on waves that share their SIMDs with
 * waves saturating the bf16 matrix pipe (out_cycles[blocks*4], s_memtime units); prio = s_setprio level. */
int dsnt_debug_starve(float* out, long long* out_cycles, int blocks, int mfma_iters, int valu_n, int prio,
                      void* stream);

/* Cost of a chip-wide grid barrier vs a kernel boundary (tools/grid_barrier.py): `blocks` (<= 256, co-resident) workgroups run
 * `iters` barriers on `counter` (one uint32, zero before the launch); dsnt_debug_empty is the dependent-launch yardstick. */
int dsnt_debug_grid_barrier(unsigned* counter, int blocks, int threads, int iters, float* out, void* stream);
/* two-level form: counter = 16 * 9 zeroed uint32 (top + one per XCD, each on its own cache line) */
int dsnt_debug_grid_barrier2(unsigned* counter, int blocks, int threads, int iters, float* out, void* stream);
int dsnt_debug_empty(int blocks, int threads, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
