// Two things a statistics-producing launch shares with its neighbours (device side):
//   * OutBoundsP: the fp16x3 operand bounds a launch leaves behind for the consumers of its output (dsnt_out_bounds);
//   * the BatchNorm finalisation a CONSUMER does in its own prologue (dsnt_bn_prologue; round 3).
// (Rounds 2 and 3 also carried a finalisation in the PRODUCER's last workgroup — agent-scope tickets, write-through partial
// sums, two levels — in this place; measured slower than the separate launches on every workload in three rounds, last at
// hg2 batch 32: 12.49 ms without, 12.53 / 12.64 ms for launches of <= 2048 / <= 8192 rows, hg8 batch 16 +-0.  Deleted in round 4.)
#pragma once
#include "common.h"

// device-side image of dsnt_out_bounds (include/dsnt_hip.h) — same layout
struct OutBoundsP {
    unsigned* amax;                    // raise this 64-slot bound to max|output| (fp16x3 operand bounds)
    unsigned* amax_bn;                 // ... and this one to max|relu?(output * amax_scale[c] + amax_shift[c])|: the operand a
    const float* amax_scale;           // consumer with an eval-mode BatchNorm prologue will form (vectors known before
    const float* amax_shift;           // the producer runs)
    int amax_relu, pad_;
};

// a per-tile partial sum of a BatchNorm statistic (one writer per element; read by a later launch)
__device__ __forceinline__ void tail_store(float* p, float v) { *p = v; }

// ---------------------------------------------------------------------------------------------------------------------
// BatchNorm finalisation in the CONSUMER's prologue (round 3).  On the 8x8 / 4x4 hourglass levels a finalise launch is
// 16 workgroups between two convolutions that take 10 us themselves: measured, the ~120 such launches of an hg2 step
// cost it 1.0 ms (the step with them simply left out: 14.17 -> 13.13 ms).  A statistics tensor of <= 64 tiles is 64 KB
// at most: every workgroup of the launch that CONSUMES the BatchNorm sums the tiles itself (fp64, fixed order: all
// workgroups get the same bits), writes the vectors to the arrays the rest of the step reads them from (all workgroups
// the same values; one designated workgroup also moves the running statistics / adds into dgamma, dbeta) and goes on
// after ONE workgroup barrier — no ticket, no store drain, no extra launch.
struct BnProP {                          // device image of dsnt_bn_prologue (forward)
    const float* partial; int tiles, C;
    double invM, unbias;
    const float* gamma; const float* beta; float* rmean; float* rvar;
    float momentum, eps;
    float* mean; float* invstd; float* scale; float* shift;
};
struct BnBwdProP {                       // backward: (sum dz, sum dz * xhat) -> coef[2][C], dgamma, dbeta
    const float* partial; int tiles, C;
    double invM;
    float* dgamma; float* dbeta; int accumulate;
    float* coef;
};

// Both sums of every channel from partial[tiles][2][C] in ONE pass: thread = (tile lane, column of the 2C-wide row), 16
// independent loads in flight per thread (the pass is pure L2 latency: two round trips for 64 tiles), lanes added in lane
// order through LDS (fixed order: every workgroup gets the same bits).  Returns (sum, sum of squares / cross term) of
// channel threadIdx.x (< C) — C <= 256, NT in {256, 512}; sh: NT doubles.  Every thread of the workgroup calls.
template <int NT>
__device__ __forceinline__ void bn_pro_sums(const float* __restrict__ partial, int tiles, int C, double* sh, double& a0, double& a1) {
    const int tid = threadIdx.x;
    const int ncol = 2 * C;
    a0 = 0.0; a1 = 0.0;
    for (int cb = 0; cb < ncol; cb += NT) {                  // one chunk unless 2C > NT (C = 256 in a 256-thread workgroup)
        const int nc = ncol - cb < NT ? ncol - cb : NT;
        const int L = NT / nc;                               // tile lanes
        const int col = tid % nc, lane = tid / nc;
        double a = 0.0;
        if (lane < L) {
            const float* q = partial + cb + col;
            for (int r0 = lane; r0 < tiles; r0 += 16 * L) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int r = r0 + u * L;
                    v[u] = r < tiles ? q[(size_t)r * ncol] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) a += (double)v[u];
            }
        }
        __syncthreads();                                     // sh may still be read from the previous chunk / use
        sh[tid] = a;
        __syncthreads();
        // column c of this chunk = channel (cb + c) % C of statistic (cb + c) / C; channel `tid` picks up its two sums
        for (int st = 0; st < 2; ++st) {
            const int c = st * C + tid - cb;                 // position of (st, channel tid) inside this chunk
            if (tid < C && c >= 0 && c < nc) {
                double t = 0.0;
                for (int l = 0; l < L; ++l) t += sh[l * nc + c];
                if (st == 0) a0 = t; else a1 = t;
            }
        }
    }
}

// sh: NT doubles of LDS nobody else is using; ends WITHOUT a barrier — the caller puts one before the first read of
// scale / shift (its own workgroup's stores: waited for at the end of this function, visible after the caller's __syncthreads())
template <int NT>
__device__ __forceinline__ void bn_pro_forward(const BnProP& q, double* sh, bool writer) {
    // (the affine parameters and the running statistics are fetched BEFORE the sums: behind them a load is a microsecond of every
    // workgroup's prologue, i.e. of the dependency chain — elementwise.hip bn_finalize_kernel, round 5)
    const int c = threadIdx.x;
    float pg = 1.f, pb = 0.f, prm = 0.f, prv = 0.f;
    if (c < q.C) {
        if (q.gamma) pg = q.gamma[c];
        if (q.beta) pb = q.beta[c];
        if (writer && q.rmean) { prm = q.rmean[c]; prv = q.rvar[c]; }
    }
    double a0, a1;
    bn_pro_sums<NT>(q.partial, q.tiles, q.C, sh, a0, a1);
    if (c < q.C) {
        const double mean = a0 * q.invM;
        double var = a1 * q.invM - mean * mean;
        if (var < 0.0) var = 0.0;
        if (writer && q.rmean) {
            q.rmean[c] = (float)((1.0 - q.momentum) * prm + q.momentum * mean);
            q.rvar[c] = (float)((1.0 - q.momentum) * prv + q.momentum * var * q.unbias);
        }
        const float is = (float)(1.0 / sqrt(var + (double)q.eps));
        const float mu = (float)mean;
        const float sc = q.gamma ? pg * is : is;
        q.mean[c] = mu; q.invstd[c] = is; q.scale[c] = sc;
        q.shift[c] = (q.beta ? pb : 0.f) - mu * sc;
    }
    // the caller's __syncthreads() fences LDS only: this thread's global stores are waited for HERE, so that the other waves of
    // the workgroup read the new vectors behind the barrier by construction, not by the in-order habit of the L1 path (round 6)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}
template <int NT>
__device__ __forceinline__ void bn_pro_backward(const BnBwdProP& q, double* sh, bool writer) {
    const int c = threadIdx.x;
    float pdg = 0.f, pdb = 0.f;
    if (c < q.C && writer && q.accumulate) { if (q.dgamma) pdg = q.dgamma[c]; if (q.dbeta) pdb = q.dbeta[c]; }
    double a0, a1;
    bn_pro_sums<NT>(q.partial, q.tiles, q.C, sh, a0, a1);
    if (c < q.C) {
        if (writer) {
            const float sdz = (float)a0, sdzx = (float)a1;
            if (q.dgamma) q.dgamma[c] = q.accumulate ? pdg + sdzx : sdzx;
            if (q.dbeta) q.dbeta[c] = q.accumulate ? pdb + sdz : sdz;
        }
        q.coef[c] = (float)(a0 * q.invM);
        q.coef[q.C + c] = (float)(a1 * q.invM);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (as in bn_pro_forward)
}

// host side: validate and copy a dsnt_out_bounds into its device-side image (null = nothing asked for)
static inline int out_bounds_fill(OutBoundsP& out, const dsnt_out_bounds* in, const char* who) {
    static_assert(sizeof(OutBoundsP) == sizeof(dsnt_out_bounds), "dsnt_out_bounds layout");
    out.amax = in ? reinterpret_cast<unsigned*>(in->amax) : nullptr;
    out.amax_bn = in ? reinterpret_cast<unsigned*>(in->amax_bn) : nullptr;
    out.amax_scale = in ? in->amax_scale : nullptr; out.amax_shift = in ? in->amax_shift : nullptr;
    out.amax_relu = in ? in->amax_relu : 0; out.pad_ = 0;
    DSNT_REQUIRE(!out.amax_bn || (out.amax_scale && out.amax_shift && dsnt_aligned16(out.amax_scale) && dsnt_aligned16(out.amax_shift)),
                 DSNT_ERR_ARG, "%s: dsnt_out_bounds.amax_bn needs 16-byte aligned amax_scale / amax_shift", who);
    return DSNT_OK;
}
