// BatchNorm bookkeeping folded into the tail of the kernel that produced the per-tile partial sums.
//
// Every kernel that leaves per-tile column sums (conv epilogues, the K-split kernel, max-pool / upsample+add with
// statistics, and — in backward — the data-gradient epilogue with its (sum dz, sum dz*xhat) pairs) used to be followed
// by a bn_finalize launch of 8..16 workgroups: 193 launches of ~8 us per hg2 step, all on the dependency chain.  Here
// the LAST workgroup to arrive does that work before the kernel ends:
//
//   * two levels, so the serial tail stays short at any size: the last arriver of each group of BN_TAIL_GROUP m-tiles
//     adds that group's partial rows (fp64) into level2[group]; the last group to finish adds the groups and writes
//     the BatchNorm vectors.  Both sums run in tile / group index order whatever the arrival order: results are
//     bit-reproducible, no float atomics.
//   * hand-off = the ticket form of the CDNA guide's publish / consume recipe (§6 guideline 16, R1): partial sums
//     are stored write-through (sc1), every storing wave drains (s_waitcnt vmcnt(0)), barrier, one lane: relaxed
//     agent-scope fetch_add on the group's counter; the workgroup that draws the last ticket: ONE agent-scope acquire,
//     wait, barrier, then plain vector loads.  Correct for any placement of workgroups on XCDs / CUs.
//   * counters are zero before the first launch (the caller allocates them zeroed and clears them once per step) and
//     every last arriver puts its counter back to zero.
#pragma once
#include "common.h"

#define BN_TAIL_GROUP 32

// device-side image of dsnt_bn_tail (include/dsnt_hip.h) — same layout
struct BnTailP {
    int mode, accumulate;              // 0: forward statistics -> mean / invstd / scale / shift (+ running statistics)
                                       // 1: backward sums -> dgamma / dbeta (+= if accumulate) and coef[2][C]
    unsigned* counters;                // [1 + groups]
    double* level2;                    // [groups][2][C]
    const float* gamma; const float* beta; float* rmean; float* rvar;
    float momentum, eps;
    float* o0; float* o1; float* o2; float* o3;
    unsigned* amax;                    // independent of the tail: raise this 64-slot bound to max|output| (fp16x3 operand bounds)
    unsigned* amax_bn;                 // ... and this one to max|relu?(output * amax_scale[c] + amax_shift[c])|: the operand a
    const float* amax_scale;           // consumer with an eval-mode BatchNorm prologue will form (vectors known before
    const float* amax_shift;           // the producer runs)
    int amax_relu, pad_;
};

// Write-through (sc1) store of a partial sum: visible to every XCD once the storing wave has drained vmcnt, so the
// producers need NO release fence (an agent-scope release writes back the XCD's whole dirty L2 — with a freshly
// written activation tensor in it that cost 4 ms per hg2 step when every workgroup did it).
__device__ __forceinline__ void tail_store(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tail_store4(float* p, float4 v) {       // p 16-byte aligned
    unsigned long long lo = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
    unsigned long long hi = ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sum_{r = lane, lane + L, ... < n} src[r * stride] for column `col` of a chunk of `ncol` columns, all NT threads at
// work (thread = lane * ncol + col, L = NT / ncol lanes; every load independent: one memory latency, not n), then the
// lanes are added in lane order through LDS: a fixed summation order.  Result in sh[col] for col < ncol after the
// trailing barrier.  T = float (tile partials) or double (level-2 sums).
template <int NT, typename T>
__device__ __forceinline__ void tail_colsum(const T* src, size_t stride, int n, int ncol, double* sh) {
    const int tid = threadIdx.x;
    const int L = NT / ncol;
    const int col = tid % ncol, lane = tid / ncol;
    double a = 0.0;
    if (lane < L) {
        const T* q = src + (size_t)lane * stride + col;
#pragma unroll 4
        for (int r = lane; r < n; r += L, q += (size_t)L * stride) a += (double)*q;
    }
    __syncthreads();                       // sh may still be read from the previous chunk
    sh[tid] = a;
    __syncthreads();
    if (tid < ncol) {
        double s = sh[tid];
        for (int l = 1; l < L; ++l) s += sh[l * ncol + tid];
        sh[tid] = s;
    }
    __syncthreads();
}

// The BatchNorm vectors from n rows of per-channel sums src[n][2][C] (T = float: tile partials, double: level-2 sums),
// the rows added in index order in fp64 (spread over thread lanes, lanes added in lane order: a fixed order).
// Same arithmetic as bn_finalize_kernel in elementwise.hip.
template <int NT, typename T>
__device__ __forceinline__ void tail_finish(const BnTailP& t, const T* src, int n, int C, long M, double* sh) {
    const int tid = threadIdx.x;
    const double invM = 1.0 / (double)M;
    const double unbias = M > 1 ? (double)M / (double)(M - 1) : 1.0;
    constexpr int half = NT / 2;                             // NT / 2 threads per statistic
    const int CH = C < half ? C : half;                      // channels per chunk
    const size_t cols = 2 * (size_t)C;
    for (int c0 = 0; c0 < C; c0 += CH) {
        const int nch = C - c0 < CH ? C - c0 : CH;
        const int st = tid / half, r = tid - st * half;
        const int L = half / nch, cl = r % nch, lane = r / nch;
        double a = 0.0;
        if (lane < L) {
            const T* q = src + ((size_t)lane * 2 + st) * C + c0 + cl;
#pragma unroll 4
            for (int i = lane; i < n; i += L, q += (size_t)L * cols) a += (double)*q;
        }
        __syncthreads();                   // sh: earlier reads are done
        sh[tid] = a;
        __syncthreads();
        if (tid < nch) {
            double a0 = 0.0, a1 = 0.0;
            for (int l = 0; l < L; ++l) { a0 += sh[l * nch + tid]; a1 += sh[half + l * nch + tid]; }
            const int c = c0 + tid;
            if (t.mode == 0) {
                const double mean = a0 * invM;
                double var = a1 * invM - mean * mean;
                if (var < 0.0) var = 0.0;
                if (t.rmean) {
                    t.rmean[c] = (float)((1.0 - t.momentum) * t.rmean[c] + t.momentum * mean);
                    t.rvar[c] = (float)((1.0 - t.momentum) * t.rvar[c] + t.momentum * var * unbias);
                }
                const float is = (float)(1.0 / sqrt(var + (double)t.eps));
                const float mu = (float)mean;
                const float sc = t.gamma ? t.gamma[c] * is : is;
                t.o0[c] = mu; t.o1[c] = is; t.o2[c] = sc;
                t.o3[c] = (t.beta ? t.beta[c] : 0.f) - mu * sc;
            } else {
                const float sdz = (float)a0, sdzx = (float)a1;
                if (t.o0) t.o0[c] = t.accumulate ? t.o0[c] + sdzx : sdzx;      // dgamma
                if (t.o1) t.o1[c] = t.accumulate ? t.o1[c] + sdz : sdz;        // dbeta
                t.o2[c] = (float)(a0 * invM);
                t.o2[C + c] = (float)(a1 * invM);
            }
        }
    }
}

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
// scale / shift (its own workgroup's stores: visible after __syncthreads())
template <int NT>
__device__ __forceinline__ void bn_pro_forward(const BnProP& q, double* sh, bool writer) {
    double a0, a1;
    bn_pro_sums<NT>(q.partial, q.tiles, q.C, sh, a0, a1);
    const int c = threadIdx.x;
    if (c < q.C) {
        const double mean = a0 * q.invM;
        double var = a1 * q.invM - mean * mean;
        if (var < 0.0) var = 0.0;
        if (writer && q.rmean) {
            q.rmean[c] = (float)((1.0 - q.momentum) * q.rmean[c] + q.momentum * mean);
            q.rvar[c] = (float)((1.0 - q.momentum) * q.rvar[c] + q.momentum * var * q.unbias);
        }
        const float is = (float)(1.0 / sqrt(var + (double)q.eps));
        const float mu = (float)mean;
        const float sc = q.gamma ? q.gamma[c] * is : is;
        q.mean[c] = mu; q.invstd[c] = is; q.scale[c] = sc;
        q.shift[c] = (q.beta ? q.beta[c] : 0.f) - mu * sc;
    }
}
template <int NT>
__device__ __forceinline__ void bn_pro_backward(const BnBwdProP& q, double* sh, bool writer) {
    double a0, a1;
    bn_pro_sums<NT>(q.partial, q.tiles, q.C, sh, a0, a1);
    const int c = threadIdx.x;
    if (c < q.C) {
        if (writer) {
            const float sdz = (float)a0, sdzx = (float)a1;
            if (q.dgamma) q.dgamma[c] = q.accumulate ? q.dgamma[c] + sdzx : sdzx;
            if (q.dbeta) q.dbeta[c] = q.accumulate ? q.dbeta[c] + sdz : sdz;
        }
        q.coef[c] = (float)(a0 * q.invM);
        q.coef[q.C + c] = (float)(a1 * q.invM);
    }
}

// partial: [mtiles][2][C] written by this launch with tail_store (this workgroup's rows among them); M = rows the
// sums run over; arrivals = workgroups contributing to one m-tile's partial row (the launch's n-tiles); lds = at
// least NT * 8 + 16 bytes of LDS, 8-byte aligned, that no thread still reads for anything else (the caller put a
// barrier behind its last use).  Call as the LAST statement of the kernel.
template <int NT>
__device__ __forceinline__ void bn_tail_run(const BnTailP& t, float* partial, int mtiles, int C, long M, int mtile,
                                            int arrivals, void* lds) {
    double* sh = reinterpret_cast<double*>(lds);
    int* flag = reinterpret_cast<int*>(sh + NT);
    const int tid = threadIdx.x;
    const int g = mtile / BN_TAIL_GROUP;
    const int ngroups = (mtiles + BN_TAIL_GROUP - 1) / BN_TAIL_GROUP;
    const int t0 = g * BN_TAIL_GROUP;
    const int t1 = t0 + BN_TAIL_GROUP < mtiles ? t0 + BN_TAIL_GROUP : mtiles;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its sc1 stores ...
    __syncthreads();
    if (tid == 0) {                                           // ... then ONE lane draws the ticket
        const unsigned ticket = __hip_atomic_fetch_add(t.counters + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = ticket + 1u == (unsigned)((t1 - t0) * arrivals);
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(t.counters + 1 + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
        }
        *flag = last ? 1 : 0;
    }
    __syncthreads();
    if (*flag == 0) return;
    if (ngroups == 1) {                    // one group (<= 32 m-tiles, the latency-bound launches): straight to the vectors
        tail_finish<NT, float>(t, partial, mtiles, C, M, sh);
        return;
    }
    // ---- level 1: this group's tiles -> level2[g][2][C] (write-through)
    const int cols = 2 * C;
    double* l2 = t.level2 + (size_t)g * cols;
    for (int cb = 0; cb < cols; cb += NT) {
        const int ncol = cols - cb < NT ? cols - cb : NT;
        tail_colsum<NT, float>(partial + (size_t)t0 * cols + cb, (size_t)cols, t1 - t0, ncol, sh);
        if (tid < ncol)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(l2 + cb + tid), (unsigned long long)__double_as_longlong(sh[tid]),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(t.counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = ticket + 1u == (unsigned)ngroups;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(t.counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *flag = last ? 1 : 0;
    }
    __syncthreads();
    if (*flag == 0) return;
    tail_finish<NT, double>(t, t.level2, ngroups, C, M, sh);
}

// host side: validate and copy a dsnt_bn_tail into its device-side image (null = no tail)
static inline int bn_tail_fill(BnTailP& out, const dsnt_bn_tail* in, const char* who) {
    static_assert(sizeof(BnTailP) == sizeof(dsnt_bn_tail), "dsnt_bn_tail layout");
    out.counters = nullptr;
    out.amax = in ? reinterpret_cast<unsigned*>(in->amax) : nullptr;
    out.amax_bn = in ? reinterpret_cast<unsigned*>(in->amax_bn) : nullptr;
    out.amax_scale = in ? in->amax_scale : nullptr; out.amax_shift = in ? in->amax_shift : nullptr;
    out.amax_relu = in ? in->amax_relu : 0; out.pad_ = 0;
    DSNT_REQUIRE(!out.amax_bn || (out.amax_scale && out.amax_shift && dsnt_aligned16(out.amax_scale) && dsnt_aligned16(out.amax_shift)),
                 DSNT_ERR_ARG, "%s: dsnt_bn_tail.amax_bn needs 16-byte aligned amax_scale / amax_shift", who);
    if (!in || !in->counters) return DSNT_OK;
    DSNT_REQUIRE(in->level2 && in->out2 && (in->mode == 0 || in->mode == 1), DSNT_ERR_ARG, "%s: incomplete dsnt_bn_tail", who);
    DSNT_REQUIRE(in->mode == 1 || (in->out0 && in->out1 && in->out3), DSNT_ERR_ARG,
                 "%s: dsnt_bn_tail mode 0 needs mean / invstd / scale / shift", who);
    DSNT_REQUIRE((in->running_mean == nullptr) == (in->running_var == nullptr), DSNT_ERR_ARG,
                 "%s: dsnt_bn_tail running_mean/var must be given together", who);
    out.mode = in->mode; out.accumulate = in->accumulate; out.counters = in->counters; out.level2 = in->level2;
    out.gamma = in->gamma; out.beta = in->beta; out.rmean = in->running_mean; out.rvar = in->running_var;
    out.momentum = in->momentum; out.eps = in->eps;
    out.o0 = in->out0; out.o1 = in->out1; out.o2 = in->out2; out.o3 = in->out3;
    return DSNT_OK;
}
