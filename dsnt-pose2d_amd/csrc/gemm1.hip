// 1x1 convolutions of the full-resolution levels (hourglass.py:20-25: conv1 / conv3 of every Bottleneck, forward and as
// data gradients; :104-148 the lin / score convolutions) on the fp16 matrix cores, fp16x3 split, as a STREAMING kernel.
//
// These launches are HBM-bound by a wide margin (128->256 @64x64, batch 32: 335 MB per launch, 8.6 GFLOP = 10 us of
// matrix pipe) — what the tiled implicit-GEMM kernel (conv.hip) loses on them is memory-level parallelism and issue
// slots: a 128 x 128 tile lives ~46 k cycles for 3 k cycles of MFMA, its loader waves hold two 8 KB K-steps in
// flight, every A element is transformed once per 128-column tile and every output element crosses LDS to be
// re-shaped for 16-byte stores.  Here:
//   * the WEIGHTS of the workgroup's column chunk (all of K, two fp16 planes: <= 139 KB) are copied into LDS once and
//     stay there; a workgroup is persistent over row blocks;
//   * the ACTIVATIONS never touch LDS: a lane's MFMA operand is 8 consecutive channels of one row = two 16-byte loads
//     straight from global memory (the k order inside a 16-wide step is permuted so that the two halves of a wave read
//     one contiguous 32 bytes per row; the weight image in LDS carries the same permutation).  BatchNorm + ReLU + the
//     exact split run in registers once per element for ALL columns.  Each wave keeps 4 K-steps of loads in flight:
//     64-128 KB per CU, with no barrier anywhere in the loop — the eight waves drift apart and cover each other;
//   * the EPILOGUE works from the MFMA result layout as it stands: a register is 2 rows x 32 consecutive columns, i.e.
//     two 128-byte runs — residuals are loaded and results stored as full lines by dword accesses, bias / BatchNorm
//     vectors are per-lane scalars (lane = column), column sums need no transposition.
// Same contract as conv_fwd_bf16x6_kernel<..., F16> (conv.hip): BN+ReLU prologue, bias, two residuals, per-128-row
// column statistics, the BatchNorm-backward epilogue of data-gradient launches, the bounds of dsnt_out_bounds.
#include "gemm1.h"
#include <stdlib.h>
#include <type_traits>

typedef unsigned g1_u32x4 __attribute__((ext_vector_type(4)));

// MODE (the epilogue, one per instantiation: a kernel that carries all of them spills): 0 no residual, 1 res1,
// 3 the BatchNorm-backward epilogue of a data-gradient launch (res1 = the BatchNorm input x)
template <int KS, int NTW, bool PRO, int MODE>
__global__ __launch_bounds__(512, 2) void gemm1_kernel(ConvP p, int niter) {
    constexpr int RT = 8 / NTW;                     // 32-row tiles per wave
    constexpr int K = KS * 16;
    constexpr int CHUNK = NTW * 32;                 // output columns per workgroup
    constexpr int ROWS_W = RT * 32, ROWS_IT = 8 * ROWS_W;
    constexpr int PB = KS * 32 + 16;                // bytes per weight row in LDS: +16 keeps ds_read_b128 conflict-free
    constexpr int PLB = CHUNK * PB;
    constexpr int D = RT == 1 ? 4 : (RT == 2 ? 2 : 1);   // K-steps of activation loads in flight per wave (32 registers)
    constexpr int DD = D < KS ? D : KS;
    constexpr int TPI = ROWS_IT / 128;              // 128-row statistics tiles per iteration
    constexpr int WPT = 8 / TPI;                    // waves per statistics tile
    const unsigned OOB = 0xF0000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char g1_smem[];
    unsigned char* Bs = g1_smem;                                    // [2 planes][CHUNK][PB]
    float* SS = reinterpret_cast<float*>(g1_smem + 2 * PLB);        // [2][K]: BN scale / shift x operand scale
    float* red = SS + 2 * K;                                        // [8 waves][CHUNK][2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.y * CHUNK;
    const float sa = pow2_scale(bound64(p.a_bound)), sw = pow2_scale(bound64(p.w_bound));
    const float osc = 1.f / (sa * sw);

    // ---- the first K-steps of this workgroup's first row block go out before anything else: they travel while the weights are copied
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.M * K * 4u), 0x00020000);
    int it = blockIdx.x;
    unsigned voff[RT];
    g1_u32x4 raw[DD][RT][2];
    auto rows_of = [&](const int iter) {                 // this lane's rows of row block `iter` (none: loads return zeros)
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int m = iter * ROWS_IT + wave * ROWS_W + t * 32 + lr;
            voff[t] = (iter < niter && m < p.M) ? (unsigned)(m * K + 4 * lh) * 4u : OOB;
        }
    };
    auto issue = [&](const int slot, const int s) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            raw[slot][t][0] = __builtin_amdgcn_raw_buffer_load_b128(xr, voff[t] + 64u * s, 0, 0);
            raw[slot][t][1] = __builtin_amdgcn_raw_buffer_load_b128(xr, voff[t] + 64u * s + 32u, 0, 0);
        }
    };
    rows_of(it);
#pragma unroll
    for (int s = 0; s < DD; ++s) issue(s, s);


    // ---- weights -> LDS (k permuted inside each 16-wide step: positions [0..3, 8..11 | 4..7, 12..15])
    // (all of a thread's 16-byte loads in flight at once — the copy is 2 * CHUNK * K * 2 bytes = up to 139 KB per workgroup,
    // 4 .. 16 loads per thread — then the LDS stores: one L2 round trip instead of one per load)
    {
        constexpr int NW = 2 * CHUNK * KS * 2 / 512;
        static_assert(NW * 512 == 2 * CHUNK * KS * 2, "weight units per thread");
        uint4 wv[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int u = tid + 512 * j;
            const int half8 = u & 1, s = (u >> 1) % KS, n = (u / (2 * KS)) % CHUNK, pl = u / (2 * KS * CHUNK);
            wv[j] = *reinterpret_cast<const uint4*>(p.wq + (size_t)pl * p.wq_stride + (size_t)(n0 + n) * K + 16 * s + 8 * half8);
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int u = tid + 512 * j;
            const int half8 = u & 1, s = (u >> 1) % KS, n = (u / (2 * KS)) % CHUNK, pl = u / (2 * KS * CHUNK);
            unsigned char* d = Bs + pl * PLB + n * PB + s * 32 + half8 * 8;      // k-quad q = 2 half8 -> position 4 half8 ...
            *reinterpret_cast<uint2*>(d) = make_uint2(wv[j].x, wv[j].y);
            *reinterpret_cast<uint2*>(d + 16) = make_uint2(wv[j].z, wv[j].w);    // ... q + 1 -> position 8 + 4 half8
        }
    }
    if (PRO) {
        for (int k = tid; k < K; k += 512) {
            SS[k] = p.in_scale[k] * sa;
            SS[K + k] = p.in_shift[k] * sa;
        }
    }
    const float relu_lo = (PRO && p.in_relu) ? 0.f : -__builtin_inff();
    // epilogue tensors through buffer descriptors: 32-bit lane offsets, the 16 row offsets of a tile as scalars
    const int ybytes = (int)((size_t)p.M * p.Cout * 4u);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res1 ? p.res1 : p.y), 0, ybytes, 0x00020000);
    const unsigned rowbytes = (unsigned)p.Cout * 4u;
    const float am2lo = p.tail.amax_relu ? 0.f : -__builtin_inff();
    float am = 0.f, am2 = 0.f;
    __syncthreads();

    for (; it < niter; it += gridDim.x) {
        const int rowbase = it * ROWS_IT + wave * ROWS_W;
        // LDS offsets of this lane, laundered once per row block: the reads below are invariant across row blocks and
        // hipcc would otherwise hoist (and keep in registers) every BatchNorm vector and weight fragment of the K loop
        int ss_lane = 4 * lh, b_lane = lr * PB + lh * 16;
        asm volatile("" : "+v"(ss_lane), "+v"(b_lane));

        f32x16 acc[RT][NTW];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][nt][e] = 0.f;

#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int slot = s % DD;
            __builtin_amdgcn_sched_barrier(0x108);          // VALU / waits of this step stay below the previous step's MFMAs
            // activations of this step: BN + ReLU (or just the operand scale), exact split into two fp16 planes
            f16x8 a1[RT], a2[RT];
            float4 c0 = make_float4(sa, sa, sa, sa), c1 = c0, h0 = make_float4(0.f, 0.f, 0.f, 0.f), h1 = h0;
            if (PRO) {
                c0 = *reinterpret_cast<const float4*>(SS + 16 * s + ss_lane);
                c1 = *reinterpret_cast<const float4*>(SS + 16 * s + 8 + ss_lane);
                h0 = *reinterpret_cast<const float4*>(SS + K + 16 * s + ss_lane);
                h1 = *reinterpret_cast<const float4*>(SS + K + 16 * s + 8 + ss_lane);
            }
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const g1_u32x4 r0 = raw[slot][t][0], r1 = raw[slot][t][1];
                float4 v0 = make_float4(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z), __uint_as_float(r0.w));
                float4 v1 = make_float4(__uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z), __uint_as_float(r1.w));
                if (PRO) {
                    v0.x = fmaxf(fmaf(v0.x, c0.x, h0.x), relu_lo); v0.y = fmaxf(fmaf(v0.y, c0.y, h0.y), relu_lo);
                    v0.z = fmaxf(fmaf(v0.z, c0.z, h0.z), relu_lo); v0.w = fmaxf(fmaf(v0.w, c0.w, h0.w), relu_lo);
                    v1.x = fmaxf(fmaf(v1.x, c1.x, h1.x), relu_lo); v1.y = fmaxf(fmaf(v1.y, c1.y, h1.y), relu_lo);
                    v1.z = fmaxf(fmaf(v1.z, c1.z, h1.z), relu_lo); v1.w = fmaxf(fmaf(v1.w, c1.w, h1.w), relu_lo);
                } else {
                    v0.x *= sa; v0.y *= sa; v0.z *= sa; v0.w *= sa;
                    v1.x *= sa; v1.y *= sa; v1.z *= sa; v1.w *= sa;
                }
                uint2 p0, q0, p1, q1;
                split4h(v0, p0, q0);
                split4h(v1, p1, q1);
                a1[t] = __builtin_bit_cast(f16x8, (g1_u32x4){p0.x, p0.y, p1.x, p1.y});
                a2[t] = __builtin_bit_cast(f16x8, (g1_u32x4){q0.x, q0.y, q1.x, q1.y});
            }
            if (s + DD < KS) issue(slot, s + DD);           // refill the slot: stays in flight for DD steps
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* brow = Bs + b_lane + s * 32;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const f16x8 b1 = *reinterpret_cast<const f16x8*>(brow + nt * 32 * PB);
                const f16x8 b2 = *reinterpret_cast<const f16x8*>(brow + nt * 32 * PB + PLB);
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[t], b1, acc[t][nt], 0, 0, 0);
                    acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[t], b2, acc[t][nt], 0, 0, 0);
                    acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[t], b1, acc[t][nt], 0, 0, 0);
                }
            }
        }
        // the next row block's first K-steps go out before this block's epilogue: its loads travel while the
        // residuals are read and the results stored
        rows_of(it + (int)gridDim.x);
#pragma unroll
        for (int s = 0; s < DD; ++s) issue(s, s);
        __builtin_amdgcn_sched_barrier(0);

        // ---- epilogue from the C layout: register e of a tile = rows (e&3) + 8 (e>>2) + 4 lh, column lr.
        // Software-pipelined over the wave's eight 32 x 32 tiles: the 16 residual loads of tile i+1 are issued BEFORE the
        // stores of tile i — loads and stores retire in order, so a batch that waits for its own loads behind the
        // previous batch's stores pays two memory round trips per tile (measured: 43 us per row block).
        {
            constexpr int T = NTW * RT, NR = 16;
            float rbuf[2][NR];
            // byte offset of this lane's element of register 0 of tile i; register e adds rowoff[e] (a scalar)
            auto tile_off = [&](const int i) {
                return (unsigned)((rowbase + (i % RT) * 32 + 4 * lh) * p.Cout + n0 + (i / RT) * 32 + lr) * 4u;
            };
            auto loadres = [&](float (&r)[NR], const int i) {
                if (MODE == 0) return;
                const unsigned o0 = tile_off(i);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned so = (unsigned)((e & 3) + 8 * (e >> 2)) * rowbytes;
                    r[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1r, o0, so, 0));
                }
            };
            struct Col { float cb, sc, sh, mu, is; };
            auto loadcol = [&](const int nt) {
                const int n = n0 + nt * 32 + lr;
                Col c = {0.f, 0.f, 0.f, 0.f, 0.f};
                if (MODE == 3) { c.sc = p.bnb_scale[n]; c.sh = p.bnb_shift[n]; c.mu = p.bnb_mean[n]; c.is = p.bnb_invstd[n]; }
                else {
                    if (p.bias) c.cb = p.bias[n];
                    if (p.tail.amax_bn) { c.sc = p.tail.amax_scale[n]; c.sh = p.tail.amax_shift[n]; }
                }
                return c;
            };
            Col col = loadcol(0), coln = col;
            loadres(rbuf[0], 0);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const int nt = i / RT, t = i % RT;
                if (i + 1 < T) {
                    loadres(rbuf[(i + 1) & 1], i + 1);
                    if ((i + 1) % RT == 0) coln = loadcol((i + 1) / RT);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float (&r)[NR] = rbuf[i & 1];
                const unsigned o0 = tile_off(i);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned so = (unsigned)((e & 3) + 8 * (e >> 2)) * rowbytes;
                    float v = acc[t][nt][e] * osc;
                    if (MODE == 3) {
                        // v = dL/d relu(bn(x)); r = x: mask by the ReLU, accumulate the BatchNorm-backward sums
                        const float xv = r[e];
                        if (p.bnb_relu && fmaf(xv, col.sc, col.sh) <= 0.f) v = 0.f;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, o0, so, 0);
                        s1 += v;
                        s2 = fmaf(v, (xv - col.mu) * col.is, s2);
                        am = fmaxf(am, fabsf(v));            // max |dz| (p.tail.amax)
                    } else {
                        v += col.cb;
                        if (MODE >= 1) v += r[e];
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, o0, so, 0);
                        am = fmaxf(am, fabsf(v));
                        if (p.tail.amax_bn) am2 = fmaxf(am2, fabsf(fmaxf(fmaf(v, col.sc, col.sh), am2lo)));
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                    }
                }
                if (t == RT - 1) {
                    if (p.stats) {
                        s1 += __shfl_xor(s1, 32, 64);
                        s2 += __shfl_xor(s2, 32, 64);
                        if (lh == 0) {
                            red[(wave * CHUNK + nt * 32 + lr) * 2 + 0] = s1;
                            red[(wave * CHUNK + nt * 32 + lr) * 2 + 1] = s2;
                        }
                    }
                    s1 = 0.f; s2 = 0.f;
                    col = coln;
                }
            }
        }
        if (p.stats) {
            __syncthreads();
            for (int u = tid; u < TPI * CHUNK * 2; u += 512) {
                const int which = u & 1, c = (u >> 1) % CHUNK, j = u / (2 * CHUNK);
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < WPT; ++w) a += red[((j * WPT + w) * CHUNK + c) * 2 + which];
                tail_store(p.stats + ((size_t)(it * TPI + j) * 2 + which) * p.Cout + n0 + c, a);
            }
            __syncthreads();
        }
    }
    if (p.tail.amax) amax_commit(am, p.tail.amax);
    if (p.tail.amax_bn) amax_commit(am2, p.tail.amax_bn, 1);
}

static int g1_enabled = -1;
static const long g1_min_rows = 65536;     // below: too few row blocks to fill 256 persistent workgroups

int dsnt_gemm1_cfg(const ConvP& p) {
    if (g1_enabled < 0) {
        g1_enabled = dsnt_kernel_off("gemm1") ? 0 : 1;
    }
    if (!g1_enabled || !p.a_bound || !p.w_bound || !p.wq) return -1;
    if (!(p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0 && p.Ho == p.H && p.Wo == p.W)) return -1;
    if (p.res2) return -1;                                      // two residuals (score_ joins): the tiled kernel
    if (p.bnb_scale && (p.in_scale || p.res2)) return -1;
    if (p.M < g1_min_rows) return -1;
    if ((size_t)p.M * p.K * 4u >= (1ull << 31) || (size_t)p.M * p.Cout * 4u >= (1ull << 31)) return -1;
    // (K, columns per workgroup): the weight chunk has to fit LDS beside the statistics scratch
    int ntw;
    if (p.K == 128) ntw = p.Cout % 256 == 0 ? 8 : (p.Cout % 128 == 0 ? 4 : (p.Cout % 64 == 0 ? 2 : 0));
    else if (p.K == 256) ntw = p.Cout % 128 == 0 ? 4 : 0;
    else if (p.K == 64) ntw = p.Cout % 128 == 0 ? 4 : (p.Cout % 64 == 0 ? 2 : 0);
    else return -1;
    if (!ntw) return -1;
    const int rows_it = 8 * (8 / ntw) * 32;
    if (p.M % rows_it != 0) return -1;
    return ntw;
}

template <int KS, int NTW, bool PRO, int MODE>
static void g1_launch_k(const ConvP& p, hipStream_t st, bool share) {
    constexpr int K = KS * 16, CHUNK = NTW * 32, PB = KS * 32 + 16;
    const int lds = 2 * CHUNK * PB + 2 * K * 4 + 8 * CHUNK * 2 * 4;
    DSNT_SET_MAX_LDS((gemm1_kernel<KS, NTW, PRO, MODE>), lds);
    const int rows_it = 8 * (8 / NTW) * 32;
    const int niter = p.M / rows_it;
    const int chunks = p.Cout / CHUNK;
    const int cus = dsnt_device_cus();
    // one workgroup per CU (LDS), persistent over its share of the row blocks; column chunks side by side
    int gx = cus / chunks;
    // DSNT_CONV_SHARE_CHIP: a launch on a side lane.  One of these workgroups takes 139 of a CU's 160 KB of LDS for the whole
    // (persistent) launch, and the dependency chain's own launches of this kernel need a CU's LDS to themselves: on half of the
    // CUs the side lane still finishes in time (skip branches have slack) and the chain starts at once.  Same box, ms/step:
    // 13.83 all CUs, 13.80 at 62 %, 13.69 at 50 %, 13.71 at 38 %, 13.67 at 25 %.
    if (share) gx = gx / 2 > 0 ? gx / 2 : 1;
    if (gx < 1) gx = 1;
    if (gx > niter) gx = niter;
    DSNT_LAUNCH((gemm1_kernel<KS, NTW, PRO, MODE>), dim3(gx, chunks), dim3(512), lds, st, p, niter);
}

template <int KS, int NTW>
static void g1_launch(const ConvP& p, bool pro, hipStream_t st, bool share) {
    if (p.bnb_scale) g1_launch_k<KS, NTW, false, 3>(p, st, share);         // data gradient with the BatchNorm-backward epilogue
    else if (pro) {
        if (p.res1) g1_launch_k<KS, NTW, true, 1>(p, st, share);
        else g1_launch_k<KS, NTW, true, 0>(p, st, share);
    } else if (p.res1) g1_launch_k<KS, NTW, false, 1>(p, st, share);
    else g1_launch_k<KS, NTW, false, 0>(p, st, share);
}

void dsnt_gemm1_launch(const ConvP& p, int ntw, bool pro, hipStream_t st, bool share) {
    if (p.K == 128) {
        if (ntw == 8) g1_launch<8, 8>(p, pro, st, share);
        else if (ntw == 4) g1_launch<8, 4>(p, pro, st, share);
        else g1_launch<8, 2>(p, pro, st, share);
    } else if (p.K == 256) {
        g1_launch<16, 4>(p, pro, st, share);
    } else {
        if (ntw == 4) g1_launch<4, 4>(p, pro, st, share);
        else g1_launch<4, 2>(p, pro, st, share);
    }
}
