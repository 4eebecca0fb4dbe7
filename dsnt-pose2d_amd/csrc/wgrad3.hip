// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions (hourglass.py:22-23, the conv2 of every Bottleneck) on the
// fp16 matrix cores, fp16x3 split (conv.hip: "fp16x3"), as a HALO kernel:
//
//   dW[n][tap][c] = sum over pixels  A[pixel + tap][c] * dY[pixel][n],   A = relu(bn(x)) (zero outside the image)
//
// The implicit-GEMM weight gradient (conv.hip) gives each (tap, 128-channel) k-tile its own workgroup: every input
// element is BatchNorm-transformed, split and TRANSPOSED nine times, every dY element nine times too — 12 VALU
// instructions per MFMA on a kernel whose bound is the SIMD's issue port.  Here a workgroup owns 64 input channels x
// ALL nine taps x 128 (or 64) output channels for a strip of pixels:
//   * the input rows it needs live in a four-slot LDS ring, pixel-major [pixel][32 channels] fp16 planes, written ONCE
//     per element (BN + ReLU + split, no transposition: a float4 of four channels becomes one 8-byte LDS store per
//     plane).  The MFMA contracts over pixels, so its operands are columns of that image: ds_read_b64_tr_b16 delivers
//     them transposed for free, and a filter tap is only an address offset of (r * slot + s) pixels — 64-byte rows,
//     any shift keeps the alignment;
//   * dY never touches LDS: a wave's B operand (32 output channels x 16 pixels) is eight dword loads per lane straight
//     from global memory (lanes = consecutive channels: 128-byte segments), scaled and split in registers once per
//     16-pixel step and reused by all nine taps (27 MFMAs);
//   * one barrier per output ROW (4 x 27 MFMAs per wave at W = 64), the next input row is staged while the current
//     one is consumed.
// Eight waves: wave = (channel half kc, 32-column tile nt[, tap group]), 9 (or 5 / 4) accumulator tiles each.
// Slabs ws[slab][Cout][K] exactly as the other weight-gradient kernels write them (reduced by wgrad_reduce_*).
#include "wgrad3.h"
#include "conv_split.h"
#include <stdlib.h>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wg3_u32x4 __attribute__((ext_vector_type(4)));
#define WG3_LDS __attribute__((address_space(3)))

struct Wg3P {
    const float* x; const float* in_scale; const float* in_shift; const float* dy;
    float* ws;
    const float* a_bound; const float* g_bound;
    int in_relu;
    int N, H, W, Cin, Cout, K;
    int strips, rps, hsplits, kchunks, nchunks, nslabs;
};

// one MFMA operand (32 channels x 16 pixels, fp16) from the pixel-major image: two transposed 4-pixel reads
__device__ __forceinline__ f16x8 wg3_tr_frag(WG3_LDS unsigned char* base, int off) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((WG3_LDS s16x4*)(base + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((WG3_LDS s16x4*)(base + off + 4 * 64));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f16x8, v);
}

template <int KC, int NT, int WS, int TAP0, int NTAPS, int STAGE_AT>
__device__ __forceinline__ void wg3_body(const Wg3P& p, unsigned char* smem, const int kc, const int nt,
                                         const bool first_group) {
    constexpr int GS = WS / 16;                       // 16-pixel steps per row
    constexpr int WP = WS + 2;                        // pixels per LDS row (one zero / halo pixel each side)
    constexpr int SUBSZ = WP * 64;                    // bytes: [WP][32 channels] fp16
    constexpr int PLSZ = KC * SUBSZ;                  // one plane of one slot
    constexpr int SLOTSZ = 2 * PLSZ;                  // two planes
    constexpr int UNITS = WS * 8 * KC;                // float4 units of one input row
    constexpr int NTH = 64 * KC * NT * (NTAPS == 9 ? 1 : 2);   // threads of the workgroup
    constexpr int NPASS = (UNITS + NTH - 1) / NTH;
    const unsigned OOB = 0xF0000000u;

    const int tid = threadIdx.x, lane = tid & 63;
    int bid;
    xcd_remap(blockIdx.x, gridDim.x, bid);
    const int kch = bid % p.kchunks; bid /= p.kchunks;
    const int nch = bid % p.nchunks; bid /= p.nchunks;
    const int slab = bid;
    const int strip = slab % p.strips;
    const int hs = (slab / p.strips) % p.hsplits;
    const int img = slab / (p.strips * p.hsplits);
    const int oh0 = hs * p.rps, w0 = strip * WS;
    const int c0 = kch * 32 * KC, n0 = nch * 32 * NT;
    const bool halo_cols = p.strips > 1;

    const float sa = pow2_scale(bound64(p.a_bound)), sg = pow2_scale(bound64(p.g_bound));
    const float lo_valid = p.in_relu ? 0.f : -__builtin_inff();

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * p.Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy), 0, (int)((size_t)p.N * p.H * p.W * p.Cout * 4u), 0x00020000);

    // ---- staging role: thread -> (32-channel sub-tile, pixel, 4-channel chunk) of an input row
    unsigned aoff[NPASS], alds[NPASS];
    bool act[NPASS];
    float4 sc[NPASS], sh[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int u = tid + NTH * i;
        act[i] = u < UNITS;
        const int sub = (u / (WS * 8)) % KC, px = (u >> 3) & (WS - 1), ch4 = u & 7;
        const int c = c0 + 32 * sub + 4 * ch4;
        aoff[i] = (unsigned)((w0 + px) * p.Cin + c) * 4u;
        alds[i] = (unsigned)(sub * SUBSZ + (px + 1) * 64 + ch4 * 8);
        sc[i] = make_float4(sa, sa, sa, sa);
        sh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.in_scale) {
            const float4 a = *reinterpret_cast<const float4*>(p.in_scale + c);
            const float4 b = *reinterpret_cast<const float4*>(p.in_shift + c);
            sc[i] = make_float4(a.x * sa, a.y * sa, a.z * sa, a.w * sa);
            sh[i] = make_float4(b.x * sa, b.y * sa, b.z * sa, b.w * sa);
        }
    }
    // strips of a wider image: the two halo pixels of a row are real neighbours, staged by 16 * KC lanes of wave 0
    const bool hact = halo_cols && tid < 16 * KC;
    const int hside = tid / (8 * KC), hsub = (tid >> 3) % KC, hch4 = tid & 7;
    const int hiw = hside ? w0 + WS : w0 - 1;
    const bool hcol_ok = hact && hiw >= 0 && hiw < p.W;
    const int hc = c0 + 32 * hsub + 4 * hch4;
    const unsigned hoff = (unsigned)(hiw * p.Cin + hc) * 4u;
    const unsigned hlds = (unsigned)(hsub * SUBSZ + (hside ? (WS + 1) * 64 : 0) + hch4 * 8);
    float4 hsc = make_float4(sa, sa, sa, sa), hsh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hact && p.in_scale) {
        const float4 a = *reinterpret_cast<const float4*>(p.in_scale + hc);
        const float4 b = *reinterpret_cast<const float4*>(p.in_shift + hc);
        hsc = make_float4(a.x * sa, a.y * sa, a.z * sa, a.w * sa);
        hsh = make_float4(b.x * sa, b.y * sa, b.z * sa, b.w * sa);
    }

    struct Row { wg3_u32x4 v[NPASS]; wg3_u32x4 hv; bool ok; };
    // loads of input row ih (a row outside the image loads nothing: out-of-range offsets return zeros)
    auto issue = [&](Row& R, const int ih) {
        R.ok = ih >= 0 && ih < p.H;
        const unsigned rowb = (unsigned)((img * p.H + ih) * p.W) * (unsigned)p.Cin * 4u;
#pragma unroll
        for (int i = 0; i < NPASS; ++i)
            R.v[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, (R.ok && act[i]) ? aoff[i] + rowb : OOB, 0, 0);
        if (halo_cols) R.hv = __builtin_amdgcn_raw_buffer_load_b128(xr, (R.ok && hcol_ok) ? hoff + rowb : OOB, 0, 0);
    };
    auto xform = [&](const wg3_u32x4 raw, const float4 s, const float4 b, const float lo, const float hi, uint2& q1,
                     uint2& q2) {
        float4 v = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w));
        v.x = __builtin_amdgcn_fmed3f(fmaf(v.x, s.x, b.x), lo, hi); v.y = __builtin_amdgcn_fmed3f(fmaf(v.y, s.y, b.y), lo, hi);
        v.z = __builtin_amdgcn_fmed3f(fmaf(v.z, s.z, b.z), lo, hi); v.w = __builtin_amdgcn_fmed3f(fmaf(v.w, s.w, b.w), lo, hi);
        split4h(v, q1, q2);
    };
    // BN + ReLU (one median: valid rows clamp to [0 or -inf, +inf), rows / columns outside the image to [0, 0]),
    // split, two 8-byte stores per float4
    auto store = [&](const Row& R, const int slot) {
        const float lo = R.ok ? lo_valid : 0.f, hi = R.ok ? __builtin_inff() : 0.f;
        unsigned char* base = smem + slot * SLOTSZ;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            uint2 q1, q2;
            xform(R.v[i], sc[i], sh[i], lo, hi, q1, q2);
            // (no branch on the thread index where every thread has a unit — W 64 and 32: a divergent `if` ends the basic block and the
            // transform + split of a row can then not be scheduled between the MFMAs of the step it is staged in; conv3s.hip, round 5)
            if (UNITS % NTH == 0 || act[i]) {
                *reinterpret_cast<uint2*>(base + alds[i]) = q1;
                *reinterpret_cast<uint2*>(base + alds[i] + PLSZ) = q2;
            }
        }
        if (halo_cols) {
            uint2 q1, q2;
            const bool ok = R.ok && hcol_ok;
            xform(R.hv, hsc, hsh, ok ? lo_valid : 0.f, ok ? __builtin_inff() : 0.f, q1, q2);
            if (hact) {
                *reinterpret_cast<uint2*>(base + hlds) = q1;
                *reinterpret_cast<uint2*>(base + hlds + PLSZ) = q2;
            }
        }
    };

    // ---- matrix role: lane (r, h) = (row / column within the 32 x 32 tile, pixel half)
    const int lr = lane & 31, lh = lane >> 5;
    // dY: lane reads column n0 + 32 nt + lr of pixels 8 lh + j (j = 0..7) of the 16-pixel step
    const unsigned glane = (unsigned)(8 * lh * p.Cout + n0 + 32 * nt + lr) * 4u;
    const unsigned gstride = (unsigned)p.Cout * 4u;
    auto gissue = [&](float (&G)[8], int t, const int g) {
        t = t < p.rps ? t : p.rps - 1;                      // the one prefetch past the end re-reads the last step
        const unsigned soff = (unsigned)(((img * p.H + oh0 + t) * p.W + w0 + 16 * g) * p.Cout) * 4u;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            G[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gr, glane, soff + j * gstride, 0));
    };
    // A: transposed-read address of this lane inside a slot: lane 16 cb + 4 q + pp of a half supplies pixel row q,
    // channels 16 cb + 4 pp .. +3 and receives channel 16 cb + (lane & 15) of four pixels (cdna guide T10)
    const int cb = (lane >> 4) & 1, q4 = (lane >> 2) & 3, pp = lane & 3;
    const unsigned lane_a = (unsigned)(kc * SUBSZ + (8 * lh + q4) * 64 + (16 * cb + 4 * pp) * 2);
    WG3_LDS unsigned char* lds0 = (WG3_LDS unsigned char*)smem;

    f32x16 acc[NTAPS];
#pragma unroll
    for (int i = 0; i < NTAPS; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float bsum = 0.f;

    // ---- prologue: zero pixels (whole-width strips), input rows oh0-1, oh0, oh0+1 into their slots
    if (!halo_cols) {
        for (int u = tid; u < 4 * 2 * KC * 2 * 8; u += NTH) {
            const int e = u & 7, side = (u >> 3) & 1, rest = u >> 4;          // rest = (slot, plane, sub)
            *reinterpret_cast<uint2*>(smem + rest * SUBSZ + (side ? (WS + 1) * 64 : 0) + e * 8) = make_uint2(0u, 0u);
        }
    }
    {
        Row R0, R1, R2;
        issue(R0, oh0 - 1); issue(R1, oh0); issue(R2, oh0 + 1);
        store(R0, (oh0 + 0) & 3); store(R1, (oh0 + 1) & 3); store(R2, (oh0 + 2) & 3);
    }
    Row RA;
    issue(RA, p.rps > 1 ? oh0 + 2 : -8);
    float Graw[8];
    gissue(Graw, 0, 0);
    __syncthreads();

    for (int t = 0; t < p.rps; ++t) {
        const int oh = oh0 + t;
        // filter row r reads input row oh-1+r = slot (oh + r) & 3
        WG3_LDS unsigned char* ab[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) ab[r] = lds0 + (lane_a + (unsigned)(((oh + r) & 3) * SLOTSZ));
#pragma unroll
        for (int g = 0; g < GS; ++g) {
            // input row oh+2 (loaded during the previous row) -> slot of row oh+2; start the loads of row oh+3.
            // Waves 0..3 do this at the head of the row, waves 4..7 (their partners on the four SIMDs) in the middle:
            // the halves then run half a step apart and one's VALU work meets the other's MFMAs (STAGE_AT)
            if (g == STAGE_AT) {
                if (t + 1 < p.rps) store(RA, (oh + 3) & 3);
                issue(RA, t + 2 < p.rps ? oh + 3 : -8);
            }
            // dY of this step: scale, split into two fp16 planes (element j = pixel 8 lh + j: the MFMA's k order).
            // (only matrix instructions and LDS reads may cross into the previous step: hipcc otherwise hoists these
            // multiplies — and with them the wait for loads issued a moment ago — up under the previous step's MFMAs)
            __builtin_amdgcn_sched_barrier(0x108);
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = Graw[j] * sg;
            // bias partial; every wave keeps it — a branch here would cut the row into basic blocks and hand the
            // placement of the loads below back to hipcc
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum += Graw[j];
            wg3_u32x4 h1, h2;
            {
                uint2 a, b, c, d;
                split4h(make_float4(v[0], v[1], v[2], v[3]), a, b);
                split4h(make_float4(v[4], v[5], v[6], v[7]), c, d);
                h1 = (wg3_u32x4){a.x, a.y, c.x, c.y};
                h2 = (wg3_u32x4){b.x, b.y, d.x, d.y};
            }
            const f16x8 g1 = __builtin_bit_cast(f16x8, h1), g2 = __builtin_bit_cast(f16x8, h2);
            // the next step's dY loads go out NOW and stay in flight under this step's 27 MFMAs (hipcc would sink
            // them to the end of the step, next to their use)
            gissue(Graw, g + 1 < GS ? t : t + 1, g + 1 < GS ? g + 1 : 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ti = 0; ti < NTAPS; ++ti) {
                const int tap = TAP0 + ti, r = tap / 3, s = tap % 3;
                const int off = (16 * g + s) * 64;
                const f16x8 a1 = wg3_tr_frag(ab[r], off), a2 = wg3_tr_frag(ab[r], off + PLSZ);
                acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, g1, acc[ti], 0, 0, 0);
                acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, g2, acc[ti], 0, 0, 0);
                acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, g1, acc[ti], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- slab store: ws[slab][n][k], D row = channel (registers, 4 consecutive), D column = n (lane); both operand
    // scales are powers of two and are undone exactly
    const float osc = 1.f / (sa * sg);
    const int n = n0 + 32 * nt + lr;
    float* slabp = p.ws + (size_t)slab * p.Cout * p.K + (size_t)n * p.K + c0 + 32 * kc + 4 * lh;
#pragma unroll
    for (int ti = 0; ti < NTAPS; ++ti) {
        const int tap = TAP0 + ti;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
            *reinterpret_cast<float4*>(slabp + tap * p.Cin + 8 * qq) =
                make_float4(acc[ti][4 * qq + 0] * osc, acc[ti][4 * qq + 1] * osc, acc[ti][4 * qq + 2] * osc,
                            acc[ti][4 * qq + 3] * osc);
    }
    // bias partial of this slab: column sums of dY (first channel chunk, first tap group only)
    if (first_group && kch == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (lh == 0) p.ws[(size_t)p.nslabs * p.Cout * p.K + (size_t)slab * p.Cout + n] = bsum;
    }
}

// CFG 0: 64 input channels x 128 output channels per workgroup, wave = (kc, nt), nine taps each;
// CFG 1: 64 x 64, wave = (kc, nt, tap group): taps 0..4 / 5..8;
// CFG 2: 32 x 128 with FOUR waves (wave = nt): half the registers of a CU, so that a launch beside a dependency chain
//        on another stream (DSNT_WGRAD_SHARE_CHIP) leaves room for the chain's kernels on every CU
template <int CFG, int WS>
__global__ __launch_bounds__(CFG == 2 ? 256 : 512, 2) void wgrad3_kernel(Wg3P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wg3_smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef WG3_NO_STAGGER                  // A/B: both halves stage at the head of the row
    constexpr int MID = 0;
#else
    constexpr int MID = WS / 32;       // half a row of 16-pixel steps
#endif
    if (CFG == 0) {
        if (wave < 4) wg3_body<2, 4, WS, 0, 9, 0>(p, wg3_smem, wave & 1, wave >> 1, (wave & 1) == 0);
        else wg3_body<2, 4, WS, 0, 9, MID>(p, wg3_smem, wave & 1, wave >> 1, (wave & 1) == 0);
    } else if (CFG == 1) {
        const int kc = wave & 1, nt = (wave >> 1) & 1;
        if ((wave >> 2) == 0) wg3_body<2, 2, WS, 0, 5, 0>(p, wg3_smem, kc, nt, kc == 0);
        else wg3_body<2, 2, WS, 5, 4, MID>(p, wg3_smem, kc, nt, false);
    } else {
        wg3_body<1, 4, WS, 0, 9, 0>(p, wg3_smem, 0, wave, true);
    }
}

static int enabled = -1, min_steps = 0, small_wg = 1;

Wg3Plan dsnt_wg3_plan(const dsnt_conv_geom* g, int share) {
    Wg3Plan pl;
    memset(&pl, 0, sizeof(pl));
    // tuning constants (measured on MI355X, hg2 batch 32): 256 workgroups = one per CU; at least 32 16-pixel steps per
    // workgroup (every workgroup pays a three-row prologue and writes a whole slab); four-wave workgroups for launches
    // that share the chip
    const long target = 256;
    if (enabled < 0) {
        enabled = dsnt_kernel_off("wgrad3") ? 0 : 1;
        min_steps = 32;
        small_wg = 1;
    }
    if (!enabled || !g) return pl;
    if (!(g->R == 3 && g->S == 3 && g->stride == 1 && g->pad == 1 && g->dil == 1 && g->Ho == g->H && g->Wo == g->W))
        return pl;
    if (g->Cin % 64 != 0 || g->Cout % 64 != 0) return pl;
    const int WS = g->W >= 64 ? 64 : g->W;
    if (!(WS == 16 || WS == 32 || WS == 64) || g->W % WS != 0) return pl;
    if ((size_t)g->N * g->H * g->W * g->Cin * 4u >= (1ull << 31) || (size_t)g->N * g->H * g->W * g->Cout * 4u >= (1ull << 31))
        return pl;
    pl.cfg = g->Cout % 128 == 0 ? 0 : 1;
    pl.WS = WS;
    pl.strips = g->W / WS;
    pl.kchunks = g->Cin / 64;
    pl.nchunks = g->Cout / (pl.cfg == 1 ? 64 : 128);
    // rows per slab: halve (while it stays a divisor of H) until the launch has `target` workgroups, but keep at least
    // `min_steps` 16-pixel steps per workgroup: every workgroup pays a three-row prologue and writes a whole slab
    int rps = g->H;
    const long per = (long)pl.kchunks * pl.nchunks * pl.strips * g->N;
    // (a launch that shares the chip runs as four-wave workgroups of 32 channels, ONE per CU: twice the workgroups per
    // slab, so half as many slabs fill the chip — and half the slab bytes are written and reduced)
    // share == 2 (DSNT_WGRAD_NARROW): half as many slabs again — 128 four-wave workgroups.  In the middle of backward the
    // weight-gradient lane has slack since the 1x1 weight gradients left it (bwd1.hip), and a 3x3 weight gradient on half of
    // the CUs takes less from the dependency chain's kernels beside it (same box, hg2 batch 32: 12.53 -> 12.27 ms/step,
    // 12.09 -> 11.93 on a second one; and half the slab bytes); NOT at the end of backward, where nothing runs beside the
    // stem's weight gradients and their duration is the step's (hg8 batch 16 with every launch narrow: +0.4 ms)
    long want = (share && small_wg && pl.cfg == 0) ? target / 2 : target;
    if (share == 2) want /= 2;
    while (per * (g->H / rps) < want && rps % 2 == 0 && (rps / 2) * (WS / 16) >= min_steps) rps /= 2;
    pl.rps = rps;
    pl.hsplits = g->H / rps;
    pl.nslabs = g->N * pl.hsplits * pl.strips;
    pl.blocks = pl.kchunks * pl.nchunks * pl.nslabs;
    pl.lds = 4 * 2 * 2 * (WS + 2) * 64;
    pl.ok = 1;
    return pl;
}

template <int CFG, int WS>
static void wg3_launch_cfg(const Wg3Plan& pl, const Wg3P& p, hipStream_t st) {
    // CFG 2 (a launch that shares the chip): more than half of the LDS keeps it to ONE four-wave workgroup per CU —
    // half of every CU's registers stay free for the dependency chain's kernels on the other streams
    const int share_lds = 84 * 1024;
    const int lds = CFG == 2 ? share_lds : pl.lds;
    const int blocks = CFG == 2 ? 2 * pl.blocks : pl.blocks;       // 32 instead of 64 input channels per workgroup
    DSNT_SET_MAX_LDS((wgrad3_kernel<CFG, WS>), lds);
    DSNT_LAUNCH((wgrad3_kernel<CFG, WS>), dim3(blocks), dim3(CFG == 2 ? 256 : 512), lds, st, p);
}

void dsnt_wg3_launch(const Wg3Plan& pl, const float* x, const float* in_scale, const float* in_shift, int in_relu,
                     const float* dy, float* ws, const float* a_bound, const float* g_bound, const dsnt_conv_geom* g,
                     hipStream_t st, int share) {
    Wg3P p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.in_scale = in_scale; p.in_shift = in_shift; p.dy = dy; p.ws = ws;
    p.a_bound = a_bound; p.g_bound = g_bound; p.in_relu = in_scale ? in_relu : 0;
    p.N = g->N; p.H = g->H; p.W = g->W; p.Cin = g->Cin; p.Cout = g->Cout; p.K = 9 * g->Cin;
    p.strips = pl.strips; p.rps = pl.rps; p.hsplits = pl.hsplits; p.kchunks = pl.kchunks; p.nchunks = pl.nchunks;
    p.nslabs = pl.nslabs;
    const bool small = share && small_wg && pl.cfg == 0;
    if (small) p.kchunks = 2 * pl.kchunks;
#define WG3_PICK(CFG)                                                   \
    do {                                                                \
        if (pl.WS == 64) wg3_launch_cfg<CFG, 64>(pl, p, st);            \
        else if (pl.WS == 32) wg3_launch_cfg<CFG, 32>(pl, p, st);       \
        else wg3_launch_cfg<CFG, 16>(pl, p, st);                        \
    } while (0)
    if (small) WG3_PICK(2);
    else if (pl.cfg == 0) WG3_PICK(0);
    else WG3_PICK(1);
#undef WG3_PICK
}
