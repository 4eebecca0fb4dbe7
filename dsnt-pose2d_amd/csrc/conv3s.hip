// 3x3 / stride 1 / pad 1 convolutions of the full-resolution levels (hourglass.py:22-23: conv2 of every Bottleneck, forward
// and as data gradients) on the fp16 matrix cores, fp16x3 split — the SYMMETRIC successor of conv3x3_bf16x6_kernel (conv.hip).
//
// What that kernel lost (profiles/r02_timeline_halo.txt, r02_pmc_issue_accounting.txt): its four loader waves share the
// SIMDs with the four MFMA waves, lose the issue arbitration against them and arrive last at 80 % of the per-K-step
// barriers (803 of 1993 cycles per step are barrier wait on the MFMA waves); prologue and epilogue (LDS transposition of
// the 128 x 128 result) are another 18 k cycles per tile with the matrix pipe idle.  Here:
//   * four waves, ALL alike: every wave copies its share of the weights and of the input halo between its own MFMAs
//     (an MFMA holds the vector issue port for 8 of its 32 cycles — the staging work of a step fits beside them);
//   * a tile is a 4 x 32 patch (8 x 16 where the image is 16 pixels wide) of output pixels x all Cout (64 / 128) columns; the
//     (4+2) x (32+2) input halo of a 16-channel chunk is transformed (BatchNorm + ReLU, operand scale) and split ONCE into LDS, double-buffered;
//   * the weights come pre-split in STREAM order ([chunk][tap][Cout][16]: dsnt_f16_prep_weights, row flag): a K-step
//     pair is one contiguous 16 KB block, copied with coalesced 16-byte loads one pair ahead into a two-slot ring;
//   * ONE barrier per pair of K-steps (24 MFMAs per wave), placed between the two steps: the fragments of the step
//     after the barrier are read while the MFMAs of the step before it run, so no fragment read is exposed;
//   * workgroups are persistent: the weight stream wraps around and the next tile's first halo chunk is staged by the
//     regular schedule of the last chunk pair — a tile has no prologue; the epilogue works from the MFMA result layout
//     (a register = 2 pixels x 32 consecutive channels = two 128-byte runs; no LDS transposition), residual loads
//     software-pipelined over the wave's tiles (as gemm1.hip).
// Same K order (16-channel chunk, tap) and MFMA order as conv3x3_bf16x6_kernel<.., F16>: the convolution sums of the 32x32x16
// form are bit-identical to that kernel's; the per-tile column statistics are summed in another order.
// Round 5: Cout 128 on 4 x 32 patches (every 3x3 of the hourglass) runs on v_mfma_f32_16x16x32_f16 instead (MF16, C3Geo below): the
// chip holds a higher clock under the 4-pass MFMA (MI355X_MICROARCH "DVFS give-back"), an MFMA covers both K-steps of a pair, and the
// LDS image needs no padding (60 KB per workgroup instead of 80).  Same products, same (chunk, tap) order of the pairs; inside a pair
// the two steps are one 32-deep dot product, so the sums agree with the 32x32x16 form to fp32 rounding, not bit for bit.
// Measured (profiles/r05_ab_switches.txt box J, r05_pmc_issue_accounting.txt last section): forward 92-95 us against 104-108 on the same
// boxes — 192 k cycles at 2.08 GHz against 205 k at 1.90 GHz, the matrix pipe busy for the same 54-58 % of them: the form buys clock.
// In the hg2 / hg8 steps the difference stays inside the run-to-run spread.
// Contract: conv_fwd_bf16x6_kernel<..., F16> minus the second residual (dsnt_conv3s_ok).
// Measured (DESIGN.md "round 3", profiles/r03_pmc_issue_accounting.txt): 3x3 128->128 @64x64, batch 32: 129 -> 104 us on one box
// (matrix pipe 53 %, 1.15 PFLOP/s of fp16 MFMA at the ~1.9 GHz this load holds), -0.42 ms per hg2 step.  Built on the same pieces,
// bit-identical, and dropped: an 8-wave ping-pong form (two groups one phase apart, 101-110 us, but one 124 KB workgroup per CU
// starves the other lanes: +0.07 ms per step), deeper halo prefetch (108 us), a start-up stagger of the CU's second workgroup (+-0).
#include "conv3s.h"
#include <stdlib.h>

typedef unsigned c3_u32x4 __attribute__((ext_vector_type(4)));
typedef float c3_f32x4 __attribute__((ext_vector_type(4)));
typedef int c3_i32x4 __attribute__((ext_vector_type(4)));
#ifndef C3_MF16_DEFAULT
#define C3_MF16_DEFAULT 1
#endif
#ifndef C3_SPLIT_TILES_DEFAULT
#define C3_SPLIT_TILES_DEFAULT 128
#endif
#ifndef C3_DMA
#define C3_DMA 1        /* 0: the MF16 form keeps the weight stream in registers (global -> VGPR -> ds_write), two ring slots (A/B) */
#endif
#ifndef C3_ABL
#define C3_ABL 0        /* ablation builds (timing only, wrong results): 1 no weight stream, 2 no halo staging, 4 no barrier, 8 no fragment reads, 16 no epilogue stores */
#endif
#ifndef C3_ABL4
#define C3_ABL4 0       /* ablation builds of MODE 4 (timing only, wrong results): 1 no dL/dy store, 2 no second-stream loads */
#endif

#define C3_HPX 204                          /* halo pixels of a 4 x 32 patch (6 x 34); an 8 x 16 patch needs 10 x 18 = 180 */
#define C3_AP 48                            /* bytes per halo pixel and plane: 16 fp16 + 16 (conflict-free ds_read_b128) */
#define C3_APL (C3_HPX * C3_AP)
#define C3_ABUF (2 * C3_APL)
#define C3_BP 80                            /* bytes per weight row and plane of a slot: 2 x 16 fp16 + 16 */
// LDS geometry (device and host).  MF16 — Cout 128 on 4 x 32 patches, the hourglass's 3x3 — runs on v_mfma_f32_16x16x32_f16: an MFMA
// then spans BOTH K-steps of a pair, a lane's 8 k-values are one 16-byte half of a pixel's (weight row's) 16-channel K-step, and the
// layouts keep the halves in planes of their own, 16 bytes per pixel / row, dense: the 16 lanes of a k-group read 16 consecutive
// 16-byte slots and the four k-groups sit a multiple of 256 bytes apart -> every ds_read_b128 is conflict-free without padding
//   halo chunk buffer  [plane hi|lo][half][208 px][16 B]      (13 KB against 19 KB)
//   ring slot          [plane hi|lo][k-group 0..3][CO][16 B]  (16 KB against 20 KB; k-group = 2 (K-step of the pair) + half)
template <int CO, int PW, bool MF> struct C3Geo {
    static constexpr bool MF16 = MF;
    static_assert(!MF || (CO == 128 && PW == 32), "MF16 shape");
    static constexpr int SUBPL = 208 * 16;
    static constexpr int APL = MF16 ? 2 * SUBPL : C3_APL, ABUF = 2 * APL;
    static constexpr int BPL = MF16 ? 4 * CO * 16 : CO * C3_BP, BSLOT = 2 * BPL;
    // DMA (MF16, one-stream modes): the weight pairs go global -> LDS directly (buffer_load_dwordx4 ... lds: a wave-instruction writes
    // 64 lanes x 16 B = one k-group's 64 rows, contiguous in the layout above), three pairs ahead into a THREE-slot ring — no VGPR round
    // trip (16 registers), no ds_write; the statistics scratch then has 2 KB of its own (no ring slot is ever free)
    static constexpr bool dma(const int mode) { return MF16 && C3_DMA && mode != 4; }
    static constexpr int nslots(const int mode) { return dma(mode) ? 3 : 2; }
    static constexpr int red_bytes(const int mode) { return dma(mode) ? 2048 : 0; }
    // halo buffers, weight ring, (statistics scratch,) BatchNorm vectors
    static constexpr int lds(const int mode) { return 2 * ABUF + nslots(mode) * BSLOT + red_bytes(mode) + (mode == 4 ? 1536 : 1024); }
};

// MODE: 0 no residual, 1 res1, 3 the BatchNorm-backward epilogue of a data-gradient launch (res1 = the BatchNorm input x),
// 4 = 3 with the BatchNorm backward of the layer BEHIND folded into the operand load (round 5): the A operand is not dL/dy but
//     dL/dz (ReLU-masked, reduced) and y, two streams, and every element is formed as  dy = scale (dz - c0 - (y - mean) invstd c1)
//     = P dz + R y + S  while it is staged (the separate dsnt_bn_act_bwd_apply pass in front of every 3x3 data gradient is gone);
//     the 128 pixels of the patch itself (not the halo ring: each pixel of the image is interior to exactly one patch) are also
//     written to p.ap_out — the materialised dL/dy the weight gradient reads afterwards.  The affine form carries the
//     conditioning of a scale / shift BatchNorm forward (eps |mean| / std), like `fmaf(x, scale, shift)` everywhere else.
// PW: the patch is 128 / PW rows of PW pixels — 4 x 32 (W % 32 == 0), or 8 x 16 for the 16-pixel-wide levels (a 32-pixel MFMA
// tile then spans two patch rows: a few two-way LDS bank conflicts on the activation fragments, immaterial at that size)
// SP (round 5): a 128-column convolution with too few patches to fill the chip (the 16 x 16 and 32 x 32 levels of the 8-stack net at batch
// 16: 32 / 128 patches on 512 slots) runs as TWO 64-column halves per patch — CO = 64 code, the weight stream and every column index
// offset by 64 x (blockIdx & 1): twice the workgroups, half the MFMAs per K-step and workgroup; both stage the same halo.  Every
// output is the same sum in the same order as in the unsplit 32x32x16 kernel (bit-identical).
template <int CO, bool PRO, int MODE, int PW, bool MF, bool SP>
__global__ __launch_bounds__(256, 2) void conv3s_kernel(ConvP p, int ntiles) {
    static_assert(!SP || (CO == 64 && !MF && MODE != 4), "column split: 64-column code, one-stream modes");
    constexpr int COG = SP ? 2 * CO : CO;          // columns of a weight row block in the stream
    const int choff = SP ? CO * (int)(blockIdx.x & 1) : 0;
    const int vstep = SP ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    constexpr int WN = CO / 64, WM = 4 / WN, TM = 4 / WM, TN = 2;
    constexpr int PH = 128 / PW, HW = PW + 2, HPX = (PH + 2) * HW, ITEMS = HPX * 4;
    static_assert(HPX <= C3_HPX, "halo buffer");
    typedef C3Geo<CO, PW, MF> G;
    constexpr bool MF16 = G::MF16;
    constexpr int SUBPL = G::SUBPL, APL = G::APL, ABUF = G::ABUF, BPL = G::BPL, BSLOT = G::BSLOT;
    constexpr bool APPLY = MODE == 4, BNB = MODE >= 3;
    static_assert(!(APPLY && PRO), "the folded BatchNorm backward replaces the prologue");
    constexpr int NJB = CO / 32;                // 16-byte weight units per thread and K-step pair
    constexpr int UPP = 4 * CO;                 // units per plane and pair
    const unsigned OOB = 0xF0000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char c3_smem[];
    unsigned char* As = c3_smem;                // [2 chunk buffers][2 planes][204 px][48]
    constexpr bool DMA = G::dma(MODE);
    unsigned char* Bs = c3_smem + 2 * ABUF;     // [2 slots][2 planes][CO][80]  (MF16: C3Geo; DMA: 3 slots)
    float* RED = reinterpret_cast<float*>(Bs + G::nslots(MODE) * BSLOT);      // DMA: the epilogue's statistics scratch [WM][CO][2]
    float* SS = reinterpret_cast<float*>(Bs + G::nslots(MODE) * BSLOT + G::red_bytes(MODE));   // PRO: [2][Cin] BN scale / shift x operand scale; APPLY: [3][Cin] P, R, S

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int nchunks = p.Cin >> 4;             // even
    const int npairs = nchunks * 9 / 2;
    const int tws = p.W / PW, ths = p.H / PH;
    const float sa = pow2_scale(bound64(p.a_bound)), sw = pow2_scale(bound64(p.w_bound));
    const float osc = 1.f / (sa * sw);

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.M * p.Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.wq), 0, (int)(((size_t)p.wq_stride + (size_t)p.Cout * p.K) * 2u), 0x00020000);
    const int ybytes = (int)((size_t)p.M * p.Cout * 4u);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res1 ? p.res1 : p.y), 0, ybytes, 0x00020000);
    const unsigned rowbytes = (unsigned)p.Cout * 4u;
    const int xbytes = (int)((size_t)p.M * p.Cin * 4u);
    const __amdgpu_buffer_rsrc_t x2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(APPLY ? p.ap_y : p.x), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dor = __builtin_amdgcn_make_buffer_rsrc(APPLY ? p.ap_out : p.y, 0, APPLY ? xbytes : ybytes, 0x00020000);

    // ---- halo staging: item = tid + 256 j (< 816) -> halo pixel item >> 2, 4-channel quad item & 3
    const int kc = tid & 3;
    // aok bits 0-3: the halo pixel (item j) lies inside the image; bits 4-7 (APPLY): it is one of the patch's own.  `aok` belongs to
    // the items being STORED, `aokn` to the tile whose offsets are in aoffs (they differ for a few iterations around a tile change)
    unsigned aoffs[4], aok = 0, aokn = 0;
    auto set_tile = [&](const int vv) {         // global offsets of the tile with virtual index vv; returns its first pixel
        aokn = 0;
        int img = 0, th = 0, tw = 0;
        const bool live = vv < ntiles;
        if (live) {
            int tile;
            xcd_remap(vv, ntiles, tile);
            tw = tile % tws;
            th = (tile / tws) % ths;
            img = tile / (tws * ths);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int px = (tid >> 2) + 64 * j;
            const int hy = px / HW, hx = px - hy * HW;
            const int ih = th * PH - 1 + hy, iw = tw * PW - 1 + hx;
            const bool in = live && px < HPX && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            aoffs[j] = in ? (unsigned)(((img * p.H + ih) * p.W + iw) * p.Cin + kc * 4) * 4u : OOB;
            aokn |= (in ? 1u : 0u) << j;
            if (APPLY) aokn |= ((in && hy >= 1 && hy <= PH && hx >= 1 && hx <= PW) ? 16u : 0u) << j;
        }
        return (img * p.H + th * PH) * p.W + tw * PW;
    };
    // Halo staging, ROLLING (round 5): a chunk is 4 items per thread (item j = halo pixel (tid >> 2) + 64 j, 4 channels); instead of
    // fetching a whole chunk at once and storing it two iterations later, ONE item is fetched and ONE stored per iteration, each
    // item LEAD iterations after its fetch, through a ring of NSLOT register sets (item j <-> set j % NSLOT).  One stream: LEAD 4,
    // four sets — the 16 registers of the chunk-at-a-time schedule; two streams (MODE 4): LEAD 2, two sets of two (with both streams
    // fetched a chunk at a time the kernel spilled: 256 VGPRs + 68 B of scratch).  What it buys is issue overlap, and only together
    // with the branch-free `store1` below: an item's ~50 VALU instructions (transform + split) are spread over the 12 MFMAs of its
    // step by the scheduling groups — with the divergent per-item guard of rounds 3-4 every item sat in a basic block of its own and
    // ran as a clump with the matrix pipe idle (then rolling was 1.5 % SLOWER than chunk-at-a-time; without the branch it is
    // 1 % faster at 64 x 64, 5 % at 32 x 32, and -0.10 ms per hg2 step: profiles/r05_ab_switches.txt box E).
    // -DC3_STAGE_ROLL=0: the chunk-at-a-time schedule of the one-stream modes (A/B).
#ifndef C3_STAGE_ROLL
#define C3_STAGE_ROLL 1
#endif
    constexpr bool ROLL = APPLY || MF16 || C3_STAGE_ROLL;
    constexpr int NSLOT = APPLY ? 2 : 4, LEAD = APPLY ? 2 : 4;
    c3_u32x4 ra[NSLOT], ra2[APPLY ? NSLOT : 1];
    const float lo_valid = p.in_relu ? 0.f : -__builtin_inff();
    if (APPLY) {
        for (int k = tid; k < p.Cin; k += 256) {
            const float sc = p.ap_scale[k], is = p.ap_invstd[k], mu = p.ap_mean[k], c0 = p.ap_coef[k], c1 = p.ap_coef[p.Cin + k];
            const float q = sc * is * c1;
            SS[k] = sc;                                             // dy = P dz + (S - Q y)
            SS[p.Cin + k] = -q;
            SS[2 * p.Cin + k] = (float)((double)q * (double)mu - (double)sc * (double)c0);
        }
        __syncthreads();
    }
    if (PRO) {
        for (int k = tid; k < p.Cin; k += 256) {
            SS[k] = p.in_scale[k] * sa;                             // the operand scale rides in the BN vectors
            SS[p.Cin + k] = p.in_shift[k] * sa;
        }
        __syncthreads();
    }
    auto load1 = [&](c3_u32x4& a, c3_u32x4& b, const int c, const int j) {
        a = __builtin_amdgcn_raw_buffer_load_b128(xr, aoffs[j], c * 64, 0);
        if (APPLY) b = (C3_ABL4 & 2) ? a : __builtin_amdgcn_raw_buffer_load_b128(x2r, aoffs[j], c * 64, 0);
    };
    // item j of chunk c: transform (BatchNorm + ReLU prologue / folded BatchNorm backward / operand scale), split, two 8-byte LDS
    // stores; okb = the aok bits of the tile the item belongs to
    auto store1 = [&](const c3_u32x4 a, const c3_u32x4 b, const int buf, const int c, const int j, const unsigned okb) {
        // NO branch on the thread index in here: a divergent `if (tid + 256 j < ITEMS)` costs nothing by itself, but it ends the
        // basic block, and the ~50 VALU instructions of an item then cannot be scheduled between the MFMAs of the step — they run
        // as one clump with the matrix pipe idle on both waves of the SIMD (round 5: the ISA of rounds 3-4 had every item behind
        // such a branch).  Whole items are decided at compile time; the threads beyond the end of the LAST, partial item run the
        // same instructions on zeros (their loads were out of range) and park the result in the pad bytes of a pixel row.
        if (256 * j >= ITEMS) return;                           // (compile time: j is a constant after unrolling)
        const bool active = (256 * j + 255 < ITEMS) || (tid + 256 * j < ITEMS);
        float4 v = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
        if (PRO) {
            const float4 sc = *reinterpret_cast<const float4*>(SS + c * 16 + kc * 4);
            const float4 sh = *reinterpret_cast<const float4*>(SS + p.Cin + c * 16 + kc * 4);
            // BN FMAs; ReLU + zero padding (applied after BN + ReLU) as one median per element
            const bool ok = (okb >> j) & 1u;
            const float lo = ok ? lo_valid : 0.f, hi = ok ? __builtin_inff() : 0.f;
            v.x = __builtin_amdgcn_fmed3f(fmaf(v.x, sc.x, sh.x), lo, hi);
            v.y = __builtin_amdgcn_fmed3f(fmaf(v.y, sc.y, sh.y), lo, hi);
            v.z = __builtin_amdgcn_fmed3f(fmaf(v.z, sc.z, sh.z), lo, hi);
            v.w = __builtin_amdgcn_fmed3f(fmaf(v.w, sc.w, sh.w), lo, hi);
        } else if (APPLY) {
            const float4 sc = *reinterpret_cast<const float4*>(SS + c * 16 + kc * 4);
            const float4 sh = *reinterpret_cast<const float4*>(SS + p.Cin + c * 16 + kc * 4);
            const float4 sq = *reinterpret_cast<const float4*>(SS + 2 * p.Cin + c * 16 + kc * 4);
            // dy = P dz + (R y + S); a pixel outside the image is zero padding of dy (both loads returned zeros: mask S)
            const bool ok = (okb >> j) & 1u;
            v.x = fmaf(v.x, sc.x, fmaf(__uint_as_float(b.x), sh.x, ok ? sq.x : 0.f));
            v.y = fmaf(v.y, sc.y, fmaf(__uint_as_float(b.y), sh.y, ok ? sq.y : 0.f));
            v.z = fmaf(v.z, sc.z, fmaf(__uint_as_float(b.z), sh.z, ok ? sq.z : 0.f));
            v.w = fmaf(v.w, sc.w, fmaf(__uint_as_float(b.w), sh.w, ok ? sq.w : 0.f));
            // (no branch: a pixel of the halo ring gets an out-of-range offset and the store is dropped.  The offsets of a tile
            // are stable from its first fetch to its last store: set_tile comes after the last store of the previous tile's items)
            if (!(C3_ABL4 & 1))
                __builtin_amdgcn_raw_buffer_store_b128((c3_u32x4){__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z),
                                                                  __float_as_uint(v.w)}, dor, ((okb >> (4 + j)) & 1u) ? aoffs[j] : OOB, c * 64, 0);
            v.x *= sa; v.y *= sa; v.z *= sa; v.w *= sa;
        } else {
            v.x *= sa; v.y *= sa; v.z *= sa; v.w *= sa;
        }
        uint2 q1, q2;
        split4h(v, q1, q2);
        // (the pad: bytes 32 .. 47 of a pixel's 48, never read)
        // (MF16: pixels 204 .. 207 of a half-plane)
        unsigned char* dst = As + buf * ABUF +
            (MF16 ? (kc >> 1) * SUBPL + (active ? (tid >> 2) + 64 * j : 204 + ((tid >> 2) & 3)) * 16 + (kc & 1) * 8
                  : (active ? ((tid >> 2) + 64 * j) * C3_AP + kc * 8 : (tid >> 2) * C3_AP + 32 + (kc & 1) * 8));
        *reinterpret_cast<uint2*>(dst) = q1;
        *reinterpret_cast<uint2*>(dst + APL) = q2;
    };
    // the ring: item j lives in register set j % NSLOT
    auto ld = [&](const int c, const int j) { load1(ra[j % NSLOT], ra2[APPLY ? j % NSLOT : 0], c, j); };
    auto st = [&](const int buf, const int c, const int j) { store1(ra[j % NSLOT], ra2[APPLY ? j % NSLOT : 0], buf, c, j, aok); };
    // the old schedule: a chunk at a time
    auto gloadA = [&](const int c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ld(c, j);
    };
    auto storeA = [&](const int buf, const int c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) st(buf, c, j);
    };

    // ---- weight stream: pair gp = K-steps 2 gp, 2 gp + 1 = 2 x [CO][16] fp16 per plane, contiguous
    unsigned bvoff[NJB], blds[NJB];
#pragma unroll
    for (int j = 0; j < NJB; ++j) {
        const int u = tid + 256 * j, pl = u / UPP, uu = u % UPP;
        bvoff[j] = (unsigned)((size_t)pl * p.wq_stride * 2u) +
                   (SP ? (unsigned)((uu / (2 * CO)) * (COG * 32) + choff * 32 + (uu % (2 * CO)) * 16) : (unsigned)uu * 16u);
        // (stream order of a pair and plane: [K-step][CO][2 halves of 16 bytes])
        blds[j] = MF16 ? (unsigned)(pl * BPL + (2 * (uu / (2 * CO)) + (uu & 1)) * (CO * 16) + ((uu >> 1) % CO) * 16)
                       : (unsigned)(pl * BPL + ((uu >> 1) % CO) * C3_BP + (uu / (2 * CO)) * 32 + (uu & 1) * 16);
    }
    c3_u32x4 rb[NJB];
    int gp = 0;                                 // the pair the NEXT gloadB fetches
    // (MF16 — CO 128, four units per thread: unit j = plane j >> 1, K-step j & 1 of the pair, weight row tid >> 1, half tid & 1:
    // ONE offset register each way, the rest compile-time / scalar)
    const unsigned wplane = (unsigned)((size_t)p.wq_stride * 2u);
    auto gloadB = [&]() {
        const unsigned so = (unsigned)gp * (unsigned)(COG * 64);
#pragma unroll
        for (int j = 0; j < NJB; ++j)
            rb[j] = MF16 ? __builtin_amdgcn_raw_buffer_load_b128(wr, bvoff[0] + (j & 1) * 4096, so + (j >> 1) * wplane, 0)
                         : __builtin_amdgcn_raw_buffer_load_b128(wr, bvoff[j], so, 0);
        gp = gp + 1 == npairs ? 0 : gp + 1;
    };
    auto storeB = [&](const unsigned slot) {
#pragma unroll
        for (int j = 0; j < NJB; ++j)
            *reinterpret_cast<c3_u32x4*>(Bs + (slot + blds[MF16 ? 0 : j]) + (MF16 ? (j >> 1) * BPL + (j & 1) * 4096 : 0)) = rb[j];
    };

    // DMA: wave w copies unit blocks 4 w .. 4 w + 3 of a pair (16 blocks of 64 rows x 16 B): block = plane (w >> 1), k-group 2 (w & 1) + (j >> 1),
    // row half j & 1.  LDS destination = M0 (wave-uniform) + lane x 16; global source = lane x 32 (VGPR) + a scalar offset (the stream
    // keeps a weight row's two halves side by side: [K-step][row][2 x 16 B]).  hipcc does not count these loads: the waits are
    // placed by hand (`dma_wait`), by the guide's rule — the wait before a barrier, the reads after it.
    const c3_i32x4 wrs = {(int)(unsigned)(size_t)p.wq, (int)((unsigned)((size_t)p.wq >> 32) & 0xffffu),
                          (int)(((size_t)p.wq_stride + (size_t)p.Cout * p.K) * 2u), 0x00020000};
    const unsigned dma_voff = (unsigned)lane * 32u;
    const unsigned dma_lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)Bs
                              + (unsigned)((wave >> 1) * BPL + (wave & 1) * 2 * (CO * 16));
    const unsigned dma_g0 = (unsigned)(wave >> 1) * wplane + (unsigned)(wave & 1) * 4096u;
    auto dmaB = [&](const unsigned slot) {
        // the four pieces: LDS + 1024 j, stream + {0, 2048, 16, 2064}; M0 is written in the statement that uses it and restored (the
        // compiler owns it); an SALU instruction between every write of M0 and the load that reads it
        const unsigned so = (unsigned)gp * (unsigned)(CO * 64) + dma_g0, la = dma_lds0 + slot;
        unsigned keep, t;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %2\n\t"
                     "s_add_u32 %1, %3, 0x800\n\t"
                     "buffer_load_dwordx4 %4, %5, %3 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %4, %5, %1 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\t"
                     "s_add_u32 %1, %3, 16\n\t"
                     "buffer_load_dwordx4 %4, %5, %1 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\t"
                     "s_add_u32 %1, %3, 0x810\n\t"
                     "buffer_load_dwordx4 %4, %5, %1 offen lds\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(t)
                     : "s"(la), "s"(so), "v"(dma_voff), "s"(wrs)
                     : "scc");
        gp = gp + 1 == npairs ? 0 : gp + 1;
    };

    // ---- fragments
    struct Frag { f16x8 a[TM][2], b[TN][2]; };
    unsigned aoff[TM], boff[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int mt = wm * TM + a;              // 32-pixel tile of the patch: one patch row (PW 32) or two (PW 16)
        const int prow = PW == 32 ? mt : 2 * mt + (lr >> 4), pcol = lr & (PW - 1);
        aoff[a] = (unsigned)((prow * HW + pcol) * C3_AP + 16 * lh);
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) boff[b] = (unsigned)(((wn * TN + b) * 32 + lr) * C3_BP + 16 * lh);
    // local K-step s of a chunk pair (0..17; 18 = step 0 of the next pair): chunk buffer (s / 9) & 1, tap s % 9
    auto rd = [&](Frag& F, const int s, const unsigned slot) {
        const int t = s % 9, buf = (s / 9) & 1;
        const int toff = buf * ABUF + ((t / 3) * HW + (t % 3)) * C3_AP;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
                F.a[a][pl] = *reinterpret_cast<const f16x8*>(As + aoff[a] + toff + pl * APL);
#pragma unroll
            for (int b = 0; b < TN; ++b)
                F.b[b][pl] = *reinterpret_cast<const f16x8*>(Bs + slot + boff[b] + pl * BPL + (s & 1) * 32);
        }
    };
    f32x16 acc[TM][TN];
    auto mm = [&](const Frag& F) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a[a][1], F.b[b][0], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a[a][0], F.b[b][1], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a[a][0], F.b[b][0], acc[a][b], 0, 0, 0);
            }
    };

    // ---- MF16: the same wave tile (2 patch rows x 64 channels) as 4 x 4 blocks of 16 pixels x 16 channels.  Fragment sets of
    // 2 blocks x 2 planes: X / W = the activation rows of block rows 0-1 / 2-3, Y / Z = the weight rows of block columns 0-1 / 2-3.
    // A lane's k-groups 0-1 belong to K-step 2 it (tap, chunk buffer of that step), 2-3 to step 2 it + 1: a per-lane select
    // between two constants.  Quarter order X Y, X Z | barrier | W Y, W Z: each set is read once per pair (16 reads, as before),
    // Z and W (this pair) in the first half, X and Y of the NEXT pair in the second — X is dead after the first half, Y after the
    // third quarter — so all reads of pair it fall between the barriers of it - 1 and it, the epoch the ring and the staging
    // schedule already keep that pair's slot and chunk buffers stable for
    typedef f16x8 FSet[2][2];
    FSet X, Y, Z, W;
    c3_f32x4 acc16[MF16 ? 4 : 1][MF16 ? 4 : 1];
    // (one address register per operand: block r of the wave tile is a compile-time offset away)
    const bool khi = (lane & 32) != 0;
    const unsigned a16base = (unsigned)(((wm * TM * HW + (lane & 15)) * 16) + ((lane >> 4) & 1) * SUBPL);
    const unsigned b16base = (unsigned)((lane >> 4) * (CO * 16) + (wn * 64 + (lane & 15)) * 16);
    auto a16off = [&](const int r) { return ((r >> 1) * HW + 16 * (r & 1)) * 16; };
    auto b16off = [&](const int r) { return r * 256; };
    auto ldA = [&](FSet& S, const int rbp, const int s0) {
        const int t0 = s0 % 9, t1 = (s0 + 1) % 9;
        const unsigned c0 = (unsigned)(((s0 / 9) & 1) * ABUF + ((t0 / 3) * HW + t0 % 3) * 16);
        const unsigned c1 = (unsigned)((((s0 + 1) / 9) & 1) * ABUF + ((t1 / 3) * HW + t1 % 3) * 16);
        const unsigned char* src = As + (a16base + (khi ? c1 : c0));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) S[i][pl] = *reinterpret_cast<const f16x8*>(src + (a16off(2 * rbp + i) + pl * APL));
    };
    auto ldB = [&](FSet& S, const int cbp, const unsigned slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) S[i][pl] = *reinterpret_cast<const f16x8*>(Bs + (slot + b16base) + (b16off(2 * cbp + i) + pl * BPL));
    };
    auto mmq = [&](const FSet& A, const FSet& B, const int rbp, const int cbp) {
        // (term-major: the three MFMAs of a block are four MFMAs apart — a dependent 4-pass MFMA issued back to back waits for the
        // result of the one before; each block still sums lo x hi, hi x lo, hi x hi in this order)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    c3_f32x4& c = acc16[MF16 ? 2 * rbp + i : 0][MF16 ? 2 * cbp + j : 0];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[i][t == 0], B[j][t == 1], c, 0, 0, 0);
                }
    };
    // result element e (0..15) of the 32-pixel x 32-channel block (a, b) of the wave tile, for the epilogue
    auto accv = [&](const int a, const int b, const int e) {
        if constexpr (MF16) return acc16[2 * a + (e >> 3)][2 * b + ((e >> 2) & 1)][e & 3];
        else return acc[a][b][e];
    };
    auto accz = [&](const int a, const int b, const int e) {
        if constexpr (MF16) acc16[2 * a + (e >> 3)][2 * b + ((e >> 2) & 1)][e & 3] = 0.f;
        else acc[a][b][e] = 0.f;
    };

    auto zero = [&]() {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) accz(a, b, e);
    };

    const float am2lo = p.tail.amax_relu ? 0.f : -__builtin_inff();
    float am = 0.f, am2 = 0.f;

    auto epilogue = [&](float* red, const int m0) {
        // ---- epilogue from the C layout: register e of a 32 x 32 tile = pixels (e&3) + 8 (e>>2) + 4 lh of patch row
        // wm TM + a, channel lr of column tile wn TN + b.  `bnxt` (the slot just consumed) is free: statistics scratch.
        // MF16: the 32 x 32 block is four 16 x 16 results (a lane: 16 channels lane & 15 of pixels 4 (lane >> 4) + r — a store
        // would write four 64-byte runs, and two instructions share every 128-byte line).  v_permlane16_swap of the two column
        // blocks' registers (odd 16-lane rows of one <-> even rows of the other) turns them into the 32 x 32 shape: element e =
        // pixel 16 (e>>3) + 4 ((e>>2)&1) + (e&3) + 8 lh, channel lr — two full lines per store, one channel column per lane.
        {
            constexpr int T = TM * TN, NR = 16, NC = 1;
            const int lc = lr, lq = lh;
            float rbuf[2][NR];
            auto tile_off = [&](const int i) {       // i = b * TM + a
                const int a = i % TM, b = i / TM;
                return (unsigned)((m0 + (wm * TM + a) * (PW == 32 ? 1 : 2) * p.W + (MF16 ? 8 : 4) * lq) * p.Cout + choff + (wn * TN + b) * 32 + lc) * 4u;
            };
            // register e -> pixel (e&3) + 8 (e>>2) + 4 lh of the 32-pixel tile; PW 16: pixels 16.. are the next patch row
            auto reg_off = [&](const int e) {
                if (MF16) return (unsigned)((e & 3) + 4 * ((e >> 2) & 1) + 16 * (e >> 3)) * rowbytes;
                return PW == 32 ? (unsigned)((e & 3) + 8 * (e >> 2)) * rowbytes
                                : (unsigned)((e >> 3) * p.W + (e & 3) + 8 * ((e >> 2) & 1)) * rowbytes;
            };
            auto loadres = [&](float (&r)[NR], const int i) {
                if (MODE == 0) return;
                const unsigned o0 = tile_off(i);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned so = reg_off(e);
                    r[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1r, o0, so, 0));
                }
            };
            struct Col { float cb, sc, sh, mu, is; };
            auto loadcol = [&](const int b, const int h) {
                const int n = choff + (wn * TN + b) * 32 + 16 * h + lc;
                Col c = {0.f, 0.f, 0.f, 0.f, 0.f};
                if (BNB) { c.sc = p.bnb_scale[n]; c.sh = p.bnb_shift[n]; c.mu = p.bnb_mean[n]; c.is = p.bnb_invstd[n]; }
                else {
                    if (p.bias) c.cb = p.bias[n];
                    if (p.tail.amax_bn) { c.sc = p.tail.amax_scale[n]; c.sh = p.tail.amax_shift[n]; }
                }
                return c;
            };
            Col cols[NC], colsn[NC];
#pragma unroll
            for (int h = 0; h < NC; ++h) { cols[h] = loadcol(0, h); colsn[h] = cols[h]; }
            loadres(rbuf[0], 0);
            float s1[NC], s2[NC];
#pragma unroll
            for (int h = 0; h < NC; ++h) { s1[h] = 0.f; s2[h] = 0.f; }
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const int a = i % TM, b = i / TM;
                if (i + 1 < T) {
                    loadres(rbuf[(i + 1) & 1], i + 1);
                    if ((i + 1) % TM == 0) {
#pragma unroll
                        for (int h = 0; h < NC; ++h) colsn[h] = loadcol((i + 1) / TM, h);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MF16) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        c3_f32x4& c0 = acc16[2 * a + (q >> 2)][2 * b], & c1 = acc16[2 * a + (q >> 2)][2 * b + 1];
                        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(c0[q & 3]), __float_as_uint(c1[q & 3]), false, false);
                        c0[q & 3] = __uint_as_float(sw[0]);
                        c1[q & 3] = __uint_as_float(sw[1]);
                    }
                }
                const float (&r)[NR] = rbuf[i & 1];
                const unsigned o0 = (C3_ABL & 16) ? OOB : tile_off(i);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned so = reg_off(e);
                    const int h = 0;
                    const Col& col = cols[h];
                    float val = accv(a, b, e) * osc;
                    if (BNB) {
                        // val = dL/d relu(bn(x)); r = x: mask by the ReLU, accumulate the BatchNorm-backward sums
                        const float xv = r[e];
                        if (p.bnb_relu && fmaf(xv, col.sc, col.sh) <= 0.f) val = 0.f;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), yr, o0, so, 0);
                        s1[h] += val;
                        s2[h] = fmaf(val, (xv - col.mu) * col.is, s2[h]);
                        am = fmaxf(am, fabsf(val));          // max |dz| (p.tail.amax: the bound of a folded BatchNorm backward)
                    } else {
                        val += col.cb;
                        if (MODE >= 1) val += r[e];
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), yr, o0, so, 0);
                        am = fmaxf(am, fabsf(val));
                        if (p.tail.amax_bn) am2 = fmaxf(am2, fabsf(fmaxf(fmaf(val, col.sc, col.sh), am2lo)));
                        s1[h] += val;
                        s2[h] = fmaf(val, val, s2[h]);
                    }
                    accz(a, b, e);
                }
                if (a == TM - 1) {
#pragma unroll
                    for (int h = 0; h < NC; ++h) {
                        if (p.stats) {
                            s1[h] += __shfl_xor(s1[h], 32, 64);
                            s2[h] += __shfl_xor(s2[h], 32, 64);
                            if (lq == 0) {
                                red[(wm * CO + (wn * TN + b) * 32 + 16 * h + lc) * 2 + 0] = s1[h];
                                red[(wm * CO + (wn * TN + b) * 32 + 16 * h + lc) * 2 + 1] = s2[h];
                            }
                        }
                        s1[h] = 0.f; s2[h] = 0.f;
                        cols[h] = colsn[h];
                    }
                }
            }
        }
    };
    // ---- prologue of the workgroup's FIRST tile
    int v = SP ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    int m0 = set_tile(v), m0n = 0;
    aok = aokn;
    if (DMA) { dmaB(0); dmaB(BSLOT); dmaB(2 * BSLOT); }     // pairs 0, 1, 2
    else gloadB();
    if (ROLL) {
        // chunk 0 whole and item 0 of chunk 1 through temporaries (one round trip); the next items of chunk 1 stay in flight in
        // the ring, as the loop's schedule expects them at it = 0
        c3_u32x4 t[5], t2[APPLY ? 5 : 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) load1(t[j], t2[APPLY ? j : 0], 0, j);
        load1(t[4], t2[APPLY ? 4 : 0], 1, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) store1(t[j], t2[APPLY ? j : 0], 0, 0, j, aok);
        store1(t[4], t2[APPLY ? 4 : 0], 1, 1, 0, aok);
        if (LEAD == 4) { ld(1, 1); ld(1, 2); ld(1, 3); }
        else { ld(1, 1); ld(1, 2); }
    } else {
        gloadA(0);
        storeA(0, 0);
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)");
    else {
        storeB(0);
        gloadB();                               // pair 1 travels
    }
    unsigned bcur = 0, bnxt = BSLOT, bnn = 2 * BSLOT;     // (bnn: the third slot of the DMA ring)
    __syncthreads();
    Frag F0, F1;
    if (MF16) { ldA(X, 0, 0); ldB(Y, 0, bcur); }
    else rd(F0, 0, bcur);
    zero();

    if (C3_ABL & 8) {
        if (MF16) { ldB(Z, 1, bcur); ldA(W, 1, 0); }
        else rd(F1, 1, bcur);
    }
    for (; v < ntiles; v += vstep) {
        for (int c2 = 0; c2 < nchunks; c2 += 2) {
#pragma unroll
            for (int it = 0; it < 9; ++it) {
                // first half: the MFMAs of step 2 it carry the fragment reads of step 2 it + 1, the copy of the pair after this
                // one into the other slot (its last readers passed the previous barrier) and the fetch of the one after that
                __builtin_amdgcn_sched_barrier(0);
                if (MF16) {
                    if (!(C3_ABL & 8)) { ldB(Z, 1, bcur); ldA(W, 1, 2 * it); }
                    if (!(C3_ABL & 1) && !DMA) { storeB(bnxt); gloadB(); }
                    mmq(X, Y, 0, 0);
                    mmq(X, Z, 0, 1);
                    // the eight fragment reads behind the first eight MFMAs (Z is due at the 13th), then a weight unit every third
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
#pragma unroll
                    for (int i = 0; i < NJB; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                } else {
                if (!(C3_ABL & 8)) rd(F1, 2 * it + 1, bcur);
                if (!(C3_ABL & 1)) { storeB(bnxt); gloadB(); }
                mm(F0);
#pragma unroll
                for (int i = 0; i < 4 * TM; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < NJB; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (DMA && !(C3_ABL & 1)) {
                    // the pair whose fragments are read behind this barrier was asked for two iterations ago: younger, and allowed
                    // to be in flight, are the halo fetch issued right behind it, the four pieces of the previous iteration's pair and
                    // that iteration's fetch (iteration 8 of a chunk pair fetches nothing).  The first two iterations of a tile
                    // have nothing to wait for (prologue / the wait before the epilogue retired everything) — and must not wait:
                    // the epilogue's stores are in the same queue
                    if (it >= 2 || c2 != 0) {
                        if (C3_ABL & 2) asm volatile("s_waitcnt vmcnt(4)");
                        else if (it <= 1) asm volatile("s_waitcnt vmcnt(5)");
                        else asm volatile("s_waitcnt vmcnt(6)");
                    }
                }
                if (!(C3_ABL & 4)) __syncthreads();                // the other slot / chunk buffer is written; this slot is read
                // second half: the MFMAs of step 2 it + 1 carry the fragment reads of step 2 it + 2 and the halo staging.  (The chunk-at-a-time
                // schedule of the one-stream modes: chunk c2 + 1 — buffer 1: read last at step 17 of the previous trip, next at step 9 — is
                // fetched whole at it = 0 and stored at it = 2; chunk c2 + 2, or chunk 0 of the next tile — buffer 0: read last at
                // step 8 — at it = 4 / it = 7.  Fetching WHOLE chunks four / five iterations ahead instead, the stores in the
                // first half: 105 -> 108 us.)
                if (MF16) { if (!(C3_ABL & 8)) ldA(X, 0, 2 * it + 2); }
                else if (!(C3_ABL & 8)) rd(F0, 2 * it + 2, bnxt);
                if (!(C3_ABL & 2)) {
                if (ROLL) {
                    // buffer 1 (the odd chunk c2 + 1: read from step 9, last read at step 17 of the previous pair) takes its items at
                    // it = 8 of the previous pair and it = 0, 1, 2; buffer 0 (the next even chunk, or chunk 0 of the next tile: last
                    // read at step 8) at it = 4 .. 7; the store comes first, the fetch of a later item re-uses its registers
                    const bool last = c2 + 2 >= nchunks;
                    const int cn0 = last ? 0 : c2 + 2, cn1 = last ? 1 : c2 + 3;
                    if (it <= 2) st(1, c2 + 1, it + 1);
                    else if (it >= 4 && it <= 7) st(0, cn0, it - 4);
                    else if (it == 8) st(1, cn1, 0);
                    if (it == 3 && last) aok = aokn;                  // from it = 4 on the stored items are the next tile's
                    // DMA: this pair's slot has been read for the last time (Z, before the barrier): the pair three ahead goes
                    // there.  Between the item's store and the next item's fetch: hipcc's own count in front of the store does not
                    // know the four pieces — `vmcnt(3)` for the three younger fetches then means "all but the youngest three
                    // operations", i.e. it also retires every fetch older than the last piece; with the fetch BEHIND the pieces that
                    // is the fetch of two iterations ago (a lead of two left of the four), with the fetch in front of them it
                    // would be the previous iteration's
                    if (DMA && !(C3_ABL & 1)) dmaB(bcur);
                    if (LEAD == 4) {
                        if (it == 0 && last) m0n = set_tile(v + vstep);
                        if (it <= 3) ld(cn0, it);
                        else if (it <= 7) ld(cn1, it - 4);
                    } else {
                        if (it == 0) ld(c2 + 1, 3);
                        if (it == 2 && last) m0n = set_tile(v + vstep);
                        if (it >= 2 && it <= 5) ld(cn0, it - 2);
                        else if (it >= 6) ld(cn1, it - 6);
                    }
                } else {
                if (it == 0) gloadA(c2 + 1);
                if (it == 2) storeA(1, c2 + 1);
                if (it == 4) {
                    if (c2 + 2 < nchunks) gloadA(c2 + 2);
                    else { m0n = set_tile(v + vstep); aok = aokn; gloadA(0); }      // the next tile's first chunk
                }
                if (it == 7) storeA(0, c2 + 2 < nchunks ? c2 + 2 : 0);
                }
                }
                if ((C3_ABL & 2) && DMA && !(C3_ABL & 1)) dmaB(bcur);
                if (MF16) {
                    mmq(W, Y, 1, 0);
                    if (!(C3_ABL & 8)) ldB(Y, 0, bnxt);
                    mmq(W, Z, 1, 1);
                    // 24 MFMAs in pairs: the reads of X (next pair) behind the first four pairs, of Y behind pairs 7-10 (Y is
                    // this pair's operand through the first 12 MFMAs), the item's VALU spread over all of them
                    // 24 MFMAs: the reads of X (next pair) behind MFMAs 1-4, of Y behind 13-16 — Y is this pair's operand through
                    // the 12th, and the next pair's first MFMA wants it 8 MFMAs after the last read —, the item's VALU spread over
                    // the first 20, its fetch and stores at the end
                    // (the item's BatchNorm vectors are LDS reads too — NSS of them: asked for right behind X, or the item's VALU
                    // waits for them in one clump)
                    constexpr int NSS = APPLY ? 3 : (PRO ? 2 : 0), VS = NSS ? 8 : 0, VPG = APPLY ? 6 : (PRO ? 4 : 1);
#pragma unroll
                    for (int i = 0; i < 24; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i < 4 || (i >= 12 && i < 16) || (it != 3 && i >= 4 && i < 4 + NSS)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        if (it != 3 && i >= VS && i < 22) __builtin_amdgcn_sched_group_barrier(0x002, VPG, 0);
                        if (it != 3 && i == 20 && APPLY) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
                        if (it != 3 && i == 22) __builtin_amdgcn_sched_group_barrier(0x020, APPLY ? 2 : 1, 0);
                        if (it != 3 && i == 23) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                    }
                } else {
                mm(F1);
                }
                if (MF16) {
                } else if (ROLL) {
                    // every MFMA of the step carries a fragment read (the first eight) and a few of the item's ~50 VALU instructions
                    // (transform + split: left in one clump they run with the matrix pipe idle on BOTH waves of the SIMD, which
                    // reach this point together); the item's fetch and its two LDS stores behind the last MFMAs
                    constexpr int VPG = APPLY ? 6 : (PRO ? 5 : 4);
#pragma unroll
                    for (int i = 0; i < 4 * TM; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        if (it != 3) __builtin_amdgcn_sched_group_barrier(0x002, VPG, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, APPLY ? 2 : 1, 0);
                    if (it != 3) {
                        __builtin_amdgcn_sched_group_barrier(0x002, VPG, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, VPG, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, VPG, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (APPLY) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
                    }
                } else {
#pragma unroll
                for (int i = 0; i < 4 * TM; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if (it == 0 || it == 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
                if (it == 2 || it == 7) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                    }
                }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (DMA) { const unsigned tsw = bcur; bcur = bnxt; bnxt = bnn; bnn = tsw; }
                else { const unsigned tsw = bcur; bcur = bnxt; bnxt = tsw; }
            }
        }

        // (DMA: every piece in flight is retired here — behind the epilogue its stores would stand in the queue before them)
        if (DMA) asm volatile("s_waitcnt vmcnt(0)");
        float* red = DMA ? RED : reinterpret_cast<float*>(Bs + bnxt);      // `bnxt` (the slot just consumed) is free: statistics scratch [WM][CO][2]
        epilogue(red, m0);
        if (p.stats) {
            int tile;
            xcd_remap(v, ntiles, tile);
            __syncthreads();
            for (int u = tid; u < CO * 2; u += 256) {
                const int which = u & 1, c = u >> 1;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) s += red[(w * CO + c) * 2 + which];
                tail_store(p.stats + ((size_t)tile * 2 + which) * p.Cout + choff + c, s);
            }
            __syncthreads();                    // the next pair is stored over `red`
        }
        m0 = m0n;
    }
    if (p.tail.amax) amax_commit(am, p.tail.amax);
    if (p.tail.amax_bn) amax_commit(am2, p.tail.amax_bn, 1);
    if (DMA) asm volatile("s_waitcnt vmcnt(0)");       // (nothing may still be writing this workgroup's LDS when it ends)
}

static int c3_enabled = -1;

bool dsnt_conv3s_geom_ok(const dsnt_conv_geom* g) {
    if (c3_enabled < 0) {
        c3_enabled = dsnt_kernel_off("conv3s") ? 0 : 1;
    }
    if (!c3_enabled || !g) return false;
    if (!(g->R == 3 && g->S == 3 && g->stride == 1 && g->pad == 1 && g->dil == 1 && g->Ho == g->H && g->Wo == g->W)) return false;
    if (!((g->H % 4 == 0 && g->W % 32 == 0) || (g->H % 8 == 0 && g->W % 16 == 0)) || g->Cin % 32 != 0 || g->Cin > 128) return false;
    if (g->Cout != 64 && g->Cout != 128) return false;
    const size_t M = (size_t)g->N * g->H * g->W;
    if (M * g->Cin * 4u >= (1ull << 31) || M * g->Cout * 4u >= (1ull << 31)) return false;
    return true;
}

bool dsnt_conv3s_ok(const ConvP& p) {
    dsnt_conv_geom g;
    memset(&g, 0, sizeof(g));
    g.N = p.N; g.H = p.H; g.W = p.W; g.Cin = p.Cin; g.Ho = p.Ho; g.Wo = p.Wo; g.Cout = p.Cout;
    g.R = p.R; g.S = p.S; g.stride = p.stride; g.pad = p.pad; g.dil = p.dil;
    if (!dsnt_conv3s_geom_ok(&g)) return false;
    if (!p.a_bound || !p.w_bound || !p.wq) return false;
    if (p.res2) return false;
    if (p.bnb_scale && p.in_scale) return false;
    if (p.ap_y && !(p.bnb_scale && p.ap_scale && p.ap_mean && p.ap_invstd && p.ap_coef && p.ap_out && !p.res2)) return false;
    return true;
}

// which launches of the Cout-128, 4 x 32-patch shape take the 16x16x32 form: bit 0 forward (MODE 0 / 1), 1 the data gradient with the
// BatchNorm-backward epilogue (MODE 3), 2 the same with the folded BatchNorm backward (MODE 4).  DSNT_X_C3_MF16=<mask>: A/B and the
// tests of the other modes.  Default: forward only — alone (batch 32, 64 x 64, one box): forward 97-99 us against 103-108, MODE 3
// 102 against 113, MODE 4 130 against 128; in the hg2 / hg8 steps none of the masks moves the step beyond the run-to-run spread
// (profiles/r05_ab_switches.txt box J), so the default is the set that is faster alone and never slower in the step.
static int c3_mf16_modes() {
    static int m = -1;
    if (m < 0) { const char* e = getenv("DSNT_X_C3_MF16"); m = e ? atoi(e) : C3_MF16_DEFAULT; }
    return m;
}

template <int CO, bool PRO, int MODE, int PW, bool MF, bool SP = false>
static void c3_launch_k(const ConvP& p, hipStream_t st, bool share) {
    const int lds = C3Geo<CO, PW, MF>::lds(MODE);
    DSNT_SET_MAX_LDS((conv3s_kernel<CO, PRO, MODE, PW, MF, SP>), lds);
    const int cus = dsnt_device_cus();
    const int ntiles = p.N * (p.H / (128 / PW)) * (p.W / PW);
    int grid = 2 * cus;                         // two workgroups per CU (LDS), persistent over the tiles
    // DSNT_CONV_SHARE_CHIP: a launch on a side lane.  Two of these workgroups take a CU's whole LDS for the life of the launch,
    // and the dependency chain's small kernels on the other streams then wait for a slot (a bn_finalize of 8 workgroups: 63 us
    // instead of 6).  3/2 workgroups per CU: -0.08 ms per hg2 step (256 / 384 / 448 workgroups measured alike; once the side
    // lane's 1x1 kernel yields half of the CUs — gemm1.hip — this one's share no longer matters: 128 .. 512 workgroups within 0.03 ms).
    if (share) grid = cus + cus / 2;
    if (grid > ntiles) grid = ntiles;
    if (SP) grid = 2 * ntiles;                  // (only asked for when that is at most a slot per workgroup)
    DSNT_LAUNCH((conv3s_kernel<CO, PRO, MODE, PW, MF, SP>), dim3(grid), dim3(256), lds, st, p, ntiles);
}

// column split (SP) up to this many patches: DSNT_X_C3_SPLIT_TILES, A/B
static int c3_split_tiles() {
    static int m = -1;
    if (m < 0) { const char* e = getenv("DSNT_X_C3_SPLIT_TILES"); m = e ? atoi(e) : C3_SPLIT_TILES_DEFAULT; }
    return m;
}

// the kernel form a launch takes — the ONE place that decides it (the launcher below and dsnt_conv3s_form, which lets a test ask
// what ran): bit 0 column split (SP: two 64-column halves per patch), bit 1 the 16x16x32 form (MF), bit 2 8 x 16 patches
static int c3_form(int cout, int mode, int N, int H, int W) {
    const bool w32 = W % 32 == 0 && H % 4 == 0;
    if (cout == 128 && mode != 4) {
        const int ntiles = N * (H / (w32 ? 4 : 8)) * (W / (w32 ? 32 : 16));
        if (ntiles <= c3_split_tiles()) return 1 | (w32 ? 0 : 4);
    }
    if (!w32) return 4;
    if (cout == 128 && (c3_mf16_modes() & (mode == 4 ? 4 : mode == 3 ? 2 : 1))) return 2;
    return 0;
}

template <int CO, bool PRO, int MODE>
static void c3_launch_m(const ConvP& p, hipStream_t st, bool share) {
    const int form = c3_form(CO, MODE, p.N, p.H, p.W);
    if constexpr (CO == 128 && MODE != 4) {
        if (form & 1) {
            if (form & 4) return c3_launch_k<64, PRO, MODE, 16, false, true>(p, st, share);
            return c3_launch_k<64, PRO, MODE, 32, false, true>(p, st, share);
        }
    }
    if constexpr (CO == 128) {
        if (form & 2) return c3_launch_k<CO, PRO, MODE, 32, true>(p, st, share);
    }
    if (form & 4) c3_launch_k<CO, PRO, MODE, 16, false>(p, st, share);
    else c3_launch_k<CO, PRO, MODE, 32, false>(p, st, share);
}

template <int CO>
static void c3_launch(const ConvP& p, bool pro, hipStream_t st, bool share) {
    if (p.bnb_scale && p.ap_y) c3_launch_m<CO, false, 4>(p, st, share);
    else if (p.bnb_scale) c3_launch_m<CO, false, 3>(p, st, share);
    else if (pro) {
        if (p.res1) c3_launch_m<CO, true, 1>(p, st, share);
        else c3_launch_m<CO, true, 0>(p, st, share);
    } else if (p.res1) c3_launch_m<CO, false, 1>(p, st, share);
    else c3_launch_m<CO, false, 0>(p, st, share);
}

void dsnt_conv3s_launch(const ConvP& p, bool pro, hipStream_t st, bool share) {
    if (p.Cout == 128) c3_launch<128>(p, pro, st, share);
    else c3_launch<64>(p, pro, st, share);
}

int dsnt_conv3s_form_of(const dsnt_conv_geom* g, int mode) {
    if (!dsnt_conv3s_geom_ok(g) || mode < 0 || mode > 4 || mode == 2) return -1;
    return c3_form(g->Cout, mode, g->N, g->H, g->W);
}
