// The stem of the hourglass (hourglass.py:157 `conv1`: 7x7 / stride 2 / pad 3 on the 3-channel image) in the form the engine runs
// it: a 4x4 / stride 1 / pad 1 convolution of the 16-channel space-to-depth image (csrc/elementwise.hip: dsnt_s2d_input), 64 output
// channels, fp16x3 split: forward (stem4_fwd_kernel) and weight gradient (stem4_wgrad_kernel, below); the stem needs no data gradient.
//
// Why a kernel of its own: K = 256 is sixteen taps of ONE 16-channel chunk and N = 64, i.e. 168 MB of traffic against 17 GFLOP —
// HBM-bound by a factor of five.  The tiled implicit-GEMM kernel re-loads, re-scales and re-splits every input pixel for each of
// the 16 taps through its loader waves (537 MB of staging for 34 MB of input) and sits at 1.3 TB/s (130 us at batch 32).  Here:
//   * a workgroup (4 waves) owns a 4 x 32 patch of output pixels; its 7 x 35 input halo is loaded ONCE (16-byte loads), scaled,
//     split into two fp16 planes and written pixel-major into LDS (48-byte pixel pitch: conflict-free ds_read_b128, as conv3s);
//   * all sixteen taps read their A fragments from that one image with shifted addresses — no barrier inside a tile;
//   * the weights are REGISTER-resident: a wave owns 32 of the 64 output channels for 2 x 32 pixels, so a tap's B operand is
//     one 16-byte fragment per plane: 16 taps x 2 planes x 4 VGPRs = 128 registers, loaded once per (persistent) workgroup;
//   * the halo image is double-buffered: the next tile's halo is fetched before this tile's MFMAs and stored after them;
//   * epilogue from the accumulator layout (a register = 128-byte runs of the 64-channel rows), bias, BatchNorm statistics
//     accumulated in registers over the workgroup's tiles: ONE statistics row per workgroup (dsnt_stem4_fwd_stats_rows).
// Same K order (tap, channel) and the same three products per fragment pair as the tiled fp16x3 kernel; the fp32 summation order
// over K differs from it in nothing (one accumulator chain per output), so results agree with dsnt_conv_fwd_f16x3_ex bit for bit
// except where that kernel's own K-step order differs — the tests hold both to the fp32 bar.
#include "conv_split.h"
#include "stem4.h"
#include <stdlib.h>

typedef unsigned s4_u32x4 __attribute__((ext_vector_type(4)));

#define S4_HW 35                            /* halo width: 32 + 3 */
#define S4_HPX (7 * S4_HW)                  /* 245 halo pixels */
#define S4_AP 48                            /* bytes per halo pixel and plane: 16 fp16 + 16 */
#define S4_APL (S4_HPX * S4_AP)
#define S4_ABUF (2 * S4_APL)

struct Stem4P {
    const float* x; const unsigned short* wq; long wq_stride; const float* bias; float* y; float* stats;
    const float* a_bound; const float* w_bound;
    int N, H, W, Ho, Wo, M;
    OutBoundsP tail;
};

__global__ __launch_bounds__(256, 2) void stem4_fwd_kernel(Stem4P p, int ntiles) {
    const unsigned OOB = 0xF0000000u;
    __shared__ __attribute__((aligned(16))) unsigned char As[2 * S4_ABUF];       // [2 buffers][2 planes][245 px][48]
    __shared__ float red[2][2][64];                                             // statistics of the two pixel halves
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;    // pixel half (patch rows 2 wm, 2 wm + 1), channel half (32 wn ..)
    const int tws = p.Wo / 32, ths = p.Ho / 4;
    const float sa = pow2_scale(bound64(p.a_bound)), sw = pow2_scale(bound64(p.w_bound));
    const float osc = 1.f / (sa * sw);

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * 16u * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.M * 64u * 4u), 0x00020000);

    // ---- weights: this wave's 32 output channels, every tap, both planes: B fragment = 8 consecutive k of channel 32 wn + lr
    f16x8 wb[16][2];
    {
        const unsigned short* w0 = p.wq + (size_t)(32 * wn + lr) * 256 + 8 * lh;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            wb[t][0] = *reinterpret_cast<const f16x8*>(w0 + 16 * t);
            wb[t][1] = *reinterpret_cast<const f16x8*>(w0 + p.wq_stride + 16 * t);
        }
    }
    const float bias = p.bias ? p.bias[32 * wn + lr] : 0.f;

    // ---- halo staging: item = tid + 256 j (< 980) -> halo pixel item >> 2, 4-channel quad item & 3
    const int kc = tid & 3;
    unsigned aoffs[4];
    auto set_tile = [&](const int vv) {          // global offsets of the tile with virtual index vv; returns its first output pixel
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
            const int hy = px / S4_HW, hx = px - hy * S4_HW;
            const int ih = th * 4 - 1 + hy, iw = tw * 32 - 1 + hx;
            const bool in = live && px < S4_HPX && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            aoffs[j] = in ? (unsigned)(((img * p.H + ih) * p.W + iw) * 16 + kc * 4) * 4u : OOB;
        }
        return (img * p.Ho + th * 4) * p.Wo + tw * 32;
    };
    s4_u32x4 ra[4];
    auto gloadA = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(xr, aoffs[j], 0, 0);
    };
    auto storeA = [&](const int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (tid + 256 * j >= S4_HPX * 4) continue;
            float4 v = make_float4(__uint_as_float(ra[j].x) * sa, __uint_as_float(ra[j].y) * sa,
                                   __uint_as_float(ra[j].z) * sa, __uint_as_float(ra[j].w) * sa);
            uint2 q1, q2;
            split4h(v, q1, q2);
            unsigned char* dst = As + buf * S4_ABUF + ((tid >> 2) + 64 * j) * S4_AP + kc * 8;
            *reinterpret_cast<uint2*>(dst) = q1;
            *reinterpret_cast<uint2*>(dst + S4_APL) = q2;
        }
    };
    // fragment addresses: 32-pixel tile a of this wave = patch row 2 wm + a, pixel lr; tap (ty, tx) adds (ty * 35 + tx) pixels
    unsigned aoff[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) aoff[a] = (unsigned)(((2 * wm + a) * S4_HW + lr) * S4_AP + 16 * lh);

    float s1 = 0.f, s2 = 0.f;                   // statistics of channel 32 wn + lr over this lane's pixels, all tiles
    float am = 0.f;
    const unsigned rowbytes = 64u * 4u;

    int v = blockIdx.x;
    int m0 = set_tile(v);
    gloadA();
    storeA(0);
    __syncthreads();
    int buf = 0;
    for (; v < ntiles; v += gridDim.x) {
        const int m0n = set_tile(v + (int)gridDim.x);           // the next tile's halo travels during this tile's MFMAs
        gloadA();
        f32x16 acc[2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
        const unsigned char* A0 = As + buf * S4_ABUF;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const unsigned toff = (unsigned)(((t >> 2) * S4_HW + (t & 3)) * S4_AP);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(A0 + aoff[a] + toff);
                const f16x8 a2 = *reinterpret_cast<const f16x8*>(A0 + aoff[a] + toff + S4_APL);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, wb[t][0], acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, wb[t][1], acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, wb[t][0], acc[a], 0, 0, 0);
            }
        }
        storeA(buf ^ 1);                        // (nobody reads that buffer before the barrier below)
        // ---- epilogue from the C layout: register e of a 32 x 32 tile = pixel (e & 3) + 8 (e >> 2) + 4 lh of the patch row,
        // channel 32 wn + lr: a store instruction writes two 128-byte runs
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const unsigned o0 = (unsigned)((m0 + (2 * wm + a) * p.Wo + 4 * lh) * 64 + 32 * wn + lr) * 4u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float val = acc[a][e] * osc + bias;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), yr, o0, (unsigned)((e & 3) + 8 * (e >> 2)) * rowbytes, 0);
                s1 += val;
                s2 = fmaf(val, val, s2);
                am = fmaxf(am, fabsf(val));
            }
        }
        __syncthreads();
        buf ^= 1;
        m0 = m0n;
    }
    // ---- one statistics row per workgroup: [2][64] = (sum, sum of squares) of every channel over the workgroup's pixels
    if (p.stats) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (lh == 0) { red[wm][0][32 * wn + lr] = s1; red[wm][1][32 * wn + lr] = s2; }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63;
            p.stats[((size_t)blockIdx.x * 2 + which) * 64 + c] = red[0][which][c] + red[1][which][c];
        }
    }
    if (p.tail.amax) amax_commit(am, p.tail.amax);
}

// ---------------------------------------------------------------- host side
static int s4_enabled = -1;

static bool s4_geom_ok(const dsnt_conv_geom* g) {
    if (s4_enabled < 0) s4_enabled = dsnt_kernel_off("stem4") ? 0 : 1;
    if (!s4_enabled || !g) return false;
    if (!(g->R == 4 && g->S == 4 && g->stride == 1 && g->pad == 1 && g->dil == 1 && g->Cin == 16 && g->Cout == 64)) return false;
    if (!(g->Ho == g->H - 1 && g->Wo == g->W - 1 && g->Ho % 4 == 0 && g->Wo % 32 == 0)) return false;
    const size_t M = (size_t)g->N * g->Ho * g->Wo;
    if (M * 64u * 4u >= (1ull << 31) || (size_t)g->N * g->H * g->W * 16u * 4u >= (1ull << 31)) return false;
    return true;
}

static int s4_grid(const dsnt_conv_geom* g) {
    const int cus = dsnt_device_cus();
    const int ntiles = g->N * (g->Ho / 4) * (g->Wo / 32);
    const int grid = 2 * cus;                   // two workgroups per CU, persistent over the tiles
    return grid < ntiles ? grid : ntiles;
}

extern "C" int dsnt_stem4_fwd_ok(const dsnt_conv_geom* g) { return s4_geom_ok(g) ? 1 : 0; }
// rows of the statistics partial the launch writes (one per workgroup): hand this to dsnt_bn_finalize as ntiles
extern "C" int dsnt_stem4_fwd_stats_rows(const dsnt_conv_geom* g) { return s4_geom_ok(g) ? s4_grid(g) : 0; }

extern "C" int dsnt_stem4_fwd_f16x3(const float* x, const void* w_planes, int64_t plane_stride, const float* w_bound,
                                    const float* a_bound, const float* bias, float* y, float* stats_partial,
                                    const dsnt_conv_geom* g, const dsnt_out_bounds* tail, void* stream) {
    DSNT_REQUIRE(x && w_planes && w_bound && a_bound && y && g, DSNT_ERR_ARG, "dsnt_stem4_fwd_f16x3: bad argument");
    DSNT_REQUIRE(s4_geom_ok(g), DSNT_ERR_SHAPE, "dsnt_stem4_fwd_f16x3: geometry not supported (dsnt_stem4_fwd_ok: 4x4 / stride 1 / pad 1, "
                 "16 -> 64 channels, Ho %% 4 == 0, Wo %% 32 == 0)");
    DSNT_REQUIRE(dsnt_aligned16(x) && dsnt_aligned16(w_planes) && dsnt_aligned16(y) && plane_stride % 8 == 0 && plane_stride >= 64 * 256,
                 DSNT_ERR_ALIGN, "dsnt_stem4_fwd_f16x3: 16-byte alignment / plane stride");
    Stem4P p;
    memset(&p, 0, sizeof(p));
    if (int e = out_bounds_fill(p.tail, tail, "dsnt_stem4_fwd_f16x3")) return e;
    DSNT_REQUIRE(!p.tail.amax_bn, DSNT_ERR_ARG, "dsnt_stem4_fwd_f16x3: dsnt_out_bounds.amax_bn is not supported by this launch");
    p.x = x; p.wq = (const unsigned short*)w_planes; p.wq_stride = plane_stride; p.bias = bias; p.y = y; p.stats = stats_partial;
    p.a_bound = a_bound; p.w_bound = w_bound;
    p.N = g->N; p.H = g->H; p.W = g->W; p.Ho = g->Ho; p.Wo = g->Wo; p.M = g->N * g->Ho * g->Wo;
    const int ntiles = g->N * (g->Ho / 4) * (g->Wo / 32);
    DSNT_LAUNCH(stem4_fwd_kernel, dim3(s4_grid(g)), dim3(256), 0, (hipStream_t)stream, p, ntiles);
    DSNT_CHECK_LAUNCH("dsnt_stem4_fwd_f16x3");
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same convolution: dW[n][(t, c)] = sum over output pixels of dY[px][n] * xs[px + tap t][c]
// (64 x 256), the LAST kernel of a backward pass — nothing runs beside it.  The implicit-GEMM kernel gathers the 16 taps of every
// pixel through its loaders and sits at 0.9 TB/s (186 us at batch 32).  Here the contraction runs over pixels, so BOTH operands are
// read TRANSPOSED (ds_read_b64_tr_b16) from pixel-major fp16 images in LDS, each element staged once per patch:
//   * the 7 x 35 halo of the input (as the forward kernel) and the 128 x 64 dY patch, scaled, split, pixel-major;
//   * v_mfma_f32_16x16x32_f16 over 32-pixel row segments: A = dY^T (16 output channels x 32 pixels), B = the halo shifted by the
//     tap (32 pixels x 16 input channels): a 16 x 16 tile of dW per (channel group, tap);
//   * eight waves: wave w owns taps 2 w, 2 w + 1 for all 64 output channels (8 tiles = 32 accumulator registers), persistent over
//     the workgroup's patches; ONE slab [64][256] and one bias partial per workgroup (256 slabs: one workgroup per CU).
typedef short s4_s16x4 __attribute__((ext_vector_type(4)));
typedef short s4_s16x8 __attribute__((ext_vector_type(8)));
typedef float s4_f32x4 __attribute__((ext_vector_type(4)));
#define S4_LDS __attribute__((address_space(3)))
#define S4_GP 160                           /* bytes per pixel and plane of the dY image: 64 fp16 + 32 */
#define S4_GPL (128 * S4_GP)

__device__ __forceinline__ f16x8 s4_tr_frag(S4_LDS unsigned char* a0, S4_LDS unsigned char* a1) {
    const s4_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((S4_LDS s4_s16x4*)a0);
    const s4_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((S4_LDS s4_s16x4*)a1);
    const s4_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f16x8, v);
}

struct Stem4WP {
    const float* x; const float* dy; float* ws; const float* a_bound; const float* g_bound;
    int N, H, W, Ho, Wo, M, nwg;
};

__global__ __launch_bounds__(512, 1) void stem4_wgrad_kernel(Stem4WP p, int ntiles) {
    const unsigned OOB = 0xF0000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char s4_smem[];
    unsigned char* Xs = s4_smem;                    // [2 planes][245 px][48]
    unsigned char* Gs = s4_smem + S4_ABUF;          // [2 planes][128 px][160]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, lg = lane >> 4;
    const int tws = p.Wo / 32, ths = p.Ho / 4;
    const float sa = pow2_scale(bound64(p.a_bound)), sg = pow2_scale(bound64(p.g_bound));
    const float osc = 1.f / (sa * sg);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * 16u * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy), 0, (int)((size_t)p.M * 64u * 4u), 0x00020000);

    // ---- staging roles.  Halo: item = tid + 512 j (< 980): pixel item >> 2, quad item & 3.  dY: item = tid + 512 j (< 2048):
    // pixel item >> 4 of the 128-pixel patch (patch row item >> 9 ... ), 4-channel group item & 15
    const int kc = tid & 3;
    unsigned aoffs[2], goffs[4];
    auto set_tile = [&](const int vv) {
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
        for (int j = 0; j < 2; ++j) {
            const int px = (tid >> 2) + 128 * j;
            const int hy = px / S4_HW, hx = px - hy * S4_HW;
            const int ih = th * 4 - 1 + hy, iw = tw * 32 - 1 + hx;
            const bool in = live && px < S4_HPX && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            aoffs[j] = in ? (unsigned)(((img * p.H + ih) * p.W + iw) * 16 + kc * 4) * 4u : OOB;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int item = tid + 512 * j, px = item >> 4, q = item & 15;
            const int m = (img * p.Ho + th * 4 + (px >> 5)) * p.Wo + tw * 32 + (px & 31);
            goffs[j] = live ? (unsigned)(m * 64 + q * 4) * 4u : OOB;
        }
    };
    s4_u32x4 ra[2], rg[4];
    auto gload = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(xr, aoffs[j], 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) rg[j] = __builtin_amdgcn_raw_buffer_load_b128(gr, goffs[j], 0, 0);
    };
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);     // bias partial: channels 4 (tid & 15) .. + 3 over this thread's pixels
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (tid + 512 * j >= S4_HPX * 4) continue;
            const float4 v = make_float4(__uint_as_float(ra[j].x) * sa, __uint_as_float(ra[j].y) * sa,
                                         __uint_as_float(ra[j].z) * sa, __uint_as_float(ra[j].w) * sa);
            uint2 q1, q2;
            split4h(v, q1, q2);
            unsigned char* dst = Xs + ((tid >> 2) + 128 * j) * S4_AP + kc * 8;
            *reinterpret_cast<uint2*>(dst) = q1;
            *reinterpret_cast<uint2*>(dst + S4_APL) = q2;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int item = tid + 512 * j, px = item >> 4, q = item & 15;
            const float4 d = make_float4(__uint_as_float(rg[j].x), __uint_as_float(rg[j].y), __uint_as_float(rg[j].z), __uint_as_float(rg[j].w));
            bs.x += d.x; bs.y += d.y; bs.z += d.z; bs.w += d.w;
            uint2 q1, q2;
            split4h(make_float4(d.x * sg, d.y * sg, d.z * sg, d.w * sg), q1, q2);
            unsigned char* dst = Gs + px * S4_GP + q * 8;
            *reinterpret_cast<uint2*>(dst) = q1;
            *reinterpret_cast<uint2*>(dst + S4_GPL) = q2;
        }
    };

    // transposed fragment of a [pixel][channel] image: 16 pixels x 16 channels per instruction; lane -> pixel row 4 lg + ((lane >> 2) & 3),
    // 8 bytes at 8 (lane & 3) of the 32-byte channel group (csrc/bwd1.hip)
    const unsigned t_row = (unsigned)(4 * lg + ((lane >> 2) & 3)), t_col = (unsigned)(8 * (lane & 3));
    S4_LDS unsigned char* xs0 = (S4_LDS unsigned char*)Xs;
    S4_LDS unsigned char* gs0 = (S4_LDS unsigned char*)Gs;

    s4_f32x4 acc[2][4];                         // [tap 2 wave + i][channel group]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) acc[i][gq] = (s4_f32x4){0.f, 0.f, 0.f, 0.f};

    set_tile(blockIdx.x);
    gload();
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        lstore();
        __syncthreads();
        set_tile(v + (int)gridDim.x);            // the next patch travels during this one's MFMAs
        gload();
#pragma unroll
        for (int seg = 0; seg < 4; ++seg) {
            f16x8 a1[4], a2[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                S4_LDS unsigned char* gb = gs0 + (seg * 32 + t_row) * S4_GP + 32 * gq + t_col;
                a1[gq] = s4_tr_frag(gb, gb + 16 * S4_GP);
                a2[gq] = s4_tr_frag(gb + S4_GPL, gb + S4_GPL + 16 * S4_GP);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int t = 2 * wave + i;
                S4_LDS unsigned char* xb = xs0 + ((seg + (t >> 2)) * S4_HW + (t & 3) + t_row) * S4_AP + t_col;
                const f16x8 b1 = s4_tr_frag(xb, xb + 16 * S4_AP);
                const f16x8 b2 = s4_tr_frag(xb + S4_APL, xb + S4_APL + 16 * S4_AP);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    acc[i][gq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[gq], b1, acc[i][gq], 0, 0, 0);
                    acc[i][gq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[gq], b2, acc[i][gq], 0, 0, 0);
                    acc[i][gq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[gq], b1, acc[i][gq], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // ---- slab of this workgroup: ws[wg][n][k], D row = n = 16 gq + 4 lg + r, D column = k = 16 t + lc
    float* slab = p.ws + (size_t)blockIdx.x * 64 * 256;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                slab[(size_t)(16 * gq + 4 * lg + r) * 256 + 16 * (2 * wave + i) + lc] = acc[i][gq][r] * osc;
    // bias partial: the 32 pixel-threads of a channel group add up through LDS in a fixed order (the loop ended on a barrier)
    {
        float4* red = reinterpret_cast<float4*>(s4_smem);
        red[tid] = bs;
        __syncthreads();
        if (tid < 16) {
            float4 t = red[tid];
            for (int j = 1; j < 32; ++j) {
                const float4 u = red[16 * j + tid];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            *reinterpret_cast<float4*>(p.ws + (size_t)p.nwg * 64 * 256 + (size_t)blockIdx.x * 64 + 4 * tid) = t;
        }
    }
}

static int s4w_grid(const dsnt_conv_geom* g) {
    const int cus = dsnt_device_cus();
    const int ntiles = g->N * (g->Ho / 4) * (g->Wo / 32);
    return cus < ntiles ? cus : ntiles;          // one eight-wave workgroup per CU
}

int dsnt_stem4_wgrad_slabs(const dsnt_conv_geom* g) {
    static int on = -1;
    if (on < 0) on = dsnt_kernel_off("stem4w") ? 0 : 1;
    return (on && s4_geom_ok(g)) ? s4w_grid(g) : 0;
}

void dsnt_stem4_wgrad_launch(const float* x, const float* dy, float* ws, const float* a_bound, const float* g_bound,
                             const dsnt_conv_geom* g, hipStream_t st) {
    Stem4WP p;
    p.x = x; p.dy = dy; p.ws = ws; p.a_bound = a_bound; p.g_bound = g_bound;
    p.N = g->N; p.H = g->H; p.W = g->W; p.Ho = g->Ho; p.Wo = g->Wo; p.M = g->N * g->Ho * g->Wo;
    p.nwg = s4w_grid(g);
    const int ntiles = g->N * (g->Ho / 4) * (g->Wo / 32);
    const int lds = S4_ABUF + 2 * S4_GPL;
    DSNT_SET_MAX_LDS((stem4_wgrad_kernel), lds);
    DSNT_LAUNCH(stem4_wgrad_kernel, dim3(p.nwg), dim3(512), lds, st, p, ntiles);
}
