// Data gradient of a STRIDED convolution (the ResNet stage transitions: 3x3 / 2, 1x1 / 2, and the 7x7 / 2 stem when
// the image gradient is asked for) without a zero-stuffed intermediate.
//
//   dX[n, ih, iw, ci] = sum over (r, s, co) with (ih + pad - r dil) % stride == 0 and (iw + pad - s dil) % stride == 0 of
//                       dY[n, (ih + pad - r dil) / stride, (iw + pad - s dil) / stride, co] * W[co, r, s, ci]
//
// The pixels of dX fall into stride^2 PHASES (ih % stride, iw % stride); inside a phase the set of contributing taps is
// the same for every pixel and dY is read at unit stride: a phase is an ordinary small convolution (3x3 / 2: 1, 2, 2 and
// 4 taps; 1x1 / 2: one tap in phase (0, 0) and none elsewhere) whose rows are written to every stride-th pixel of dX.
// Against dsnt_zero_insert + the stride-1 kernel (round 1..2) that is 1 / stride^2 of the multiply-adds, no
// stride^2-times-dY scratch tensor and one launch instead of two.
//
// Kernel: the K-split form of conv.hip (conv_ksplit_kernel) — these launches have few rows and long reductions.  A
// 512-thread workgroup owns one 32 x 32 tile (32 pixels of ONE phase x 32 channels of dX); its eight waves split the
// phase's reduction (taps x Cout, 16 channels at a time), every lane streaming its operands straight from memory into
// v_mfma_f32_32x32x2_f32 (exact fp32: the results match the stuffed path to accumulation order), the eight partial
// tiles are summed through LDS in wave order (deterministic), and the epilogue offers what the engine's data-gradient
// launches use: accumulate into dX (res1), the BatchNorm-backward mask + per-tile sums (Bottleneck: the strided 3x3
// reads relu(bn(x))), max|written| as the next operand bound.  blockIdx.y = phase.
//
// Replaces cuDNN's strided backward-data of /root/reference/src/dsnt/model.py:103-121 (torchvision ResNet conv1,
// layerN[0].conv1 / conv2 and downsample[0]).
#include "common.h"
#include "bn_pro.h"
#include "conv_split.h"

#define UP_MAX_STRIDE 4

struct UpP {
    const float* dy; const float* wd; float* dx; const float* res1; float* stats;
    const float* bnb_scale; const float* bnb_shift; const float* bnb_mean; const float* bnb_invstd;
    unsigned* amax;
    int bnb_relu;
    int N, Hy, Wy, Cy;         // dY (= the forward convolution's output)
    int Hx, Wx, Cx;            // dX
    int R, S, s, K;            // K = R * S * Cy: one row of wd ([Cx][R][S][Cy], taps flipped: dsnt_conv_pack_dgrad)
    // per axis (0: rows, 1: columns) and phase: first contributing (flipped) tap, their count, dY offset of the first;
    // consecutive contributing taps are `per` apart and move the dY coordinate by `dstep`
    int r0[2][UP_MAX_STRIDE], cnt[2][UP_MAX_STRIDE], d0[2][UP_MAX_STRIDE];
    int per[2], dstep[2];
    int mtiles;                // 32-row tiles per phase (of the largest phase)
};

__global__ __launch_bounds__(512) void conv_dgrad_up_kernel(UpP p) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float part[8][32][33];
    const int nt32 = (p.Cx + 31) >> 5;
    const int ntile = blockIdx.x % nt32, mtile = blockIdx.x / nt32;
    const int phase = blockIdx.y, ph = phase / p.s, pw = phase - ph * p.s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const unsigned OOB = 0xF0000000u;
    // the phase's pixels: (a, b) -> dX pixel (a s + ph, b s + pw)
    const int Hp = ph < p.Hx ? (p.Hx - ph + p.s - 1) / p.s : 0, Wp = pw < p.Wx ? (p.Wx - pw + p.s - 1) / p.s : 0;
    const int HpWp = Hp * Wp, Mph = p.N * HpWp;
    const int cy = p.cnt[0][ph], cx = p.cnt[1][pw];
    const int ry0 = p.r0[0][ph], rx0 = p.r0[1][pw], dy0 = p.d0[0][ph], dx0 = p.d0[1][pw];
    // A row of this lane
    const int m = mtile * 32 + i;
    const bool vm = m < Mph;
    const int mm = vm ? m : 0;
    const int img = HpWp ? mm / HpWp : 0, rem = mm - img * HpWp;
    const int a = Wp ? rem / Wp : 0, b = rem - a * Wp;
    // weight row of this lane
    const int nb = ntile * 32 + i;
    // a lane owns row i and CW / 2 = 8 consecutive k values of a 16-wide chunk: two adjacent 16-byte loads per operand, so
    // that a row's two lanes use 64 bytes of a line at once (conv_ksplit_kernel in conv.hip has the measurements)
    constexpr int CW = 16, NL = CW / 8;
    const int kl = (CW / 2) * h;
    const unsigned boff = nb < p.Cx ? (unsigned)((size_t)nb * p.K + kl) * 4u : OOB;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy), 0, (int)((size_t)p.N * p.Hy * p.Wy * p.Cy * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.wd), 0, (int)((size_t)p.Cx * p.K * 4u), 0x00020000);
    const int cpt = p.Cy / CW;                      // chunks per tap
    const int nch = cy * cx * cpt;
    const int c0 = wave * nch / 8, c1 = (wave + 1) * nch / 8;
    struct Frag { u32x4 a[NL], b[NL]; };
    // position of the next chunk to load: (tap row, tap column, channel base), stepped; it stops at the wave's last chunk,
    // which the tail of the loop re-loads (never used) to stay straight-line
    int pos = c0, piy, pix, pcb;
    {
        const int tap = c0 / cpt;
        pcb = (c0 - tap * cpt) * CW; piy = cx ? tap / cx : 0; pix = tap - piy * cx;
    }
    auto load_next = [&]() {
        Frag f;
        const int ih = a + dy0 + piy * p.dstep[0], iw = b + dx0 + pix * p.dstep[1];
        const bool ok = vm && (unsigned)ih < (unsigned)p.Hy && (unsigned)iw < (unsigned)p.Wy;
        const unsigned aoff = ok ? (unsigned)(((img * p.Hy + ih) * p.Wy + iw) * p.Cy + pcb + kl) * 4u : OOB;
        const int kb4 = (((ry0 + piy * p.per[0]) * p.S + rx0 + pix * p.per[1]) * p.Cy + pcb) * 4;
#pragma unroll
        for (int q = 0; q < NL; ++q) f.a[q] = __builtin_amdgcn_raw_buffer_load_b128(xr, aoff + 16u * q, 0, 0);
#pragma unroll
        for (int q = 0; q < NL; ++q) f.b[q] = __builtin_amdgcn_raw_buffer_load_b128(wr, boff + 16u * q, kb4, 0);
        if (pos < c1 - 1) {
            ++pos;
            pcb += CW;
            if (pcb == p.Cy) { pcb = 0; if (++pix == cx) { pix = 0; ++piy; } }
        }
        return f;
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(f.a[q].x), __uint_as_float(f.b[q].x), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(f.a[q].y), __uint_as_float(f.b[q].y), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(f.a[q].z), __uint_as_float(f.b[q].z), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(f.a[q].w), __uint_as_float(f.b[q].w), acc, 0, 0, 0);
        }
    };
    if (c0 < c1) {
        // two chunks in flight beside the one being multiplied; three register sets in rotation (no copies)
        Frag f0 = load_next(), f1 = load_next(), f2;
        for (int ch = c0;;) {
            f2 = load_next(); mma(f0); if (++ch >= c1) break;
            f0 = load_next(); mma(f1); if (++ch >= c1) break;
            f1 = load_next(); mma(f2); if (++ch >= c1) break;
        }
    }
    // partial tiles -> LDS (C/D layout: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5))
#pragma unroll
    for (int e = 0; e < 16; ++e) part[wave][(e & 3) + 8 * (e >> 2) + 4 * h][i] = acc[e];
    __syncthreads();
    const int col = tid & 31, rg = tid >> 5;        // two rows per thread: rg, rg + 16
    const int n = ntile * 32 + col;
    const bool vn = n < p.Cx;
    const bool bnb = p.bnb_scale != nullptr;
    float bsc = 0.f, bsh = 0.f, bmu = 0.f, bis = 0.f;
    if (bnb && vn) { bsc = p.bnb_scale[n]; bsh = p.bnb_shift[n]; bmu = p.bnb_mean[n]; bis = p.bnb_invstd[n]; }
    float s1 = 0.f, s2 = 0.f, am = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = rg + 16 * j;
        const int mo = mtile * 32 + row;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += part[w][row][col];
        if (vn && mo < Mph) {
            const int im = mo / HpWp, rm = mo - im * HpWp;
            const int ra = rm / Wp, rb = rm - ra * Wp;
            const size_t o = ((size_t)(im * p.Hx + ra * p.s + ph) * p.Wx + rb * p.s + pw) * p.Cx + n;
            if (bnb) {
                const float xv = p.res1[o];
                if (p.bnb_relu && fmaf(xv, bsc, bsh) <= 0.f) v = 0.f;
                p.dx[o] = v;
                s1 += v;
                s2 = fmaf(v, (xv - bmu) * bis, s2);
            } else {
                v += p.res1 ? p.res1[o] : 0.f;
                p.dx[o] = v;
                am = fmaxf(am, fabsf(v));
            }
        }
    }
    if (p.amax) amax_commit(am, p.amax);
    if (p.stats) {
        __syncthreads();                             // every thread has read its part of `part`
        float* red = &part[0][0][0];                 // [16][32][2]
        red[(rg * 32 + col) * 2 + 0] = s1;
        red[(rg * 32 + col) * 2 + 1] = s2;
        __syncthreads();
        if (tid < 32 && vn) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) { a0 += red[(w * 32 + tid) * 2 + 0]; a1 += red[(w * 32 + tid) * 2 + 1]; }
            const size_t t = (size_t)phase * p.mtiles + mtile;
            p.stats[(t * 2 + 0) * p.Cx + n] = a0;
            p.stats[(t * 2 + 1) * p.Cx + n] = a1;
        }
    }
}

static int gcd_i(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }
static int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

static bool up_geom_ok(const dsnt_conv_geom* g) {
    if (!g || g->stride < 2 || g->stride > UP_MAX_STRIDE || g->Cout % 16 != 0) return false;
    if (g->N <= 0 || g->H <= 0 || g->W <= 0 || g->Cin <= 0 || g->R <= 0 || g->S <= 0 || g->dil <= 0 || g->pad < 0) return false;
    if ((size_t)g->N * g->Ho * g->Wo * g->Cout * 4u >= (1ull << 31) || (size_t)g->N * g->H * g->W * g->Cin * 4u >= (1ull << 31) ||
        (size_t)g->Cin * g->R * g->S * g->Cout * 4u >= (1ull << 31)) return false;
    return !dsnt_kernel_off("dgrad_up");
}
static int up_mtiles(const dsnt_conv_geom* g) {
    const long m0 = (long)g->N * ((g->H + g->stride - 1) / g->stride) * ((g->W + g->stride - 1) / g->stride);
    return (int)((m0 + 31) / 32);
}

extern "C" int dsnt_conv_dgrad_strided_ok(const dsnt_conv_geom* g) { return up_geom_ok(g) ? 1 : 0; }
extern "C" int dsnt_conv_dgrad_strided_tiles(const dsnt_conv_geom* g) {
    return up_geom_ok(g) ? g->stride * g->stride * up_mtiles(g) : 0;
}

extern "C" int dsnt_conv_dgrad_strided(const float* dy, const float* wd, float* dx, const float* res1, float* stats_partial,
                                       const dsnt_conv_geom* g, const dsnt_bn_bwd_epilogue* bnb, const dsnt_out_bounds* tail,
                                       void* stream) {
    DSNT_REQUIRE(g && dy && wd && dx, DSNT_ERR_ARG, "dsnt_conv_dgrad_strided: null argument");
    const int ho = (g->H + 2 * g->pad - g->dil * (g->R - 1) - 1) / g->stride + 1;
    const int wo = (g->W + 2 * g->pad - g->dil * (g->S - 1) - 1) / g->stride + 1;
    DSNT_REQUIRE(ho == g->Ho && wo == g->Wo, DSNT_ERR_SHAPE, "dsnt_conv_dgrad_strided: output %dx%d inconsistent with "
                 "input/filter (expected %dx%d)", g->Ho, g->Wo, ho, wo);
    DSNT_REQUIRE(up_geom_ok(g), DSNT_ERR_SHAPE, "dsnt_conv_dgrad_strided: needs 2 <= stride <= %d, Cout %% 16 == 0, "
                 "tensors under 2 GiB (ask dsnt_conv_dgrad_strided_ok)", UP_MAX_STRIDE);
    DSNT_REQUIRE(dsnt_aligned16(dy) && dsnt_aligned16(wd), DSNT_ERR_ALIGN, "dsnt_conv_dgrad_strided: dy / wd must be 16-byte aligned");
    DSNT_REQUIRE(!bnb || (bnb->x && bnb->scale && bnb->shift && bnb->mean && bnb->invstd && stats_partial && !res1), DSNT_ERR_ARG,
                 "dsnt_conv_dgrad_strided: the batch-norm-backward epilogue needs x/scale/shift/mean/invstd and stats_partial, "
                 "and excludes res1");
    DSNT_REQUIRE(bnb || !stats_partial, DSNT_ERR_ARG, "dsnt_conv_dgrad_strided: stats_partial goes with the batch-norm-backward epilogue");
    DSNT_REQUIRE(!tail || !tail->amax_bn, DSNT_ERR_ARG, "dsnt_conv_dgrad_strided: of dsnt_out_bounds only amax is supported");
    DSNT_REQUIRE(!(tail && tail->amax && bnb), DSNT_ERR_ARG, "dsnt_conv_dgrad_strided: dsnt_out_bounds.amax excludes the batch-norm-backward epilogue");
    UpP p;
    memset(&p, 0, sizeof(p));
    p.dy = dy; p.wd = wd; p.dx = dx; p.res1 = res1; p.stats = stats_partial;
    if (bnb) {
        p.res1 = bnb->x; p.bnb_scale = bnb->scale; p.bnb_shift = bnb->shift; p.bnb_mean = bnb->mean; p.bnb_invstd = bnb->invstd;
        p.bnb_relu = bnb->relu;
    }
    p.amax = tail ? reinterpret_cast<unsigned*>(tail->amax) : nullptr;
    p.N = g->N; p.Hy = g->Ho; p.Wy = g->Wo; p.Cy = g->Cout; p.Hx = g->H; p.Wx = g->W; p.Cx = g->Cin;
    p.R = g->R; p.S = g->S; p.s = g->stride; p.K = g->R * g->S * g->Cout;
    // flipped tap q of wd is the forward tap T - 1 - q: it reaches dY coordinate (x + pad - (T - 1 - q) dil) / stride
    const int s = g->stride, gg = gcd_i(s, g->dil);
    for (int ax = 0; ax < 2; ++ax) {
        const int T = ax == 0 ? g->R : g->S;
        const int off = g->pad - g->dil * (T - 1);           // x + off + q dil
        p.per[ax] = s / gg;
        p.dstep[ax] = p.per[ax] * g->dil / s;
        for (int ph = 0; ph < s; ++ph) {
            int q0 = -1;
            for (int q = 0; q < p.per[ax] && q < T; ++q)
                if (((ph + off + q * g->dil) % s + s) % s == 0) { q0 = q; break; }
            p.r0[ax][ph] = q0 < 0 ? 0 : q0;
            p.cnt[ax][ph] = q0 < 0 ? 0 : (T - 1 - q0) / p.per[ax] + 1;
            p.d0[ax][ph] = q0 < 0 ? 0 : floor_div(ph + off + q0 * g->dil, s);
        }
    }
    p.mtiles = up_mtiles(g);
    const int nt32 = (p.Cx + 31) / 32;
    DSNT_LAUNCH(conv_dgrad_up_kernel, dim3(p.mtiles * nt32, s * s), dim3(512), 0, (hipStream_t)stream, p);
    DSNT_CHECK_LAUNCH("dsnt_conv_dgrad_strided");
}
