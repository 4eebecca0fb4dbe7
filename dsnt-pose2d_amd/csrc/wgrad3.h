// Internal interface of the halo weight-gradient kernels (wgrad3.hip), used by the dispatch in conv.hip.
#pragma once
#include "common.h"

// Plan of one launch: which instantiation, how the pixels are cut into slabs.  ok == 0: geometry not supported.
struct Wg3Plan {
    int ok;
    int cfg;            // 0: 64 x 128 channels per workgroup (9 taps per wave), 1: 64 x 64 (taps split over two wave
                        // groups); a cfg-0 launch that SHARES the chip runs as 32 x 128 four-wave workgroups instead
                        // (same slabs: twice the channel chunks)
    int WS;             // strip width in pixels (16, 32 or 64)
    int strips;         // W / WS
    int rps;            // output rows per slab (divides H)
    int hsplits;        // H / rps
    int kchunks;        // Cin / 64
    int nchunks;        // Cout / (128 or 64)
    int nslabs;         // N * hsplits * strips: slabs ws[nslabs][Cout][K] (+ [nslabs][Cout] bias partials)
    int blocks;         // workgroups
    int lds;            // dynamic LDS bytes
};
// share = DSNT_WGRAD_SHARE_CHIP: the plan of a launch that runs beside a dependency chain (fewer, longer slabs)
Wg3Plan dsnt_wg3_plan(const dsnt_conv_geom* g, int share);      // share: 0 whole chip, 1 beside a chain, 2 the same, narrower
// enqueue (or record) the launch; tensors as dsnt_conv_wgrad_f16x3 (in_scale / in_shift may be null); share =
// DSNT_WGRAD_SHARE_CHIP: the launch runs beside a dependency chain on another stream
void dsnt_wg3_launch(const Wg3Plan& pl, const float* x, const float* in_scale, const float* in_shift, int in_relu,
                     const float* dy, float* ws, const float* a_bound, const float* g_bound, const dsnt_conv_geom* g,
                     hipStream_t st, int share);

// ---- 1x1 weight gradients (wgrad1.hip): a four-wave workgroup owns ck x cn channels of the weight matrix for its pixels
struct Wg1Plan {
    int ok;
    int ck, cn;             // input / output channels per workgroup (64, 128 or 256)
    int kchunks, nchunks;   // Cin / ck, Cout / cn
    int nsplits;            // slabs ws[nsplits][Cout][Cin] (+ [nsplits][Cout] bias partials)
    int rows_per_split;     // multiple of 16
    int blocks, lds;
};
Wg1Plan dsnt_wg1_plan(const dsnt_conv_geom* g, bool share);
void dsnt_wg1_launch(const Wg1Plan& pl, const float* x, const float* in_scale, const float* in_shift, int in_relu,
                     const float* dy, float* ws, const float* a_bound, const float* g_bound, const dsnt_conv_geom* g,
                     hipStream_t st);
