// Internal interface of the one-pass 1x1 backward kernel (bwd1.hip).
#pragma once
#include "conv_split.h"

// Plan of one launch.  ok == 0: geometry not supported (or DSNT_OFF=bwd1).
struct Bwd1Plan {
    int ok;
    int cfg;            // index into the shape table of bwd1.hip
    int chunks;         // column chunks (gridDim.y)
    int nstages;        // 32-pixel stages
    int spw;            // stages per workgroup
    int nwg;            // workgroups = slabs ws[nwg][Cout][Cin] (+ [nwg][Cout] bias partials) = rows of the statistics partials
    int lds;
};
Bwd1Plan dsnt_bwd1_plan(const dsnt_conv_geom* g, bool share);
void dsnt_bwd1_launch(const Bwd1Plan& pl, const dsnt_bn_bwd_epilogue* xs, const float* dy, const dsnt_bn_bwd_apply* ap,
                      const void* wd_planes, int64_t plane_stride, const float* w_bound, const float* a_bound,
                      const float* g_bound, float* dz_out, float* stats, float* ws, float* dz_amax, int accumulate,
                      const dsnt_conv_geom* g, hipStream_t st);
