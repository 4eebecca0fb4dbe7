// Internal interface of the LDS-staged streaming 1x1 forward kernel (fwd1.hip).
#pragma once
#include "conv_split.h"

struct Fwd1Plan {
    int ok;
    int cfg;            // index into the shape table of fwd1.hip
    int chunks;         // column chunks (gridDim.y)
    int nstages;        // 32-pixel stages
    int spw;            // stages per workgroup
    int nwg;            // workgroups (gridDim.x) = rows of the statistics partials
    int lds;
};
Fwd1Plan dsnt_fwd1_plan(const dsnt_conv_geom* g, bool share);
