// Internal interface of the stem's weight-gradient kernel (stem4.hip), used by the dispatch in conv.hip.
#pragma once
#include "common.h"

// slabs ws[nslabs][64][256] (+ [nslabs][64] bias partials) of a launch: one per workgroup; 0 = geometry not supported
int dsnt_stem4_wgrad_slabs(const dsnt_conv_geom* g);
// tensors as dsnt_conv_wgrad_f16x3 without a BatchNorm prologue (the stem's operand is the raw space-to-depth image)
void dsnt_stem4_wgrad_launch(const float* x, const float* dy, float* ws, const float* a_bound, const float* g_bound,
                             const dsnt_conv_geom* g, hipStream_t st);
