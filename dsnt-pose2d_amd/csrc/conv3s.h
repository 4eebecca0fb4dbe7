// Internal interface of the symmetric 3x3 convolution kernel (conv3s.hip), used by the dispatch in conv.hip.
#pragma once
#include "conv_split.h"

// true: conv3s runs this launch (fp16x3 with the weights in the STREAM layout of dsnt_f16_prep_weights, 3x3 / stride 1 /
// pad 1, H % 4 == 0 and W % 32 == 0 or H % 8 == 0 and W % 16 == 0, Cin % 32 == 0 and <= 128, Cout 64 or 128, at most one
// residual)
bool dsnt_conv3s_ok(const ConvP& p);
// geometry part of the same test (the engine asks before it chooses the weight layout)
bool dsnt_conv3s_geom_ok(const dsnt_conv_geom* g);
// share: the launch runs on a lane beside the dependency chain — fewer persistent workgroups, so that the chain's small
// kernels find a CU with free LDS (two conv3s workgroups fill a CU's 160 KB)
void dsnt_conv3s_launch(const ConvP& p, bool pro, hipStream_t st, bool share = false);
// which form of the kernel the launch of `mode` (0 / 1 forward or plain data gradient, 3 BatchNorm-backward epilogue, 4 folded
// BatchNorm backward) takes on this geometry: bit 0 column split, bit 1 the 16x16x32 form, bit 2 8 x 16 patches; -1: not conv3s's
int dsnt_conv3s_form_of(const dsnt_conv_geom* g, int mode);
