// Internal interface of the streaming 1x1 convolution kernel (gemm1.hip), used by the dispatch in conv.hip.
#pragma once
#include "conv_split.h"

// > 0: 32-column tiles per wave of the instantiation that runs this launch (fp16x3, 1x1 / stride 1, K in {64, 128, 256},
// enough rows); -1: not eligible, the tiled implicit-GEMM kernel runs it.
int dsnt_gemm1_cfg(const ConvP& p);
void dsnt_gemm1_launch(const ConvP& p, int ntw, bool pro, hipStream_t st, bool share = false);
