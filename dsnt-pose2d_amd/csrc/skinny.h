// Skinny 1x1 convolutions (round 6): the score / re-injection convolutions of the intermediate supervision
// (/root/reference/src/dsnt/hourglass.py:166-175: `score` 256 -> 16, `score_` 16 -> 256) and their data gradients are pure
// streaming passes — 16 channels on one side — that the 128 x 64 / 128 x 128 matrix-core tiles run at 1.7-2.4 TB/s.
#pragma once
#include "conv_split.h"

// true: the launch was taken (enqueued or recorded).  f16: two fp16 weight planes + w_bound (else three bf16 planes).
bool dsnt_skinny_launch(const ConvP& p, bool f16, hipStream_t st);
