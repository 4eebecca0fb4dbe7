// The persistent low-resolution stage (round 6): runs of small dependent launches of ONE lane of a launch list — the 8 x 8 and
// 4 x 4 levels of an hourglass (/root/reference/src/dsnt/hourglass.py:78-90), ~70 launches per hourglass and step, each a few
// microseconds of work behind a ~5-us launch boundary on the step's critical path — replayed as ONE launch: a persistent kernel
// whose workgroups walk the recorded launches' virtual workgroup indices through the very device functions the stand-alone kernels
// are made of (bit-identical results), with a chip-wide barrier where a kernel boundary was.
//
// Host side (api.cpp): a launch that can take part is recorded with DSNT_LAUNCH_OP — its code, grid and parameter block besides
// the usual closure; dsnt_list_fuse replaces every run of >= min_run such entries on one lane (unbroken by a lane synchronisation
// or a bucket mark that involves the lane) by one dsnt_stage_launch over a table of DsntStageOp in caller-owned device memory.
#pragma once
#include "common.h"

#define DSNT_STAGE_NT 512                 // threads per stage workgroup (the K-split convolution's eight waves)
#define DSNT_STAGE_PARAM_BYTES 496

enum DsntStageCode {
    DSNT_ST_NONE = 0,
    DSNT_ST_KSPLIT_PRO,                   // conv_ksplit_kernel<true, 16>   (ConvP)
    DSNT_ST_KSPLIT,                       // conv_ksplit_kernel<false, 16>  (ConvP)
    DSNT_ST_APPLY_FIXED,                  // bn_act_bwd_apply_kernel<true>  (BnApplyP)
    DSNT_ST_APPLY,                        // bn_act_bwd_apply_kernel<false> (BnApplyP)
    DSNT_ST_TILE_POOL,                    // tile_op_stats_kernel<0>        (TileOpP)
    DSNT_ST_TILE_UPADD,                   // tile_op_stats_kernel<1>        (TileOpP)
    DSNT_ST_POOL_BWD,                     // maxpool2_bwd_kernel            (PoolBwdP)
    DSNT_ST_UP_BWD,                       // upsample2_bwd_kernel           (UpBwdP)
    DSNT_ST_FIN_FWD,                      // bn_finalize_kernel<0>          (BnFinP; 1024 virtual threads)
    DSNT_ST_FIN_BWD,                      // bn_finalize_kernel<1>          (BnFinP)
    DSNT_ST_CODES
};

struct alignas(16) DsntStageOp {
    int code;                             // DsntStageCode
    int gx, gy;                           // the recorded grid (virtual workgroups: gx * gy)
    int nt;                               // the recorded workgroup size (256, 512; 1024: bn_finalize, two virtual threads per thread)
    unsigned char params[DSNT_STAGE_PARAM_BYTES];
};
static_assert(sizeof(DsntStageOp) == 512, "DsntStageOp");

// sync: 4 zeroed uint32 per stage launch, owned by it alone — [0] arrivals, [1] workgroups done, [2] error (a barrier gave up).
// The launch restores [0] and [1] to zero when its last workgroup ends.
#define DSNT_STAGE_SYNC_WORDS 4
void dsnt_stage_launch(const DsntStageOp* ops_dev, int nops, unsigned* sync_dev, int grid, hipStream_t st);
