"""dsnt — the dsnt-pose2d hot path on AMD Instinct MI355X (gfx950).

Drop-in for the reference package of the same name for the training/inference hot path:
`dsnt.model.build_mpii_pose_model`, the model methods and the `dsnt.nn` operators.  Device work
is done by hand-written HIP kernels in `../csrc/libdsnt_hip.so` (C ABI: `include/dsnt_hip.h`),
loaded lazily on the first operator call; there is no CPU fallback.
"""
__all__ = ['nn', 'model', 'hourglass', 'data', 'evaluator', 'optim', 'parallel', 'synthetic']
