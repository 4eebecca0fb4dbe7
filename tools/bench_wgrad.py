"""Time the bf16x6 and fp16x3 weight-gradient kernels alone (slabs only, no reduction) on the hourglass shapes."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr, ConvGeom
dev = torch.device('cuda:0')
B = 32
shapes = [(64, 128, 128, 3), (64, 256, 128, 1), (64, 128, 256, 1), (32, 128, 128, 3), (32, 128, 256, 1), (16, 128, 128, 3), (128, 64, 64, 3), (64, 256, 256, 1)]
def timeit(fn, args, iters=20):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): assert fn(*args, st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn(*args, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for (H, Cin, Cout, k) in shapes:
    g = ConvGeom(B, H, H, Cin, H, H, Cout, k, k, 1, k // 2, 1)
    M = B * H * H; K = k * k * Cin
    x = torch.randn(B, H, H, Cin, device=dev); gy = torch.randn(B, H, H, Cout, device=dev)
    sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
    ws = torch.empty(_lib.fn('dsnt_conv_wgrad_ws_floats')(C.byref(g)), device=dev)
    flops = 2.0 * M * K * Cout
    t6 = timeit(_lib.fn('dsnt_conv_wgrad_bf16x6'), (ptr(x), ptr(sc), ptr(sh), 1, ptr(gy), ptr(ws), None, None, 0, C.byref(g)))
    ab, gb = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    ab.fill_(float(torch.relu(x * sc + sh).max()) * 4.0)
    assert _lib.fn('dsnt_amax')(ptr(gy), gy.numel(), ptr(gb), torch.cuda.current_stream().cuda_stream) == 0
    t16 = timeit(_lib.fn('dsnt_conv_wgrad_f16x3'), (ptr(x), ptr(sc), ptr(sh), 1, ptr(gy), ptr(ws), None, None, 0, ptr(ab), ptr(gb), C.byref(g)))
    print('H%3d %3d->%3d k%d | wgrad bf16x6 %7.1f us %6.1f TF-equiv (%.0f%% of 416.7) | fp16x3 %7.1f us %6.1f TF-equiv (%.0f%% of 833.3)' % (
        H, Cin, Cout, k, t6 * 1e6, flops / t6 / 1e12, flops / t6 / 4.167e12, t16 * 1e6, flops / t16 / 1e12, flops / t16 / 8.333e12))
