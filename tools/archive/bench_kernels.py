"""Time the conv kernels (fwd with BN prologue, dgrad, wgrad) on the hourglass shapes: TFLOP/s each."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr, ConvGeom
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [  # H, Cin, Cout, k, count per hg2 step(fwd)
    (64, 128, 128, 3), (64, 256, 128, 1), (64, 128, 256, 1), (32, 128, 128, 3), (32, 256, 128, 1), (32, 128, 256, 1),
    (16, 128, 128, 3), (16, 128, 256, 1), (8, 128, 128, 3), (4, 128, 128, 3), (128, 64, 64, 3), (64, 256, 256, 1), (64, 256, 16, 1)]
def timeit(fn, args, iters=10):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2): assert fn(*args, st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn(*args, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for (H, Cin, Cout, k) in shapes:
    pad = k // 2
    g = ConvGeom(B, H, H, Cin, H, H, Cout, k, k, 1, pad, 1)
    M = B * H * H; K = k * k * Cin
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    b = torch.zeros(Cout, device=dev); sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
    y = torch.empty(B, H, H, Cout, device=dev); gy = torch.randn(B, H, H, Cout, device=dev)
    bm = _lib.fn('dsnt_conv_fwd_bm')(C.byref(g)); stats = torch.empty((M + bm - 1) // bm, 2, Cout, device=dev)
    flops = 2.0 * M * K * Cout
    t_f = timeit(_lib.fn('dsnt_conv_fwd'), (ptr(x), ptr(w), ptr(b), ptr(y), ptr(sc), ptr(sh), 1, None, None, ptr(stats), C.byref(g)))
    wd = torch.empty(Cin, k, k, Cout, device=dev)
    gd = ConvGeom(B, H, H, Cout, H, H, Cin, k, k, 1, pad, 1)
    da = torch.empty(B, H, H, Cin, device=dev)
    t_d = timeit(_lib.fn('dsnt_conv_fwd'), (ptr(gy), ptr(wd), None, ptr(da), None, None, 0, None, None, None, C.byref(gd)))
    ws = torch.empty(_lib.fn('dsnt_conv_wgrad_ws_floats')(C.byref(g)), device=dev); dw = torch.empty_like(w); db = torch.empty(Cout, device=dev)
    t_w = timeit(_lib.fn('dsnt_conv_wgrad'), (ptr(x), ptr(sc), ptr(sh), 1, ptr(gy), ptr(ws), ptr(dw), ptr(db), 0, C.byref(g)))
    print('H%3d %3d->%3d k%d  M=%7d  GFLOP %6.2f | fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | wgrad(+reduce) %7.1f us %6.1f TF' % (
        H, Cin, Cout, k, M, flops / 1e9, t_f * 1e6, flops / t_f / 1e12, t_d * 1e6, flops / t_d / 1e12, t_w * 1e6, flops / t_w / 1e12))
