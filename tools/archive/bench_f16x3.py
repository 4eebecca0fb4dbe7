"""fp16x3 vs bf16x6 vs fp32-MFMA forward convolution: time and difference to bf16x6 (which is ~6e-9 from fp64)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr, ConvGeom
dev = torch.device('cuda:0')
B = 32
shapes = [(64, 128, 128, 3), (64, 256, 128, 1), (64, 128, 256, 1), (64, 256, 256, 1), (32, 128, 256, 1), (32, 256, 128, 1), (32, 128, 128, 3), (128, 64, 64, 3)]
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, args, iters=10):
    for _ in range(2): assert fn(*args, st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn(*args, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for (H, Cin, Cout, k) in shapes:
    g = ConvGeom(B, H, H, Cin, H, H, Cout, k, k, 1, k // 2, 1)
    M = B * H * H; K = k * k * Cin
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    b = torch.randn(Cout, device=dev) * 0.1; sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
    y32, y6, y16 = (torch.empty(B, H, H, Cout, device=dev) for _ in range(3))
    stats = torch.empty((M + 127) // 128, 2, Cout, device=dev)
    planes = torch.empty(3 * w.numel(), dtype=torch.bfloat16, device=dev)
    assert _lib.fn('dsnt_split_bf16x3')(ptr(w), ptr(planes), w.numel(), st) == 0
    planes16 = torch.empty(2 * w.numel(), dtype=torch.float16, device=dev)
    wb = torch.zeros(64, device=dev); ab = torch.zeros(64, device=dev)
    assert _lib.fn('dsnt_amax')(ptr(w), w.numel(), ptr(wb), st) == 0
    assert _lib.fn('dsnt_split_f16x2')(ptr(w), ptr(planes16), w.numel(), w.numel(), ptr(wb), st) == 0
    flops = 2.0 * M * K * Cout
    for pro in (True, False):
        a = torch.relu(x * sc + sh) if pro else x
        ab.fill_(a.abs().max().item() * 3.0)          # a loose bound on purpose
        scp, shp = (ptr(sc), ptr(sh)) if pro else (None, None)
        a32 = (ptr(x), ptr(w), ptr(b), ptr(y32), scp, shp, 1, None, None, ptr(stats), C.byref(g))
        a6 = (ptr(x), ptr(planes), w.numel(), ptr(b), ptr(y6), scp, shp, 1, None, None, ptr(stats), C.byref(g))
        a16 = (ptr(x), ptr(planes16), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y16), scp, shp, 1, None, None, ptr(stats), C.byref(g), None, None)
        t32 = timeit(_lib.fn('dsnt_conv_fwd'), a32); t6 = timeit(_lib.fn('dsnt_conv_fwd_bf16x6'), a6)
        t16 = timeit(_lib.fn('dsnt_conv_fwd_f16x3_ex'), a16)
        sc_ = y6.abs().max().item()
        print('H%3d %3d->%3d k%d pro=%d | fp32 %6.1f us | bf16x6 %6.1f us | fp16x3 %6.1f us (x%.2f) %6.1f TF-equiv | max|y-y6|/max|y6|: fp32 %.1e  fp16x3 %.1e' % (
            H, Cin, Cout, k, pro, t32 * 1e6, t6 * 1e6, t16 * 1e6, t6 / t16, flops / t16 / 1e12,
            (y32 - y6).abs().max().item() / sc_, (y16 - y6).abs().max().item() / sc_))
