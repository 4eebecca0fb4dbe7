import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr, ConvGeom
dev = torch.device('cuda:0')
B = 32
shapes = [(64, 128, 128, 3), (64, 256, 128, 1), (64, 128, 256, 1), (32, 128, 128, 3), (32, 128, 256, 1), (16, 128, 128, 3), (8, 128, 128, 3), (128, 64, 64, 3), (64, 256, 256, 1)]
def timeit(fn, args, iters=10):
    st = torch.cuda.current_stream().cuda_stream
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
    b = torch.zeros(Cout, device=dev); sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
    y = torch.empty(B, H, H, Cout, device=dev); y6 = torch.empty_like(y)
    stats = torch.empty((M + 31) // 32, 2, Cout, device=dev)
    planes = torch.empty(3 * w.numel(), dtype=torch.bfloat16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert _lib.fn('dsnt_split_bf16x3')(ptr(w), ptr(planes), w.numel(), st) == 0
    flops = 2.0 * M * K * Cout
    for pro in (True, False):
        a32 = (ptr(x), ptr(w), ptr(b), ptr(y), ptr(sc) if pro else None, ptr(sh) if pro else None, 1, None, None, ptr(stats), C.byref(g))
        a6 = (ptr(x), ptr(planes), w.numel(), ptr(b), ptr(y6), ptr(sc) if pro else None, ptr(sh) if pro else None, 1, None, None, ptr(stats), C.byref(g))
        t32 = timeit(_lib.fn('dsnt_conv_fwd'), a32); t6 = timeit(_lib.fn('dsnt_conv_fwd_bf16x6'), a6)
        err = (y - y6).abs().max().item() / y.abs().max().item()
        print('H%3d %3d->%3d k%d pro=%d | fp32 %7.1f us %6.1f TF | bf16x6 %7.1f us %6.1f TF-equiv | x%.2f | rel diff %.1e' % (
            H, Cin, Cout, k, pro, t32 * 1e6, flops / t32 / 1e12, t6 * 1e6, flops / t6 / 1e12, t32 / t6, err))

print('--- wgrad')
for (H, Cin, Cout, k) in shapes:
    g = ConvGeom(B, H, H, Cin, H, H, Cout, k, k, 1, k // 2, 1)
    M = B * H * H; K = k * k * Cin
    x = torch.randn(B, H, H, Cin, device=dev); gy = torch.randn(B, H, H, Cout, device=dev)
    sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
    ws = torch.empty(_lib.fn('dsnt_conv_wgrad_ws_floats')(C.byref(g)), device=dev)
    dw = torch.empty(Cout, k, k, Cin, device=dev); dw6 = torch.empty_like(dw); db = torch.empty(Cout, device=dev)
    flops = 2.0 * M * K * Cout
    t32 = timeit(_lib.fn('dsnt_conv_wgrad'), (ptr(x), ptr(sc), ptr(sh), 1, ptr(gy), ptr(ws), ptr(dw), ptr(db), 0, C.byref(g)))
    t6 = timeit(_lib.fn('dsnt_conv_wgrad_bf16x6'), (ptr(x), ptr(sc), ptr(sh), 1, ptr(gy), ptr(ws), ptr(dw6), ptr(db), 0, C.byref(g)))
    print('H%3d %3d->%3d k%d | fp32 %7.1f us %6.1f TF | bf16x6 %7.1f us %6.1f TF-equiv | x%.2f | rel diff %.1e' % (
        H, Cin, Cout, k, t32 * 1e6, flops / t32 / 1e12, t6 * 1e6, flops / t6 / 1e12, t32 / t6, (dw - dw6).abs().max().item() / dw.abs().max().item()))
