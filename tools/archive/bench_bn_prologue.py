"""BatchNorm finalisation as a separate launch vs in the consumer convolution's prologue, back to back on one stream
(the chain of the 8x8 / 4x4 hourglass levels):  python tools/bench_bn_prologue.py"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr, ConvGeom, BnPrologue
dev = torch.device('cuda:0')
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for (N, H, Cin, Cout, k, tiles_rows) in ((32, 8, 128, 128, 3, 32), (32, 8, 256, 128, 1, 128), (32, 8, 256, 128, 1, 32), (32, 4, 128, 128, 3, 32),
                                         (32, 4, 256, 128, 1, 32), (32, 16, 256, 128, 1, 128)):
    g = ConvGeom(N, H, H, Cin, H, H, Cout, k, k, 1, k // 2, 1)
    M = N * H * H
    tiles = (M + tiles_rows - 1) // tiles_rows
    x = torch.randn(N, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05; b = torch.zeros(Cout, device=dev)
    y = torch.empty(N, H, H, Cout, device=dev)
    part = torch.randn(tiles, 2, Cin, device=dev).abs() + 1.0
    part[:, 1] += 50.0
    gamma, beta = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    rm, rv = torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev)
    vec = [torch.empty(Cin, device=dev) for _ in range(4)]
    bm = lib.dsnt_conv_fwd_bm(C.byref(g))
    stats = torch.empty((M + bm - 1) // bm, 2, Cout, device=dev)
    pro = BnPrologue(ptr(part), tiles, Cin, M, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 0.1, 1e-5, *[ptr(v) for v in vec])
    ok = lib.dsnt_conv_fwd_pro_ok(C.byref(g), tiles, Cin)

    def sep():
        lib.dsnt_bn_finalize(ptr(part), tiles, M, Cin, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 0.1, 1e-5, 1, *[ptr(v) for v in vec], st)
        lib.dsnt_conv_fwd_ex(ptr(x), ptr(w), ptr(b), ptr(y), ptr(vec[2]), ptr(vec[3]), 1, None, None, ptr(stats), C.byref(g), None, None, st)

    def fused():
        lib.dsnt_conv_fwd_pro(ptr(x), ptr(w), ptr(b), ptr(y), C.byref(pro), 1, None, None, ptr(stats), C.byref(g), None, st)

    def conv_only():
        lib.dsnt_conv_fwd_ex(ptr(x), ptr(w), ptr(b), ptr(y), ptr(vec[2]), ptr(vec[3]), 1, None, None, ptr(stats), C.byref(g), None, None, st)

    def timeit(fn, n=200):
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000 / n
    t_conv, t_sep = timeit(conv_only), timeit(sep)
    t_fused = timeit(fused) if ok else float('nan')
    print('%2dx%-2d %3d->%3d k%d, %3d tiles x %3d ch (%5.1f KB): conv alone %5.1f us | finalize + conv %5.1f us | fused %5.1f us' % (
        H, H, Cin, Cout, k, tiles, Cin, tiles * 2 * Cin * 4 / 1024, t_conv, t_sep, t_fused))
