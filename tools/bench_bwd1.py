"""Time dsnt_conv1x1_bwd_f16x3 (csrc/bwd1.hip) against the three launches it replaces — dsnt_bn_act_bwd_apply_amax +
the 1x1 data gradient with the BatchNorm-backward epilogue (gemm1) + dsnt_conv_wgrad_f16x3 (wgrad1) — on the hg2 shapes.
Usage: python tools/bench_bwd1.py [batch]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr, call, ConvGeom, BnBwdEpilogue, BnBwdApply
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (H, Cin, Cout, apply) in [(64, 256, 128, True), (64, 128, 256, False), (64, 128, 128, True), (32, 256, 128, True),
                              (32, 128, 256, False), (128, 64, 64, True), (128, 64, 128, False), (64, 256, 256, False), (16, 256, 128, True), (16, 128, 256, False)]:
    M = B * H * H
    g = ConvGeom(B, H, H, Cin, H, H, Cout, 1, 1, 1, 0, 1)
    gd = ConvGeom(B, H, H, Cout, H, H, Cin, 1, 1, 1, 0, 1)
    if not _lib.fn('dsnt_conv1x1_bwd_ok')(C.byref(g)):
        print('H%d %d->%d: not supported' % (H, Cin, Cout))
        continue
    x = torch.randn(M, Cin, device=dev)
    sc = torch.rand(Cin, device=dev) + 0.5
    sh = torch.randn(Cin, device=dev) * 0.1
    mu = torch.randn(Cin, device=dev) * 0.1
    istd = torch.rand(Cin, device=dev) + 0.5
    wd = torch.randn(Cin, Cout, device=dev) * 0.05
    wb = torch.zeros(64, device=dev)
    call('dsnt_amax', ptr(wd), wd.numel(), ptr(wb))
    planes = torch.empty(2 * wd.numel(), dtype=torch.float16, device=dev)
    call('dsnt_split_f16x2', ptr(wd), ptr(planes), wd.numel(), wd.numel(), ptr(wb))
    dz = torch.randn(M, Cout, device=dev) * 1e-3
    y = torch.randn(M, Cout, device=dev)
    ysc, ymu, yis = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1, torch.rand(Cout, device=dev) + 0.5
    coef = torch.randn(2, Cout, device=dev) * 1e-5
    ab = torch.full((64,), 40.0, device=dev)
    gb = torch.full((64,), 0.05, device=dev)
    xs = BnBwdEpilogue(ptr(x), ptr(sc), ptr(sh), ptr(mu), ptr(istd), 1)
    ap = BnBwdApply(ptr(y), ptr(ysc), ptr(ymu), ptr(yis), ptr(coef))
    splits = _lib.fn('dsnt_conv1x1_bwd_splits')(C.byref(g), 0)
    ws = torch.empty(_lib.fn('dsnt_conv1x1_bwd_ws_floats')(C.byref(g), 0), device=dev)
    stats = torch.empty(splits, 2, Cin, device=dev)
    dzx = torch.empty(M, Cin, device=dev)
    st = lambda: torch.cuda.current_stream().cuda_stream
    fused = _lib.fn('dsnt_conv1x1_bwd_f16x3')

    def run_fused():
        assert fused(C.byref(xs), ptr(dz), C.byref(ap) if apply else None, ptr(planes), wd.numel(), ptr(wb), ptr(ab), ptr(gb),
                     ptr(dzx), ptr(stats), ptr(ws), None, 0, C.byref(g), st()) == 0
    t_f = timeit(run_fused)
    # the launches it replaces
    dy = torch.empty(M, Cout, device=dev)
    amax = torch.zeros(64, device=dev)
    part = torch.empty((M + 127) // 128, 2, Cin, device=dev)
    ws1 = torch.empty(_lib.fn('dsnt_conv_wgrad_f16x3_ws_floats')(C.byref(g), 2), device=dev)
    fa, fd, fw = _lib.fn('dsnt_bn_act_bwd_apply_amax'), _lib.fn('dsnt_conv_fwd_f16x3_ex'), _lib.fn('dsnt_conv_wgrad_f16x3')

    def run_apply():
        assert fa(ptr(dz), ptr(y), ptr(ysc), ptr(ysc), ptr(ymu), ptr(yis), ptr(coef), 0, ptr(dy), 0, M, Cout, ptr(amax), st()) == 0

    def run_dgrad():
        assert fd(ptr(dy if apply else dz), ptr(planes), wd.numel(), ptr(wb), ptr(gb), None, ptr(dzx), None, None, 0, None, None,
                  ptr(part), C.byref(gd), C.byref(xs), None, st()) == 0

    def run_wgrad():
        assert fw(ptr(x), ptr(sc), ptr(sh), 1, ptr(dy if apply else dz), ptr(ws1), None, None, 2, ptr(ab), ptr(gb), C.byref(g), st()) == 0
    t_a = timeit(run_apply) if apply else 0.0
    t_d, t_w = timeit(run_dgrad), timeit(run_wgrad)
    mb_f = 4e-6 * M * ((2 if apply else 1) * Cout + 2 * Cin)
    mb_old = 4e-6 * M * ((3 * Cout if apply else 0) + (Cout + 2 * Cin) + (Cout + Cin))
    print('H%3d %3d->%3d %s M=%7d | fused %6.1f us (%5.0f MB, %.2f TB/s) | apply %5.1f + dgrad %5.1f + wgrad %5.1f = %6.1f us (%5.0f MB)'
          % (H, Cin, Cout, 'apply' if apply else 'given', M, t_f, mb_f, mb_f / t_f, t_a, t_d, t_w, t_a + t_d + t_w, mb_old))
