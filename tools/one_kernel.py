"""Run only the dominant conv kernel a few times (for rocprofv3 --pmc)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr, ConvGeom
dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
B = 32
if which in ('stem4', 'stem4w'):
    # the stem's own kernels (csrc/stem4.hip): 4x4 / stride 1 / pad 1 on the 16-channel space-to-depth image, 64 channels, batch 32
    Hs, Cin, Cout, k = 129, 16, 64, 4
    g = ConvGeom(B, Hs, Hs, Cin, 128, 128, Cout, k, k, 1, 1, 1)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(B, Hs, Hs, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    y = torch.empty(B, 128, 128, Cout, device=dev); gy = torch.randn(B, 128, 128, Cout, device=dev) * 1e-3
    planes16 = torch.empty(2 * w.numel(), dtype=torch.float16, device=dev)
    wb, ab, gb = torch.zeros(64, device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    for t_, s_ in ((w, wb), (x, ab), (gy, gb)):
        _lib.fn('dsnt_amax')(ptr(t_), t_.numel(), ptr(s_), st)
    _lib.fn('dsnt_split_f16x2')(ptr(w), ptr(planes16), w.numel(), w.numel(), ptr(wb), st)
    stats = torch.empty(_lib.fn('dsnt_stem4_fwd_stats_rows')(C.byref(g)), 2, Cout, device=dev)
    ws = torch.empty(_lib.fn('dsnt_conv_wgrad_f16x3_ws_floats')(C.byref(g), 0), device=dev)
    for it in range(int(os.environ.get('ONE_KERNEL_REPS', '5'))):
        if which == 'stem4':
            assert _lib.fn('dsnt_stem4_fwd_f16x3')(ptr(x), ptr(planes16), w.numel(), ptr(wb), ptr(ab), None, ptr(y), ptr(stats), C.byref(g), None, st) == 0
        else:
            assert _lib.fn('dsnt_conv_wgrad_f16x3')(ptr(x), None, None, 0, ptr(gy), ptr(ws), None, None, 0, ptr(ab), ptr(gb), C.byref(g), st) == 0
    torch.cuda.synchronize()
    sys.exit(0)
if which == 'head':
    # the DSNT head's two train-step kernels at batch 1024 (16384 rows of 64 x 64): dsnt_head_fwd, dsnt_head_loss_grad with JS
    rows, h, w = 1024 * 16, 64, 64
    logits = torch.randn(rows, h * w, device=dev) * 3
    hm = torch.empty_like(logits); g0 = torch.empty_like(logits)
    coords = torch.empty(rows, 2, device=dev); target = torch.rand(rows, 2, device=dev) * 2 - 1
    mask = torch.ones(rows, device=dev); dist = torch.empty(rows, device=dev); reg = torch.empty(rows, device=dev)
    denom2 = torch.empty(2, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert _lib.fn('dsnt_mask_denom')(ptr(mask), ptr(denom2), rows, st) == 0
    for it in range(int(os.environ.get('ONE_KERNEL_REPS', '5'))):
        assert _lib.fn('dsnt_head_fwd')(ptr(logits), ptr(hm), ptr(coords), rows, h, w, st) == 0
        assert _lib.fn('dsnt_head_loss_grad')(ptr(hm), ptr(coords), ptr(target), ptr(mask), ptr(denom2), ptr(dist), ptr(reg), ptr(g0),
                                               rows, h, w, 2.0 / 64, 0, 1.0, st) == 0
    torch.cuda.synchronize()
    sys.exit(0)
H, Cin, Cout, k = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (64, 128, 128, 3)
g = ConvGeom(B, H, H, Cin, H, H, Cout, k, k, 1, k // 2, 1)
x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
b = torch.zeros(Cout, device=dev); sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
y = torch.empty(B, H, H, Cout, device=dev); gy = torch.randn(B, H, H, Cout, device=dev)
M = B * H * H
stats = torch.empty((M + 63) // 64, 2, Cout, device=dev)     # (room for 64-row tiles: kernel experiments)
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(max(_lib.fn('dsnt_conv_wgrad_ws_floats')(C.byref(g)), _lib.fn('dsnt_conv_wgrad_f16x3_ws_floats')(C.byref(g), 0)), device=dev); dw = torch.empty_like(w); db = torch.empty(Cout, device=dev)
planes = torch.empty(3 * w.numel(), dtype=torch.bfloat16, device=dev)
_lib.fn('dsnt_split_bf16x3')(ptr(w), ptr(planes), w.numel(), st)
planes16 = torch.empty(2 * w.numel(), dtype=torch.float16, device=dev)
wb, ab, gb = torch.zeros(64, device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev)
_lib.fn('dsnt_amax')(ptr(w), w.numel(), ptr(wb), st)
_lib.fn('dsnt_split_f16x2')(ptr(w), ptr(planes16), w.numel(), w.numel(), ptr(wb), st)
_lib.fn('dsnt_amax')(ptr(gy), gy.numel(), ptr(gb), st)
ab.fill_(float(torch.relu(x * sc + sh).max()) * 4.0)
planes16s = torch.empty(2 * w.numel(), dtype=torch.float16, device=dev)
if k == 3 and _lib.fn('dsnt_conv_fwd_stream_ok')(C.byref(g)):
    tab = torch.tensor([[w.data_ptr(), planes16s.data_ptr(), wb.data_ptr(), w.numel(), w.numel(), Cout, Cin]], dtype=torch.int64).to(dev)
    _lib.fn('dsnt_f16_prep_weights')(ptr(tab), 1, 7, st)
if which.startswith('bwd1'):
    # the one-pass 1x1 backward (csrc/bwd1.hip): x [M][Cin] seen through BN + ReLU, dY [M][Cout]; 'bwd1a' folds the BatchNorm
    # backward of the layer behind (dY from dz and y), 'bwd1' takes dY as given
    from dsnt._lib import BnBwdEpilogue, BnBwdApply
    wd = torch.randn(Cin, Cout, device=dev) * 0.05
    wbd = torch.zeros(64, device=dev)
    _lib.fn('dsnt_amax')(ptr(wd), wd.numel(), ptr(wbd), st)
    pld = torch.empty(2 * wd.numel(), dtype=torch.float16, device=dev)
    _lib.fn('dsnt_split_f16x2')(ptr(wd), ptr(pld), wd.numel(), wd.numel(), ptr(wbd), st)
    mu, isd = torch.randn(Cin, device=dev) * 0.1, torch.rand(Cin, device=dev) + 0.5
    ysc, ymu, yis = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1, torch.rand(Cout, device=dev) + 0.5
    coef = torch.randn(2, Cout, device=dev) * 1e-3
    gbd = torch.full((64,), float(gy.abs().max()) * 16.0, device=dev)
    xs1 = BnBwdEpilogue(ptr(x), ptr(sc), ptr(sh), ptr(mu), ptr(isd), 1)
    ap1 = BnBwdApply(ptr(y), ptr(ysc), ptr(ymu), ptr(yis), ptr(coef))
    nsp1 = _lib.fn('dsnt_conv1x1_bwd_splits')(C.byref(g), 0)
    ws1 = torch.empty(_lib.fn('dsnt_conv1x1_bwd_ws_floats')(C.byref(g), 0), device=dev)
    part1, dzx1 = torch.empty(nsp1, 2, Cin, device=dev), torch.empty(B, H, H, Cin, device=dev)
if which in ('dgrad16s', 'fold3'):
    # the 3x3 data gradient on the persistent kernel with the BatchNorm-backward epilogue (conv3s MODE 3), and with the backward of
    # the BatchNorm BEHIND the convolution folded into its operand load (MODE 4, dsnt_conv_dgrad_f16x3_stream_apply)
    from dsnt._lib import BnBwdEpilogue, BnBwdApply, BnTail
    mu3, is3 = torch.randn(Cout, device=dev) * 0.1, torch.rand(Cout, device=dev) + 0.5
    bnb3 = BnBwdEpilogue(ptr(x), ptr(sc), ptr(sh), ptr(mu3), ptr(is3), 1)
    coef3 = torch.randn(2, Cin, device=dev) * 1e-3
    ap3 = BnBwdApply(ptr(y), ptr(sc), ptr(mu3), ptr(is3), ptr(coef3))
    y.normal_()
    dyo3, out3 = torch.empty(B, H, H, Cin, device=dev), torch.empty(B, H, H, Cout, device=dev)
    stats3 = torch.empty(M // 128, 2, Cout, device=dev)
    gb3 = torch.full((64,), float(gy.abs().max()) * 4.0, device=dev)
    amax3 = torch.zeros(64, device=dev)
    tail3 = BnTail()
    tail3.amax = amax3.data_ptr()
reps = int(os.environ.get('ONE_KERNEL_REPS', '5'))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
for it in range(reps):
    ev[it].record()
    if which == 'fwd6':
        _lib.fn('dsnt_conv_fwd_bf16x6')(ptr(x), ptr(planes), w.numel(), ptr(b), ptr(y), ptr(sc), ptr(sh), 1, None, None, ptr(stats), C.byref(g), st)
    elif which == 'fwd16':
        _lib.fn('dsnt_conv_fwd_f16x3_ex')(ptr(x), ptr(planes16), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y), ptr(sc), ptr(sh), 1, None, None, ptr(stats), C.byref(g), None, None, st)
    elif which == 'fwd16s':
        _lib.fn('dsnt_conv_fwd_f16x3_stream')(ptr(x), ptr(planes16s), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y), ptr(sc), ptr(sh), 1, None, None, ptr(stats), C.byref(g), None, None, st)
    elif which == 'dgrad16s':
        assert _lib.fn('dsnt_conv_fwd_f16x3_stream')(ptr(gy), ptr(planes16s), w.numel(), ptr(wb), ptr(gb3), None, ptr(out3), None, None, 0, None, None,
                                                     ptr(stats3), C.byref(g), C.byref(bnb3), C.byref(tail3), st) == 0
    elif which == 'fold3':
        assert _lib.fn('dsnt_conv_dgrad_f16x3_stream_apply')(ptr(gy), C.byref(ap3), ptr(dyo3), ptr(planes16s), w.numel(), ptr(wb), ptr(gb3), ptr(out3),
                                                             ptr(stats3), 0, C.byref(g), C.byref(bnb3), C.byref(tail3), st) == 0
    elif which == 'wgrad16':
        _lib.fn('dsnt_conv_wgrad_f16x3')(ptr(x), ptr(sc), ptr(sh), 1, ptr(gy), ptr(ws), None, None, 0, ptr(ab), ptr(gb), C.byref(g), st)
    elif which == 'wgrad6':
        _lib.fn('dsnt_conv_wgrad_bf16x6')(ptr(x), ptr(sc), ptr(sh), 1, ptr(gy), ptr(ws), None, None, 0, C.byref(g), st)
    elif which.startswith('bwd1'):
        assert _lib.fn('dsnt_conv1x1_bwd_f16x3')(C.byref(xs1), ptr(gy), C.byref(ap1) if which == 'bwd1a' else None, ptr(pld), wd.numel(),
                                                 ptr(wbd), ptr(ab), ptr(gbd), ptr(dzx1), ptr(part1), ptr(ws1), None, 0, C.byref(g), st) == 0
    elif which == 'fwd1':
        assert _lib.fn('dsnt_conv1x1_fwd_f16x3')(ptr(x), ptr(planes16), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y), ptr(sc), ptr(sh), 1,
                                                 ptr(gy) if Cout > Cin else None, ptr(stats), C.byref(g), None, st) == 0
    elif which == 'fwd':
        _lib.fn('dsnt_conv_fwd')(ptr(x), ptr(w), ptr(b), ptr(y), ptr(sc), ptr(sh), 1, None, None, ptr(stats), C.byref(g), st)
    else:
        _lib.fn('dsnt_conv_wgrad')(ptr(x), ptr(sc), ptr(sh), 1, ptr(gy), ptr(ws), ptr(dw), ptr(db), 0, C.byref(g), st)
ev[reps].record()
torch.cuda.synchronize()
print(which, H, Cin, Cout, k, 'us per launch:', ' '.join('%.1f' % (1e3 * ev[i].elapsed_time(ev[i + 1])) for i in range(reps)))
