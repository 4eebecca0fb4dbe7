"""HIP implicit-GEMM convolution / BN / pool kernels vs torch CPU ops (the oracle's building
blocks), called through the C ABI.  fp32 MFMA is an exact-f32 fmaf chain, so forward values are
held to 2e-5 relative to the output scale (accumulation-order differences only)."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from dsnt import synthetic

pytestmark = pytest.mark.gpu


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _geom(N, H, W, Cin, Cout, R, S, stride, pad, dil):
    from dsnt._lib import ConvGeom
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (S - 1) - 1) // stride + 1
    return ConvGeom(N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil)


CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil
    (2, 16, 16, 128, 128, 3, 1, 1, 1),
    (2, 16, 16, 256, 128, 1, 1, 0, 1),
    (2, 16, 16, 128, 256, 1, 1, 0, 1),
    (3, 8, 8, 256, 16, 1, 1, 0, 1),      # score
    (3, 8, 8, 16, 256, 1, 1, 0, 1),      # score_
    (2, 32, 32, 4, 64, 7, 2, 3, 1),      # stem (Cin padded 3 -> 4)
    (2, 12, 12, 64, 64, 3, 1, 1, 1),
    (1, 5, 7, 64, 128, 3, 1, 1, 1),      # ragged M (35 rows)
    (2, 16, 16, 64, 128, 3, 2, 1, 1),    # stride 2 (ResNet)
    (2, 16, 16, 64, 64, 3, 1, 2, 2),     # dilated (ResNet dilate)
    (4, 64, 64, 128, 128, 3, 1, 1, 1),   # multi-tile, XCD remap
]


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('pro', [False, True])
def test_conv_fwd_wgrad_dgrad(case, pro):
    from dsnt import _lib
    from dsnt._lib import ptr, call
    N, H, W, Cin, Cout, k, stride, pad, dil = case
    dev = torch.device('cuda:0')
    tag = 'c' + '_'.join(map(str, case))
    x = synthetic.tensor(tag + 'x', (N, Cin, H, W), seed=1)
    w = synthetic.tensor(tag + 'w', (Cout, Cin, k, k), seed=1, scale=(2.0 / (Cin * k * k)) ** 0.5)
    b = synthetic.tensor(tag + 'b', (Cout,), seed=1, scale=0.1)
    sc = synthetic.tensor(tag + 's', (Cin,), seed=1, kind='uniform').abs() + 0.5
    sh = synthetic.tensor(tag + 'h', (Cin,), seed=1, scale=0.3)
    g = _geom(N, H, W, Cin, Cout, k, k, stride, pad, dil)
    act = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) if pro else x
    act = act.clone().requires_grad_()
    wr = w.clone().requires_grad_()
    br = b.clone().requires_grad_()
    y_ref = F.conv2d(act, wr, br, stride=stride, padding=pad, dilation=dil)
    res = synthetic.tensor(tag + 'r', tuple(y_ref.shape), seed=1)
    y_ref2 = y_ref + res
    gy = synthetic.tensor(tag + 'g', tuple(y_ref.shape), seed=2)
    y_ref2.backward(gy)

    xd = _nhwc(x).to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)       # OHWI
    bd, scd, shd = b.to(dev), sc.to(dev), sh.to(dev)
    resd = _nhwc(res).to(dev)
    y = torch.empty(N, g.Ho, g.Wo, Cout, device=dev)
    M = N * g.Ho * g.Wo
    bm = _lib.fn('dsnt_conv_fwd_bm')(C.byref(g))
    tiles = (M + bm - 1) // bm
    stats = torch.zeros(tiles, 2, Cout, device=dev)
    call('dsnt_conv_fwd', ptr(xd), ptr(wd), ptr(bd), ptr(y), ptr(scd) if pro else None,
         ptr(shd) if pro else None, 1, ptr(resd), None, ptr(stats), C.byref(g))
    got = y.cpu().permute(0, 3, 1, 2)
    scale = y_ref2.abs().max().item()
    assert (got - y_ref2.detach()).abs().max().item() <= 2e-5 * scale
    # fused statistics epilogue == column sums of the output
    yd = y_ref2.detach().permute(0, 2, 3, 1).reshape(M, Cout).double()
    s = stats.cpu().double().sum(0)
    assert (s[0] - yd.sum(0)).abs().max().item() <= 1e-4 * max(1.0, yd.sum(0).abs().max().item())
    assert (s[1] - (yd * yd).sum(0)).abs().max().item() <= 1e-4 * (yd * yd).sum(0).max().item()

    # weight / bias gradient
    gyd = _nhwc(gy).to(dev)
    ws = torch.empty(_lib.fn('dsnt_conv_wgrad_ws_floats')(C.byref(g)), device=dev)
    dw = torch.empty_like(wd)
    db = torch.empty(Cout, device=dev)
    call('dsnt_conv_wgrad', ptr(xd), ptr(scd) if pro else None, ptr(shd) if pro else None, 1,
         ptr(gyd), ptr(ws), ptr(dw), ptr(db), 0, C.byref(g))
    dw_ref = wr.grad.permute(0, 2, 3, 1)
    assert (dw.cpu() - dw_ref).abs().max().item() <= 3e-5 * max(1.0, dw_ref.abs().max().item())
    assert (db.cpu() - br.grad).abs().max().item() <= 3e-5 * max(1.0, br.grad.abs().max().item())
    call('dsnt_conv_wgrad', ptr(xd), ptr(scd) if pro else None, ptr(shd) if pro else None, 1,
         ptr(gyd), ptr(ws), ptr(dw), ptr(db), 1, C.byref(g))
    assert (dw.cpu() - 2 * dw_ref).abs().max().item() <= 6e-5 * max(1.0, dw_ref.abs().max().item())

    # data gradient (stride-1 convs): forward kernel on dy with the re-packed weights
    if stride == 1:
        wdg = torch.empty(Cin, k, k, Cout, device=dev)
        call('dsnt_conv_pack_dgrad', ptr(wd), ptr(wdg), Cout, k, k, Cin)
        gd = _geom(N, g.Ho, g.Wo, Cout, Cin, k, k, 1, dil * (k - 1) - pad, dil)
        assert gd.Ho == H and gd.Wo == W
        if Cout % 4 == 0:
            da = torch.empty(N, H, W, Cin, device=dev)
            call('dsnt_conv_fwd', ptr(gyd), ptr(wdg), None, ptr(da), None, None, 0, None, None,
                 None, C.byref(gd))
            da_ref = act.grad.permute(0, 2, 3, 1)
            assert (da.cpu() - da_ref).abs().max().item() <= 2e-5 * da_ref.abs().max().item()
            # in-place accumulate through res1 == y
            call('dsnt_conv_fwd', ptr(gyd), ptr(wdg), None, ptr(da), None, None, 0, ptr(da), None,
                 None, C.byref(gd))
            assert (da.cpu() - 2 * da_ref).abs().max().item() <= 4e-5 * da_ref.abs().max().item()


def test_conv_errors():
    from dsnt import _lib
    from dsnt._lib import ptr, call
    dev = torch.device('cuda:0')
    g = _geom(1, 8, 8, 6, 8, 1, 1, 1, 0, 1)   # Cin not a multiple of 4
    t = torch.zeros(1024, device=dev)
    with pytest.raises(RuntimeError, match='multiple of 4'):
        call('dsnt_conv_fwd', ptr(t), ptr(t), None, ptr(t), None, None, 0, None, None, None, C.byref(g))
    g = _geom(1, 8, 8, 8, 8, 3, 3, 1, 1, 1)
    g.Ho = 7
    with pytest.raises(RuntimeError, match='inconsistent'):
        call('dsnt_conv_fwd', ptr(t), ptr(t), None, ptr(t), None, None, 0, None, None, None, C.byref(g))


@pytest.mark.parametrize('M,Cc', [(2 * 16 * 16, 256), (3 * 5 * 7, 64), (2 * 64 * 64, 128), (37, 16)])
def test_batchnorm_pieces(M, Cc):
    """stats -> finalize -> act fwd, and reduce -> finalize -> apply bwd vs F.batch_norm+relu."""
    from dsnt._lib import ptr, call
    dev = torch.device('cuda:0')
    x = synthetic.tensor('bnx%d_%d' % (M, Cc), (M, Cc), seed=3) * 1.7 + 0.4
    gamma = synthetic.tensor('bng', (Cc,), seed=3, kind='uniform') + 1.5
    beta = synthetic.tensor('bnb', (Cc,), seed=3, scale=0.2)
    gy = synthetic.tensor('bngy%d_%d' % (M, Cc), (M, Cc), seed=4)
    xr = x.clone().requires_grad_()
    gr, br = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    rm, rv = torch.zeros(Cc), torch.ones(Cc)
    y_ref = F.relu(F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5))
    y_ref.backward(gy)

    xd, gd, bd = x.to(dev), gamma.to(dev), beta.to(dev)
    tiles = (M + 127) // 128
    part = torch.empty(tiles, 2, Cc, device=dev)
    call('dsnt_bn_stats', ptr(xd), ptr(part), M, Cc)
    mean, invstd, scale, shift = (torch.empty(Cc, device=dev) for _ in range(4))
    rmd, rvd = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    call('dsnt_bn_finalize', ptr(part), tiles, M, Cc, ptr(gd), ptr(bd), ptr(rmd), ptr(rvd), 0.1, 1e-5,
         1, ptr(mean), ptr(invstd), ptr(scale), ptr(shift))
    y = torch.empty(M, Cc, device=dev)
    call('dsnt_bn_act_fwd', ptr(xd), ptr(scale), ptr(shift), 1, ptr(y), M, Cc)
    assert (y.cpu() - y_ref.detach()).abs().max().item() <= 1e-5 * max(1, y_ref.abs().max().item())
    assert (rmd.cpu() - rm).abs().max().item() <= 1e-6 and (rvd.cpu() - rv).abs().max().item() <= 1e-5
    gyd = gy.to(dev)
    call('dsnt_bn_act_bwd_reduce', ptr(gyd), ptr(xd), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), 1,
         ptr(part), M, Cc)
    dgamma, dbeta = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    coef = torch.empty(2, Cc, device=dev)
    call('dsnt_bn_bwd_finalize', ptr(part), tiles, M, Cc, ptr(dgamma), ptr(dbeta), 0, ptr(coef))
    dx = torch.empty(M, Cc, device=dev)
    call('dsnt_bn_act_bwd_apply', ptr(gyd), ptr(xd), ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
         ptr(coef), 1, ptr(dx), 0, M, Cc)
    tol = 2e-5
    assert (dx.cpu() - xr.grad).abs().max().item() <= tol * max(1, xr.grad.abs().max().item())
    assert (dgamma.cpu() - gr.grad).abs().max().item() <= tol * max(1, gr.grad.abs().max().item())
    assert (dbeta.cpu() - br.grad).abs().max().item() <= tol * max(1, br.grad.abs().max().item())
    # eval mode: scale/shift from the running statistics
    call('dsnt_bn_finalize', None, 0, M, Cc, ptr(gd), ptr(bd), ptr(rmd), ptr(rvd), 0.1, 1e-5, 0,
         ptr(mean), ptr(invstd), ptr(scale), ptr(shift))
    call('dsnt_bn_act_fwd', ptr(xd), ptr(scale), ptr(shift), 0, ptr(y), M, Cc)
    y_eval = F.batch_norm(x, rm, rv, gamma, beta, False, 0.1, 1e-5)
    assert (y.cpu() - y_eval).abs().max().item() <= 1e-5 * max(1, y_eval.abs().max().item())


@pytest.mark.parametrize('N,H,W,Cc', [(2, 8, 8, 256), (3, 4, 6, 64), (1, 64, 64, 128)])
def test_pool_upsample(N, H, W, Cc):
    from dsnt._lib import ptr, call
    from dsnt._lib import fn as _lib_fn
    dev = torch.device('cuda:0')
    x = synthetic.tensor('px', (N, Cc, H, W), seed=5).requires_grad_()
    y_ref = F.max_pool2d(x, 2, stride=2)
    gy = synthetic.tensor('pg', tuple(y_ref.shape), seed=5)
    y_ref.backward(gy)
    xd = _nhwc(x.detach()).to(dev)
    y = torch.empty(N, H // 2, W // 2, Cc, device=dev)
    idx = torch.empty(N, H // 2, W // 2, Cc, dtype=torch.uint8, device=dev)
    call('dsnt_maxpool2_fwd', ptr(xd), ptr(y), ptr(idx), N, H, W, Cc)
    assert torch.equal(y.cpu().permute(0, 3, 1, 2), y_ref.detach())
    dx = torch.empty_like(xd)
    gyd = _nhwc(gy).to(dev)   # keep device temporaries alive until the kernels have run
    call('dsnt_maxpool2_bwd', ptr(gyd), ptr(idx), ptr(dx), 0, N, H, W, Cc)
    assert torch.equal(dx.cpu().permute(0, 3, 1, 2), x.grad)
    call('dsnt_maxpool2_bwd', ptr(gyd), ptr(idx), ptr(dx), 1, N, H, W, Cc)
    assert torch.equal(dx.cpu().permute(0, 3, 1, 2), 2 * x.grad)
    # ... with a second gradient of x added in the same pass (dsnt_maxpool2_bwd_add): dx (+)= extra + routed dy, and max |dx| out
    extra = _nhwc(synthetic.tensor('pe', (N, Cc, H, W), seed=7)).to(dev)
    base = dx.clone()
    am = torch.zeros(64, device=dev)
    call('dsnt_maxpool2_bwd_add', ptr(gyd), ptr(idx), ptr(dx), 1, ptr(extra), N, H, W, Cc, ptr(am))
    want = (base + extra) + _nhwc(x.grad).to(dev)
    assert (dx - want).abs().max().item() <= 1e-6 * max(1.0, want.abs().max().item())
    assert am.max().item() == dx.abs().max().item()
    call('dsnt_maxpool2_bwd_add', ptr(gyd), ptr(idx), ptr(dx), 0, ptr(extra), N, H, W, Cc, None)
    assert torch.equal(dx, extra + _nhwc(x.grad).to(dev))
    assert _lib_fn('dsnt_maxpool2_bwd_add')(ptr(gyd), ptr(idx), ptr(dx), 1, ptr(dx), N, H, W, Cc, None, None) != 0     # extra must be a second tensor

    up = synthetic.tensor('uu', (N, Cc, H, W), seed=6)
    low = synthetic.tensor('ul', (N, Cc, H // 2, W // 2), seed=6).requires_grad_()
    out_ref = up + F.interpolate(low, scale_factor=2, mode='nearest')
    go = synthetic.tensor('ug', (N, Cc, H, W), seed=6)
    out_ref.backward(go)
    out = torch.empty(N, H, W, Cc, device=dev)
    upd, lowd, god = _nhwc(up).to(dev), _nhwc(low.detach()).to(dev), _nhwc(go).to(dev)
    call('dsnt_upsample2_add_fwd', ptr(upd), ptr(lowd), ptr(out), N, H, W, Cc)
    assert torch.equal(out.cpu().permute(0, 3, 1, 2), out_ref.detach())
    dl = torch.empty(N, H // 2, W // 2, Cc, device=dev)
    call('dsnt_upsample2_bwd', ptr(god), ptr(dl), 0, N, H, W, Cc)
    assert (dl.cpu().permute(0, 3, 1, 2) - low.grad).abs().max().item() <= 1e-6

    # the same ops with the BatchNorm statistics of their output fused in: identical tensors, and partial sums
    # bit-identical to a separate dsnt_bn_stats pass over the stored result
    for name, args_plain, res, shape in (
            ('dsnt_maxpool2_fwd_stats', (ptr(xd),), y, (N, H // 2, W // 2, Cc)),
            ('dsnt_upsample2_add_fwd_stats', (ptr(upd), ptr(lowd)), out, (N, H, W, Cc))):
        M = shape[0] * shape[1] * shape[2]
        tiles = (M + 127) // 128
        res2 = torch.empty(*shape, device=dev)
        part, want = torch.empty(tiles, 2, Cc, device=dev), torch.empty(tiles, 2, Cc, device=dev)
        if name.startswith('dsnt_maxpool'):
            idx2 = torch.empty_like(idx)
            call(name, *args_plain, ptr(res2), ptr(idx2), ptr(part), N, H, W, Cc, None)
            assert torch.equal(idx2, idx)
        else:
            call(name, *args_plain, ptr(res2), ptr(part), N, H, W, Cc, None)
        call('dsnt_bn_stats', ptr(res), ptr(want), M, Cc)
        assert torch.equal(res2, res)
        assert torch.equal(part, want)

    # the stem's materialised BatchNorm + ReLU with the statistics of its output (and the output's bound) in the same pass:
    # identical to dsnt_bn_act_fwd followed by dsnt_bn_stats
    from dsnt._lib import BnTail
    bsc, bsh = torch.randn(Cc, device=dev), torch.randn(Cc, device=dev)
    M = N * H * W
    tiles = (M + 127) // 128
    y1, y2 = torch.empty(N, H, W, Cc, device=dev), torch.empty(N, H, W, Cc, device=dev)
    call('dsnt_bn_act_fwd', ptr(upd), ptr(bsc), ptr(bsh), 1, ptr(y1), M, Cc)
    part, want, slots = torch.empty(tiles, 2, Cc, device=dev), torch.empty(tiles, 2, Cc, device=dev), torch.zeros(64, device=dev)
    t = BnTail()
    t.amax = slots.data_ptr()
    call('dsnt_bn_act_fwd_stats', ptr(upd), ptr(bsc), ptr(bsh), 1, ptr(y2), ptr(part), M, Cc, C.byref(t))
    call('dsnt_bn_stats', ptr(y1), ptr(want), M, Cc)
    assert torch.equal(y1, y2) and torch.equal(part, want) and float(slots.max()) == float(y1.abs().max())
    assert (y1 - torch.relu(torch.addcmul(bsh, upd, bsc))).abs().max().item() <= 1e-5      # (an fma on the device)

    # layout round trip with channel padding
    img = synthetic.tensor('im', (N, 3, H, W), seed=7)
    nhwc = torch.empty(N, H, W, 4, device=dev)
    imgd = img.to(dev)
    call('dsnt_nchw_to_nhwc', ptr(imgd), ptr(nhwc), N, 3, H * W, 4)
    assert torch.equal(nhwc.cpu()[..., :3].permute(0, 3, 1, 2), img) and nhwc[..., 3].abs().max().item() == 0
    back = torch.empty(N, 3, H, W, device=dev)
    call('dsnt_nhwc_to_nchw', ptr(nhwc), ptr(back), N, 3, H * W, 4)
    assert torch.equal(back.cpu(), img)


BF16X6_CASES = [c for c in CASES if c[3] % 16 == 0 and c[4] % 4 == 0] + [
    # 3x3 / stride 1 / pad 1 with H % 8 == 0, W % 16 == 0, Cin % 32 == 0 run on the LDS halo-tile kernel
    (2, 8, 32, 64, 64, 3, 1, 1, 1),      # 64-wide N tile
    (1, 24, 48, 32, 160, 3, 1, 1, 1),    # ragged Cout (second N tile masked), two chunks
    (3, 16, 16, 96, 128, 3, 1, 1, 1),    # six chunks
    (1, 8, 16, 32, 128, 3, 1, 1, 1),     # a single tile: every halo edge is padding
    # 1x1 convolutions of >= 65536 rows run on the streaming kernel (csrc/gemm1.hip) when both fp16x3 bounds exist
    (16, 64, 64, 128, 256, 1, 1, 0, 1),  # expanding conv3 of a Bottleneck: 256 columns per workgroup
    (16, 64, 64, 256, 128, 1, 1, 0, 1),  # reducing conv1: K = 256
    (16, 64, 64, 128, 128, 1, 1, 0, 1),  # 128 columns, two row tiles per wave
    (16, 64, 64, 256, 256, 1, 1, 0, 1),  # lin / fc_: two column chunks
    (4, 128, 128, 64, 64, 1, 1, 0, 1),   # stem Bottleneck: K = 64, four row tiles per wave
    (4, 128, 128, 64, 128, 1, 1, 0, 1),
    (4, 128, 128, 128, 64, 1, 1, 0, 1),  # K = 128 with 64 columns (data gradient of the stem's 64 -> 128 expansion)
]


@pytest.mark.parametrize('case', BF16X6_CASES)
@pytest.mark.parametrize('pro', [False, True])
def test_conv_bf16x6_matches_fp32_accuracy(case, pro):
    """bf16x6 split-precision kernel: same bar as the fp32-MFMA kernel (2e-5 of the output scale vs
    torch fp32 on the CPU) and an fp64 check that its error is of fp32 size."""
    from dsnt import _lib
    from dsnt._lib import ptr, call
    N, H, W, Cin, Cout, k, stride, pad, dil = case
    dev = torch.device('cuda:0')
    tag = 'c' + '_'.join(map(str, case))
    x = synthetic.tensor(tag + 'x', (N, Cin, H, W), seed=1)
    w = synthetic.tensor(tag + 'w', (Cout, Cin, k, k), seed=1, scale=(2.0 / (Cin * k * k)) ** 0.5)
    b = synthetic.tensor(tag + 'b', (Cout,), seed=1, scale=0.1)
    sc = synthetic.tensor(tag + 's', (Cin,), seed=1, kind='uniform').abs() + 0.5
    sh = synthetic.tensor(tag + 'h', (Cin,), seed=1, scale=0.3)
    g = _geom(N, H, W, Cin, Cout, k, k, stride, pad, dil)
    assert _lib.fn('dsnt_conv_bf16x6_ok')(C.byref(g))
    act = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) if pro else x
    y32 = F.conv2d(act, w, b, stride=stride, padding=pad, dilation=dil)
    y64 = F.conv2d(act.double(), w.double(), b.double(), stride=stride, padding=pad, dilation=dil)
    res = synthetic.tensor(tag + 'r', tuple(y32.shape), seed=1)
    xd = _nhwc(x).to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    planes = torch.empty(3 * wd.numel(), dtype=torch.bfloat16, device=dev)
    call('dsnt_split_bf16x3', ptr(wd), ptr(planes), wd.numel())
    # the split is exact: the three planes add up to the fp32 weights
    assert torch.equal(planes.view(3, -1).float().sum(0), wd.reshape(-1))
    bd, scd, shd, resd = b.to(dev), sc.to(dev), sh.to(dev), _nhwc(res).to(dev)
    y = torch.empty(N, g.Ho, g.Wo, Cout, device=dev)
    M = N * g.Ho * g.Wo
    stats = torch.zeros((M + 127) // 128, 2, Cout, device=dev)
    call('dsnt_conv_fwd_bf16x6', ptr(xd), ptr(planes), wd.numel(), ptr(bd), ptr(y), ptr(scd) if pro else None,
         ptr(shd) if pro else None, 1, ptr(resd), None, ptr(stats), C.byref(g))
    got = y.cpu().permute(0, 3, 1, 2)
    scale = y32.abs().max().item()
    assert (got - (y32 + res)).abs().max().item() <= 2e-5 * scale
    err6 = (got.double() - (y64 + res.double())).abs().max().item()
    err32 = ((y32 + res).double() - (y64 + res.double())).abs().max().item()
    assert err6 <= max(4 * err32, 2e-6 * scale), (err6, err32)
    yd = (y32 + res).permute(0, 2, 3, 1).reshape(M, Cout).double()
    s = stats.cpu().double().sum(0)
    assert (s[0] - yd.sum(0)).abs().max().item() <= 1e-4 * max(1.0, yd.sum(0).abs().max().item())


@pytest.mark.parametrize('case', [c for c in CASES if c[3] % 4 == 0 and c[4] % 4 == 0])
@pytest.mark.parametrize('pro', [False, True])
def test_wgrad_bf16x6(case, pro):
    from dsnt import _lib
    from dsnt._lib import ptr, call
    N, H, W, Cin, Cout, k, stride, pad, dil = case
    dev = torch.device('cuda:0')
    tag = 'c' + '_'.join(map(str, case))
    g = _geom(N, H, W, Cin, Cout, k, k, stride, pad, dil)
    if not _lib.fn('dsnt_conv_wgrad_bf16x6_ok')(C.byref(g)):
        pytest.skip('geometry not supported by the bf16x6 weight gradient (Wo % 4 != 0)')
    x = synthetic.tensor(tag + 'x', (N, Cin, H, W), seed=1)
    sc = synthetic.tensor(tag + 's', (Cin,), seed=1, kind='uniform').abs() + 0.5
    sh = synthetic.tensor(tag + 'h', (Cin,), seed=1, scale=0.3)
    act = (F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) if pro else x).double()
    w = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(act, w, b, stride=stride, padding=pad, dilation=dil)
    gy = synthetic.tensor(tag + 'g', tuple(y.shape), seed=2)
    y.backward(gy.double())
    xd, gyd, scd, shd = _nhwc(x).to(dev), _nhwc(gy).to(dev), sc.to(dev), sh.to(dev)
    ws = torch.empty(_lib.fn('dsnt_conv_wgrad_ws_floats')(C.byref(g)), device=dev)
    dw6, dw32 = torch.empty(Cout, k, k, Cin, device=dev), torch.empty(Cout, k, k, Cin, device=dev)
    db6, db32 = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
    args = (ptr(xd), ptr(scd) if pro else None, ptr(shd) if pro else None, 1, ptr(gyd), ptr(ws))
    call('dsnt_conv_wgrad_bf16x6', *args, ptr(dw6), ptr(db6), 0, C.byref(g))
    call('dsnt_conv_wgrad', *args, ptr(dw32), ptr(db32), 0, C.byref(g))
    ref = w.grad.permute(0, 2, 3, 1)
    scale = max(1.0, ref.abs().max().item())
    e6 = (dw6.cpu().double() - ref).abs().max().item()
    e32 = (dw32.cpu().double() - ref).abs().max().item()
    assert e6 <= 3e-5 * scale and e6 <= max(4 * e32, 2e-6 * scale), (e6, e32)
    assert (db6.cpu().double() - b.grad).abs().max().item() <= 3e-5 * max(1.0, b.grad.abs().max().item())


def test_deferred_slab_reduction_is_bit_identical():
    """dsnt_conv_wgrad(dw=NULL) + one dsnt_wgrad_reduce_all over several convolutions == the per-conv reduction."""
    from dsnt import _lib
    from dsnt._lib import ptr, call
    dev = torch.device('cuda:0')
    cases = [(2, 16, 16, 128, 128, 3, 1, 1, 1), (2, 16, 16, 256, 128, 1, 1, 0, 1), (3, 8, 8, 16, 256, 1, 1, 0, 1)]
    rows, keep, want = [], [], []
    for ci, (N, H, W, Cin, Cout, k, stride, pad, dil) in enumerate(cases):
        g = _geom(N, H, W, Cin, Cout, k, k, stride, pad, dil)
        x = synthetic.tensor('dr%dx' % ci, (N, H, W, Cin), seed=3).to(dev)
        gy = synthetic.tensor('dr%dg' % ci, (N, g.Ho, g.Wo, Cout), seed=4).to(dev)
        fn = 'dsnt_conv_wgrad_bf16x6' if ci < 2 else 'dsnt_conv_wgrad'
        nws = _lib.fn('dsnt_conv_wgrad_ws_floats')(C.byref(g))
        ws1, ws2 = torch.empty(nws, device=dev), torch.empty(nws, device=dev)
        dw1, db1 = torch.empty(Cout, k, k, Cin, device=dev), torch.empty(Cout, device=dev)
        dw2, db2 = torch.full((Cout, k, k, Cin), 9.0, device=dev), torch.full((Cout,), 9.0, device=dev)
        call(fn, ptr(x), None, None, 0, ptr(gy), ptr(ws1), ptr(dw1), ptr(db1), 0, C.byref(g))
        call(fn, ptr(x), None, None, 0, ptr(gy), ptr(ws2), None, None, 0, C.byref(g))     # slabs only
        with_bias = ci != 1
        rows.append([ws2.data_ptr(), dw2.data_ptr(), db2.data_ptr() if with_bias else 0,
                     _lib.fn('dsnt_conv_wgrad_splits')(C.byref(g)), Cout * k * k * Cin, Cout, 0])
        keep.append((x, gy, ws1, ws2))
        want.append((dw1, db1, dw2, db2, with_bias))
    table = torch.tensor(rows, dtype=torch.int64).to(dev)
    blocks = max((r[4] // 4 + (r[5] + 3) // 4 + 63) // 64 for r in rows)
    call('dsnt_wgrad_reduce_all', ptr(table), len(rows), blocks)
    for dw1, db1, dw2, db2, with_bias in want:
        assert torch.equal(dw1, dw2)
        assert torch.equal(db1, db2) if with_bias else bool((db2 == 9.0).all())
    assert _lib.fn('dsnt_wgrad_reduce_all')(None, 1, 1, None) != 0
    assert _lib.fn('dsnt_conv_wgrad')(ptr(keep[0][0]), None, None, 0, ptr(keep[0][1]), ptr(keep[0][2]), None,
                                       ptr(want[0][1]), 0, C.byref(_geom(*cases[0][:5], 3, 3, 1, 1, 1)), None) != 0


def test_grouped_wgrad_is_bit_identical():
    """Several small bf16x6 weight gradients in ONE dsnt_conv_wgrad_group launch write the same slabs as the
    per-convolution launches (the 16x16 ... 4x4 hourglass levels)."""
    from dsnt import _lib
    from dsnt._lib import ptr, call
    dev = torch.device('cuda:0')
    cases = [(2, 16, 16, 128, 128, 3, 1, 1, 1), (2, 16, 16, 256, 128, 1, 1, 0, 1), (3, 8, 8, 128, 256, 1, 1, 0, 1),
             (2, 4, 4, 128, 128, 3, 1, 1, 1)]
    nbytes = _lib.fn('dsnt_conv_wgrad_desc_bytes')()
    assert nbytes > 0
    blob, keep, nblk = b'', [], []
    for ci, (N, H, W, Cin, Cout, k, stride, pad, dil) in enumerate(cases):
        g = _geom(N, H, W, Cin, Cout, k, k, stride, pad, dil)
        x = synthetic.tensor('gw%dx' % ci, (N, H, W, Cin), seed=5).to(dev)
        gy = synthetic.tensor('gw%dg' % ci, (N, g.Ho, g.Wo, Cout), seed=6).to(dev)
        sc = (synthetic.tensor('gw%ds' % ci, (Cin,), seed=7).abs() + 0.5).to(dev)
        sh = (synthetic.tensor('gw%dh' % ci, (Cin,), seed=8) * 0.1).to(dev)
        relu = ci % 2
        nws = _lib.fn('dsnt_conv_wgrad_ws_floats')(C.byref(g))
        ws1, ws2 = torch.zeros(nws, device=dev), torch.full((nws,), 7.0, device=dev)
        call('dsnt_conv_wgrad_bf16x6', ptr(x), ptr(sc), ptr(sh), relu, ptr(gy), ptr(ws1), None, None, 0, C.byref(g))
        desc = C.create_string_buffer(nbytes)
        n = _lib.fn('dsnt_conv_wgrad_desc')(ptr(x), ptr(sc), ptr(sh), relu, ptr(gy), ptr(ws2), C.byref(g), desc)
        assert n > 0
        blob += desc.raw
        nblk.append(n)
        keep.append((x, gy, sc, sh, ws1, ws2))
    table = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    call('dsnt_conv_wgrad_group', ptr(table), len(cases), (max(nblk) + 7) // 8 * 8)
    torch.cuda.synchronize()
    for x, gy, sc, sh, ws1, ws2 in keep:
        assert torch.equal(ws1, ws2)
    # errors: no BN vectors, null table
    g = _geom(*cases[0][:5], 3, 3, 1, 1, 1)
    desc = C.create_string_buffer(nbytes)
    assert _lib.fn('dsnt_conv_wgrad_desc')(ptr(keep[0][0]), None, None, 0, ptr(keep[0][1]), ptr(keep[0][5]), C.byref(g), desc) < 0
    assert _lib.fn('dsnt_conv_wgrad_group')(None, 1, 8, None) != 0


@pytest.mark.parametrize('case', BF16X6_CASES)
@pytest.mark.parametrize('pro', [False, True])
@pytest.mark.parametrize('loose', [1.0, 64.0])
def test_conv_f16x3_matches_fp32_accuracy(case, pro, loose):
    """fp16x3 (two fp16 planes after a power-of-two scale, three MFMAs): the same bars as bf16x6 — 2e-5 of the output
    scale vs torch fp32 and an fp64 check that its error is of fp32 size — with the exact operand bound and with a
    bound 64x too large (what the analytic BatchNorm bound typically is)."""
    from dsnt import _lib
    from dsnt._lib import ptr, call
    N, H, W, Cin, Cout, k, stride, pad, dil = case
    dev = torch.device('cuda:0')
    tag = 'c' + '_'.join(map(str, case))
    x = synthetic.tensor(tag + 'x', (N, Cin, H, W), seed=1)
    w = synthetic.tensor(tag + 'w', (Cout, Cin, k, k), seed=1, scale=(2.0 / (Cin * k * k)) ** 0.5)
    b = synthetic.tensor(tag + 'b', (Cout,), seed=1, scale=0.1)
    sc = synthetic.tensor(tag + 's', (Cin,), seed=1, kind='uniform').abs() + 0.5
    sh = synthetic.tensor(tag + 'h', (Cin,), seed=1, scale=0.3)
    g = _geom(N, H, W, Cin, Cout, k, k, stride, pad, dil)
    act = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) if pro else x
    y32 = F.conv2d(act, w, b, stride=stride, padding=pad, dilation=dil)
    y64 = F.conv2d(act.double(), w.double(), b.double(), stride=stride, padding=pad, dilation=dil)
    res = synthetic.tensor(tag + 'r', tuple(y32.shape), seed=1)
    xd = _nhwc(x).to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    wb, ab = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    call('dsnt_amax', ptr(wd), wd.numel(), ptr(wb))
    assert wb.max().item() == wd.abs().max().item()
    planes = torch.empty(2 * wd.numel(), dtype=torch.float16, device=dev)
    call('dsnt_split_f16x2', ptr(wd), ptr(planes), wd.numel(), wd.numel(), ptr(wb))
    # the two planes carry the scaled weights to ~2^-23 of the largest
    import math
    s_w = 2.0 ** (13 - math.floor(math.log2(wb.max().item())))
    back = planes.view(2, -1).double().sum(0) / s_w
    assert (back - wd.reshape(-1).double()).abs().max().item() <= 2.0 ** -22 * wd.abs().max().item()
    ab[7] = act.abs().max().item() * loose            # any slot may hold the maximum
    bd, scd, shd, resd = b.to(dev), sc.to(dev), sh.to(dev), _nhwc(res).to(dev)
    y = torch.empty(N, g.Ho, g.Wo, Cout, device=dev)
    M = N * g.Ho * g.Wo
    stats = torch.zeros((M + 127) // 128, 2, Cout, device=dev)
    call('dsnt_conv_fwd_f16x3_ex', ptr(xd), ptr(planes), wd.numel(), ptr(wb), ptr(ab), ptr(bd), ptr(y),
         ptr(scd) if pro else None, ptr(shd) if pro else None, 1, ptr(resd), None, ptr(stats), C.byref(g), None, None)
    got = y.cpu().permute(0, 3, 1, 2)
    scale = y32.abs().max().item()
    assert (got - (y32 + res)).abs().max().item() <= 2e-5 * scale
    err16 = (got.double() - (y64 + res.double())).abs().max().item()
    err32 = ((y32 + res).double() - (y64 + res.double())).abs().max().item()
    assert err16 <= max(4 * err32, 2e-6 * scale), (err16, err32)
    yd = (y32 + res).permute(0, 2, 3, 1).reshape(M, Cout).double()
    s = stats.cpu().double().sum(0)
    assert (s[0] - yd.sum(0)).abs().max().item() <= 1e-4 * max(1.0, yd.sum(0).abs().max().item())


@pytest.mark.parametrize('case', [c for c in CASES if c[3] % 4 == 0 and c[4] % 4 == 0])
@pytest.mark.parametrize('pro', [False, True])
def test_wgrad_f16x3(case, pro):
    """fp16x3 weight gradient vs fp64, same bars as test_wgrad_bf16x6; gradient-sized dy (1e-4) so that the scale
    matters; the bound of dy is found by dsnt_amax, the one of the A operand is 16x loose."""
    from dsnt import _lib
    from dsnt._lib import ptr, call
    N, H, W, Cin, Cout, k, stride, pad, dil = case
    dev = torch.device('cuda:0')
    tag = 'c' + '_'.join(map(str, case))
    g = _geom(N, H, W, Cin, Cout, k, k, stride, pad, dil)
    if not _lib.fn('dsnt_conv_wgrad_bf16x6_ok')(C.byref(g)):
        pytest.skip('geometry not supported by the split-precision weight gradient (Wo % 4 != 0)')
    x = synthetic.tensor(tag + 'x', (N, Cin, H, W), seed=1)
    sc = synthetic.tensor(tag + 's', (Cin,), seed=1, kind='uniform').abs() + 0.5
    sh = synthetic.tensor(tag + 'h', (Cin,), seed=1, scale=0.3)
    act = (F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) if pro else x).double()
    w = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(act, w, b, stride=stride, padding=pad, dilation=dil)
    gy = synthetic.tensor(tag + 'g', tuple(y.shape), seed=2) * 1e-4
    y.backward(gy.double())
    xd, gyd, scd, shd = _nhwc(x).to(dev), _nhwc(gy).to(dev), sc.to(dev), sh.to(dev)
    ab, gb = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    ab.fill_(act.abs().max().item() * 16.0)
    call('dsnt_amax', ptr(gyd), gyd.numel(), ptr(gb))
    ws = torch.empty(_lib.fn('dsnt_conv_wgrad_ws_floats')(C.byref(g)), device=dev)
    dw16, dw32 = torch.empty(Cout, k, k, Cin, device=dev), torch.empty(Cout, k, k, Cin, device=dev)
    db16, db32 = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
    args = (ptr(xd), ptr(scd) if pro else None, ptr(shd) if pro else None, 1, ptr(gyd), ptr(ws))
    call('dsnt_conv_wgrad_f16x3', *args, ptr(dw16), ptr(db16), 0, ptr(ab), ptr(gb), C.byref(g))
    call('dsnt_conv_wgrad', *args, ptr(dw32), ptr(db32), 0, C.byref(g))
    ref = w.grad.permute(0, 2, 3, 1)
    scale = ref.abs().max().item()
    e16 = (dw16.cpu().double() - ref).abs().max().item()
    e32 = (dw32.cpu().double() - ref).abs().max().item()
    assert e16 <= 3e-5 * scale and e16 <= max(4 * e32, 2e-6 * scale), (e16, e32)
    assert (db16.cpu().double() - b.grad).abs().max().item() <= 3e-5 * b.grad.abs().max().item()


HALO_CASES = [
    # N, H, W, Cin, Cout: 3x3 / stride 1 / pad 1 shapes of the halo weight gradient (csrc/wgrad3.hip)
    (2, 64, 64, 128, 128),     # the dominant shape: 64 channels x 9 taps x 128 columns per workgroup, W = 64
    (1, 32, 32, 128, 128),     # W = 32
    (2, 16, 16, 128, 256),     # W = 16, two 128-column chunks
    (2, 32, 32, 256, 128),     # four 64-channel chunks
    (2, 16, 16, 64, 64),       # 64 output channels: taps split over two wave groups
    (1, 128, 128, 64, 64),     # the stem Bottleneck: two 64-pixel strips with real halo columns
    (1, 128, 64, 64, 128),     # H != W
    (3, 8, 16, 64, 128),       # H = 8: a slab is a whole image
]


@pytest.mark.parametrize('case', HALO_CASES)
@pytest.mark.parametrize('pro,relu', [(True, 1), (True, 0), (False, 0)])
def test_wgrad_halo_f16x3(case, pro, relu):
    """The halo weight gradient (every 3x3 Bottleneck conv2, hourglass.py:22-23) vs fp64 autograd, same bars as
    test_wgrad_f16x3, and its slab plan: dsnt_conv_wgrad_f16x3_splits slabs reduced by dsnt_wgrad_reduce_all equal
    the reduction inside the call bit for bit."""
    from dsnt import _lib
    from dsnt._lib import ptr, call
    N, H, W, Cin, Cout = case
    dev = torch.device('cuda:0')
    tag = 'h' + '_'.join(map(str, case))
    g = _geom(N, H, W, Cin, Cout, 3, 3, 1, 1, 1)
    assert _lib.fn('dsnt_conv_wgrad_halo_ok')(C.byref(g)) == 1
    x = synthetic.tensor(tag + 'x', (N, Cin, H, W), seed=1)
    sc = synthetic.tensor(tag + 's', (Cin,), seed=1, kind='uniform').abs() + 0.5
    sh = synthetic.tensor(tag + 'h', (Cin,), seed=1, scale=0.3)
    act = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) if pro else x
    act = (F.relu(act) if relu else act).double()
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(act, w, b, padding=1)
    gy = synthetic.tensor(tag + 'g', tuple(y.shape), seed=2) * 1e-4
    y.backward(gy.double())
    xd, gyd, scd, shd = _nhwc(x).to(dev), _nhwc(gy).to(dev), sc.to(dev), sh.to(dev)
    ab, gb = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    ab.fill_(act.abs().max().item() * 16.0)
    call('dsnt_amax', ptr(gyd), gyd.numel(), ptr(gb))
    nws = _lib.fn('dsnt_conv_wgrad_f16x3_ws_floats')(C.byref(g), 0)
    splits = _lib.fn('dsnt_conv_wgrad_f16x3_splits')(C.byref(g), 0)
    assert nws == splits * Cout * (9 * Cin + 1)
    ws = torch.full((nws,), float('nan'), device=dev)
    dw16, dw32 = torch.empty(Cout, 3, 3, Cin, device=dev), torch.empty(Cout, 3, 3, Cin, device=dev)
    db16, db32 = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
    args = (ptr(xd), ptr(scd) if pro else None, ptr(shd) if pro else None, relu, ptr(gyd))
    call('dsnt_conv_wgrad_f16x3', *args, ptr(ws), ptr(dw16), ptr(db16), 0, ptr(ab), ptr(gb), C.byref(g))
    assert bool(torch.isfinite(ws).all()), 'a slab element was never written'
    ws32 = torch.empty(_lib.fn('dsnt_conv_wgrad_ws_floats')(C.byref(g)), device=dev)
    call('dsnt_conv_wgrad', *args, ptr(ws32), ptr(dw32), ptr(db32), 0, C.byref(g))
    ref = w.grad.permute(0, 2, 3, 1)
    scale = ref.abs().max().item()
    e16 = (dw16.cpu().double() - ref).abs().max().item()
    e32 = (dw32.cpu().double() - ref).abs().max().item()
    assert e16 <= 3e-5 * scale and e16 <= max(4 * e32, 2e-6 * scale), (e16, e32)
    assert (db16.cpu().double() - b.grad).abs().max().item() <= 3e-5 * b.grad.abs().max().item()
    # slabs only + the table-driven reduction: the same bits; accumulate adds
    ws2 = torch.empty(nws, device=dev)
    dw2, db2 = torch.zeros_like(dw16), torch.zeros_like(db16)
    call('dsnt_conv_wgrad_f16x3', *args, ptr(ws2), None, None, 0, ptr(ab), ptr(gb), C.byref(g))
    assert torch.equal(ws, ws2)
    # DSNT_WGRAD_SHARE_CHIP (bit 1): four-wave workgroups of 32 input channels that leave half of every CU to the
    # other streams, over its own plan of (at most as many) slabs: every slab element written, the same gradient
    nws_s = _lib.fn('dsnt_conv_wgrad_f16x3_ws_floats')(C.byref(g), 2)
    splits_s = _lib.fn('dsnt_conv_wgrad_f16x3_splits')(C.byref(g), 2)
    assert nws_s == splits_s * Cout * (9 * Cin + 1) and splits_s <= splits
    ws3 = torch.full((nws_s,), float('nan'), device=dev)
    dw3, db3 = torch.empty_like(dw16), torch.empty_like(db16)
    call('dsnt_conv_wgrad_f16x3', *args, ptr(ws3), ptr(dw3), ptr(db3), 2, ptr(ab), ptr(gb), C.byref(g))
    assert bool(torch.isfinite(ws3).all())
    e3 = (dw3.cpu().double() - ref).abs().max().item()
    assert e3 <= 3e-5 * scale and e3 <= max(4 * e32, 2e-6 * scale), (e3, e32)
    assert (db3.cpu().double() - b.grad).abs().max().item() <= 3e-5 * b.grad.abs().max().item()
    if splits_s == splits:
        assert torch.equal(ws, ws3)      # same slabs: the two workgroup shapes add in the same order, bit for bit
    table = torch.tensor([[ws2.data_ptr(), dw2.data_ptr(), db2.data_ptr(), splits, Cout * 9 * Cin, Cout, 0]],
                         dtype=torch.int64).to(dev)
    call('dsnt_wgrad_reduce_all', ptr(table), 1, (Cout * 9 * Cin // 4 + (Cout + 3) // 4 + 63) // 64)
    assert torch.equal(dw2, dw16) and torch.equal(db2, db16)
    call('dsnt_conv_wgrad_f16x3', *args, ptr(ws), ptr(dw16), ptr(db16), 1, ptr(ab), ptr(gb), C.byref(g))
    assert (dw16.cpu().double() - 2 * ref).abs().max().item() <= 6e-5 * scale


W1_CASES = [
    # N, H, W, Cin, Cout: 1x1 / stride 1 shapes of >= 16384 rows: the transposition-free weight gradient (csrc/wgrad1.hip)
    (4, 64, 64, 128, 256),     # expanding conv3: the whole 128 x 256 matrix per workgroup
    (4, 64, 64, 256, 128),     # reducing conv1: 256 x 128
    (4, 64, 64, 256, 256),     # lin: two input-channel chunks
    (4, 64, 64, 128, 128),
    (2, 128, 128, 64, 64),     # stem Bottleneck
    (2, 128, 128, 64, 128),
    (2, 128, 128, 128, 64),
    (5, 64, 64, 128, 256),     # 20480 rows: the last slab is shorter
]


@pytest.mark.parametrize('case', W1_CASES)
@pytest.mark.parametrize('pro,relu', [(True, 1), (False, 0)])
def test_wgrad_1x1_f16x3(case, pro, relu):
    """The four-wave 1x1 weight gradient (conv1 / conv3 of every Bottleneck, hourglass.py:20-25) vs fp64 autograd, the
    bars of test_wgrad_f16x3, every slab element written, slabs-only + table-driven reduction bit-identical to the
    reduction inside the call, and the DSNT_WGRAD_SHARE_CHIP launch (same kernel) bit-identical too."""
    from dsnt import _lib
    from dsnt._lib import ptr, call
    N, H, W, Cin, Cout = case
    dev = torch.device('cuda:0')
    tag = 'w1' + '_'.join(map(str, case))
    g = _geom(N, H, W, Cin, Cout, 1, 1, 1, 0, 1)
    x = synthetic.tensor(tag + 'x', (N, Cin, H, W), seed=1)
    sc = synthetic.tensor(tag + 's', (Cin,), seed=1, kind='uniform').abs() + 0.5
    sh = synthetic.tensor(tag + 'h', (Cin,), seed=1, scale=0.3)
    act = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) if pro else x
    act = (F.relu(act) if relu else act).double()
    gy = synthetic.tensor(tag + 'g', (N, Cout, H, W), seed=2) * 1e-4
    a2 = act.permute(0, 2, 3, 1).reshape(-1, Cin)
    g2 = gy.double().permute(0, 2, 3, 1).reshape(-1, Cout)
    ref = (g2.t() @ a2).view(Cout, 1, 1, Cin)
    bref = g2.sum(0)
    xd, gyd, scd, shd = _nhwc(x).to(dev), _nhwc(gy).to(dev), sc.to(dev), sh.to(dev)
    ab, gb = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    ab.fill_(act.abs().max().item() * 16.0)
    call('dsnt_amax', ptr(gyd), gyd.numel(), ptr(gb))
    nws = _lib.fn('dsnt_conv_wgrad_f16x3_ws_floats')(C.byref(g), 0)
    splits = _lib.fn('dsnt_conv_wgrad_f16x3_splits')(C.byref(g), 0)
    assert nws == splits * Cout * (Cin + 1)
    ws = torch.full((nws,), float('nan'), device=dev)
    dw16, dw32 = torch.empty(Cout, 1, 1, Cin, device=dev), torch.empty(Cout, 1, 1, Cin, device=dev)
    db16, db32 = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
    args = (ptr(xd), ptr(scd) if pro else None, ptr(shd) if pro else None, relu, ptr(gyd))
    call('dsnt_conv_wgrad_f16x3', *args, ptr(ws), ptr(dw16), ptr(db16), 0, ptr(ab), ptr(gb), C.byref(g))
    assert bool(torch.isfinite(ws).all()), 'a slab element was never written'
    ws32 = torch.empty(_lib.fn('dsnt_conv_wgrad_ws_floats')(C.byref(g)), device=dev)
    call('dsnt_conv_wgrad', *args, ptr(ws32), ptr(dw32), ptr(db32), 0, C.byref(g))
    scale = ref.abs().max().item()
    e16 = (dw16.cpu().double() - ref).abs().max().item()
    e32 = (dw32.cpu().double() - ref).abs().max().item()
    assert e16 <= 3e-5 * scale and e16 <= max(4 * e32, 2e-6 * scale), (e16, e32)
    assert (db16.cpu().double() - bref).abs().max().item() <= 3e-5 * bref.abs().max().item()
    ws2 = torch.empty(nws, device=dev)
    dw2, db2 = torch.zeros_like(dw16), torch.zeros_like(db16)
    call('dsnt_conv_wgrad_f16x3', *args, ptr(ws2), None, None, 0, ptr(ab), ptr(gb), C.byref(g))      # slabs only
    assert torch.equal(ws, ws2)
    table = torch.tensor([[ws2.data_ptr(), dw2.data_ptr(), db2.data_ptr(), splits, Cout * Cin, Cout, 0]],
                         dtype=torch.int64).to(dev)
    call('dsnt_wgrad_reduce_all', ptr(table), 1, (Cout * Cin // 4 + (Cout + 3) // 4 + 63) // 64)
    assert torch.equal(dw2, dw16) and torch.equal(db2, db16)
    # DSNT_WGRAD_SHARE_CHIP: half as many (longer) slabs on half the CUs — its own plan, the same gradient
    nws_s = _lib.fn('dsnt_conv_wgrad_f16x3_ws_floats')(C.byref(g), 2)
    splits_s = _lib.fn('dsnt_conv_wgrad_f16x3_splits')(C.byref(g), 2)
    assert nws_s == splits_s * Cout * (Cin + 1) and splits_s <= splits
    ws3 = torch.full((nws_s,), float('nan'), device=dev)
    dw3, db3 = torch.empty_like(dw16), torch.empty_like(db16)
    call('dsnt_conv_wgrad_f16x3', *args, ptr(ws3), ptr(dw3), ptr(db3), 2, ptr(ab), ptr(gb), C.byref(g))
    assert bool(torch.isfinite(ws3).all())
    e3 = (dw3.cpu().double() - ref).abs().max().item()
    assert e3 <= 3e-5 * scale and e3 <= max(4 * e32, 2e-6 * scale), (e3, e32)
    assert (db3.cpu().double() - bref).abs().max().item() <= 3e-5 * bref.abs().max().item()


def test_f16x3_preparation_launches():
    """The per-step table-driven launches of the fp16x3 path: weight planes + bounds of several tensors at once equal
    the single-tensor entry points; the BatchNorm bound is max_c(|gamma_c| sqrt(M) + |beta_c|) in all 64 slots; the
    gradient-tensor writers leave max|written| in their slot (spread over the 64 slots, order-independent)."""
    import struct
    from dsnt import _lib
    from dsnt._lib import ptr, call
    dev = torch.device('cuda:0')
    ws = [synthetic.tensor('prep.w%d' % i, (n,), seed=30 + i, scale=sc).to(dev)
          for i, (n, sc) in enumerate([(128 * 9 * 128, 0.03), (256 * 128, 0.5), (64,  1e-4)])]
    rows, outs, bounds = [], [], []
    for w in ws:
        planes = torch.empty(2 * w.numel(), dtype=torch.float16, device=dev)
        bound = torch.zeros(64, device=dev)
        rows.append([w.data_ptr(), planes.data_ptr(), bound.data_ptr(), w.numel(), w.numel(), 0, 0])
        outs.append(planes); bounds.append(bound)
    table = torch.tensor(rows, dtype=torch.int64).to(dev)
    call('dsnt_f16_prep_weights', ptr(table), len(rows), 7)
    for w, planes, bound in zip(ws, outs, bounds):
        assert bool((bound == w.abs().max()).all())
        ref_b = torch.zeros(64, device=dev)
        ref_p = torch.empty_like(planes)
        call('dsnt_amax', ptr(w), w.numel(), ptr(ref_b))
        call('dsnt_split_f16x2', ptr(w), ptr(ref_p), w.numel(), w.numel(), ptr(ref_b))
        assert ref_b.max().item() == bound.max().item()
        assert torch.equal(ref_p.view(torch.int16), planes.view(torch.int16))
    gamma = synthetic.tensor('prep.g', (128,), seed=40).to(dev)
    beta = synthetic.tensor('prep.b', (128,), seed=41).to(dev)
    out = torch.zeros(64, device=dev)
    sqrt_m = float(2048) ** 0.5
    bits = struct.unpack('<I', struct.pack('<f', sqrt_m))[0]
    t2 = torch.tensor([[gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), 128, bits]], dtype=torch.int64).to(dev)
    call('dsnt_f16_prep_bn_bounds', ptr(t2), 1)
    want = (gamma.abs() * torch.tensor(sqrt_m, device=dev) + beta.abs()).max()
    assert bool(((out - want).abs() <= 1e-6 * want).all())
    # writers with an amax side output
    x = synthetic.tensor('prep.x', (2, 8, 8, 64), seed=42).to(dev)
    y = synthetic.tensor('prep.y', (2, 8, 8, 64), seed=43).to(dev)
    slot = torch.zeros(64, device=dev)
    call('dsnt_axpy_amax', ptr(x), ptr(y), 1.0, 1, x.numel(), ptr(slot))
    assert slot.max().item() == y.abs().max().item()
    call('dsnt_fill_zero', ptr(slot), 64)
    assert float(slot.abs().max()) == 0.0
    low = torch.empty(2, 4, 4, 64, device=dev)
    call('dsnt_upsample2_bwd_amax', ptr(x), ptr(low), 0, 2, 8, 8, 64, ptr(slot))
    assert slot.max().item() == low.abs().max().item()


# (N, H, W, Cin, Cout) of 3x3 / stride 1 / pad 1 convolutions the symmetric persistent kernel runs (csrc/conv3s.hip)
STREAM_CASES = [
    (2, 16, 32, 64, 64),          # 8 patches, one chunk pair ... two per tile
    (4, 64, 64, 128, 128),        # the Bottleneck's 3x3 at the 64 x 64 level: 512 patches
    (3, 8, 32, 128, 64),
    (2, 32, 64, 64, 128),
    (5, 4, 32, 32, 128),          # one chunk pair per tile: the slot parity flips from tile to tile
    (40, 64, 64, 32, 64),         # 1280 patches on 512 persistent workgroups: 2-3 tiles each, ragged
    (2, 32, 32, 128, 128),        # the oracle-sized model tests: 16 patches, one per image row block
    (2, 64, 64, 64, 64),
    (32, 16, 16, 128, 128),       # 16 pixels wide: 8 x 16 patches (the 16 x 16 hourglass level at batch 32: 64 patches)
    (3, 8, 16, 64, 64),
    (2, 24, 48, 32, 128),         # W % 32 != 0 -> 8 x 16 patches, three per row
    (5, 64, 64, 64, 128),         # 640 patches of 128 columns: the 16x16x32 form (forward), one or two tiles per workgroup
]
# (128 columns on at most 128 patches run as two 64-column halves per patch — csrc/conv3s.hip SP: cases 1, 3, 4, 6, 8, 10)


@pytest.mark.parametrize('case', STREAM_CASES)
@pytest.mark.parametrize('mode', ['plain', 'pro', 'pro_res', 'res', 'bnb', 'pro_norelu', 'bnb_norelu'])
def test_conv3_stream_kernel(case, mode):
    """dsnt_conv_fwd_f16x3_stream (persistent symmetric 3x3 kernel, weights in stream order) against
    dsnt_conv_fwd_f16x3_ex on the same operands: the stream planes are the plain planes permuted; without a residual the
    outputs of the 32x32x16 form are bit-identical (same K order, same MFMA order, same epilogue arithmetic), those of the
    16x16x32 form (Cout 128, 4 x 32 patches) agree to fp32 rounding of the sum; with a residual to an ulp of the sum; per-patch statistics add up to the same column sums; the BatchNorm-backward epilogue masks identically; and
    the result holds the fp32 bar against torch fp64."""
    from dsnt import _lib
    from dsnt._lib import ptr, call, BnBwdEpilogue
    N, H, W, Cin, Cout = case
    dev = torch.device('cuda:0')
    tag = 's3' + '_'.join(map(str, case))
    g = _geom(N, H, W, Cin, Cout, 3, 3, 1, 1, 1)
    assert _lib.fn('dsnt_conv_fwd_stream_ok')(C.byref(g)) == 1
    M = N * H * W
    x = synthetic.tensor(tag + 'x', (N, Cin, H, W), seed=1)
    w = synthetic.tensor(tag + 'w', (Cout, Cin, 3, 3), seed=1, scale=(2.0 / (Cin * 9)) ** 0.5)
    b = synthetic.tensor(tag + 'b', (Cout,), seed=1, scale=0.1)
    sc = synthetic.tensor(tag + 's', (Cin,), seed=1, kind='uniform').abs() + 0.5
    sh = synthetic.tensor(tag + 'h', (Cin,), seed=1, scale=0.3)
    res = synthetic.tensor(tag + 'r', (N, Cout, H, W), seed=1)
    relu = 0 if mode.endswith('_norelu') else 1
    mode = mode.replace('_norelu', '')
    pro = mode in ('pro', 'pro_res')
    xd = _nhwc(x).to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    n = wd.numel()
    plain = torch.empty(2 * n, dtype=torch.float16, device=dev)
    strm = torch.full((2 * n,), float('nan'), dtype=torch.float16, device=dev)
    wb, wb2 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    table = torch.tensor([[wd.data_ptr(), plain.data_ptr(), wb.data_ptr(), n, n, 0, 0],
                          [wd.data_ptr(), strm.data_ptr(), wb2.data_ptr(), n, n, Cout, Cin]], dtype=torch.int64).to(dev)
    call('dsnt_f16_prep_weights', ptr(table), 2, 7)
    assert torch.equal(wb, wb2)
    want = plain.view(2, Cout, 9, Cin // 16, 16).permute(0, 3, 2, 1, 4).contiguous().view(-1)
    assert torch.equal(want.view(torch.int16), strm.view(torch.int16))
    bd, scd, shd, resd = b.to(dev), sc.to(dev), sh.to(dev), _nhwc(res).to(dev)
    tiles = M // 128
    if mode == 'bnb':
        # the launch as a data gradient: A = a gradient tensor, epilogue = ReLU mask of bn(xin) + the two BN-backward sums
        xin = _nhwc(synthetic.tensor(tag + 'xin', (N, Cout, H, W), seed=3)).to(dev)
        mean = synthetic.tensor(tag + 'm', (Cout,), seed=4, scale=0.2).to(dev)
        invstd = (synthetic.tensor(tag + 'i', (Cout,), seed=5, kind='uniform').abs() + 0.5).to(dev)
        bsc = (synthetic.tensor(tag + 'bs', (Cout,), seed=6, kind='uniform') + 0.2).to(dev)
        bsh = synthetic.tensor(tag + 'bh', (Cout,), seed=7, scale=0.3).to(dev)
        bnb = BnBwdEpilogue(ptr(xin), ptr(bsc), ptr(bsh), ptr(mean), ptr(invstd), relu)
    ab = torch.zeros(64, device=dev)
    act = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) if pro else x
    if pro and relu:
        act = F.relu(act)
    ab[11] = act.abs().max().item() * 3.0
    outs = []
    for fn, planes in (('dsnt_conv_fwd_f16x3_ex', plain), ('dsnt_conv_fwd_f16x3_stream', strm)):
        y = torch.full((N, H, W, Cout), float('nan'), device=dev)
        stats = torch.full((tiles, 2, Cout), float('nan'), device=dev)
        amax = torch.zeros(64, device=dev)
        tail = _lib.BnTail()
        if mode != 'bnb':
            tail.amax = amax.data_ptr()
        call(fn, ptr(xd), ptr(planes), n, ptr(wb), ptr(ab), None if mode == 'bnb' else ptr(bd), ptr(y),
             ptr(scd) if pro else None, ptr(shd) if pro else None, relu,
             ptr(resd) if mode in ('res', 'pro_res') else None, None, ptr(stats), C.byref(g),
             C.byref(bnb) if mode == 'bnb' else None, C.byref(tail))
        outs.append((y, stats, amax))
    if mode == 'pro':       # DSNT_CONV_SHARE_CHIP (bit 1 of in_relu): fewer persistent workgroups, the same result bit for bit
        y2 = torch.full((N, H, W, Cout), float('nan'), device=dev)
        st2 = torch.full((tiles, 2, Cout), float('nan'), device=dev)
        call('dsnt_conv_fwd_f16x3_stream', ptr(xd), ptr(strm), n, ptr(wb), ptr(ab), ptr(bd), ptr(y2), ptr(scd), ptr(shd), relu | 2,
             None, None, ptr(st2), C.byref(g), None, None)
        assert torch.equal(y2, outs[1][0]) and torch.equal(st2, outs[1][1])
    torch.cuda.synchronize()
    (y0, st0, am0), (y1, st1, am1) = outs
    assert not bool(torch.isnan(y1).any()) and not bool(torch.isnan(st1).any())
    scale = y0.abs().max().item()
    # Cout 128 on 32-pixel-wide patches runs on v_mfma_f32_16x16x32_f16 (csrc/conv3s.hip MF16): an MFMA sums BOTH K-steps of a
    # pair, so the same products are added in another association — fp32 rounding of the 1152..3456-term sums, not bit identity
    mf16 = Cout == 128 and W % 32 == 0 and H % 4 == 0 and M // 128 > 128
    if mode in ('plain', 'pro', 'bnb') and H % 8 == 0 and not mf16:   # (H % 8 != 0: _ex runs the implicit-GEMM kernel, K order (tap, channel))
        assert torch.equal(y0, y1)
    else:
        assert (y0 - y1).abs().max().item() <= (5e-7 if H % 8 == 0 else 2e-6) * scale
        if mode == 'bnb':        # the ReLU mask comes from the residual operand, not from the sums: the same zeros
            assert torch.equal(y0 == 0, y1 == 0)
    if mode == 'bnb':
        assert float((y1 == 0).float().mean()) > (0.2 if relu else -1.0)
    else:
        assert am1.max().item() == y1.abs().max().item()
    # statistics: other tiles, other order — the column sums agree to fp32 rounding of a 128-term sum
    for k in range(2):
        a0, a1 = st0[:, k].double().sum(0), st1[:, k].double().sum(0)
        assert (a0 - a1).abs().max().item() <= 2e-6 * max(1.0, st0[:, k].double().abs().sum(0).max().item())
    if mode != 'bnb':
        y64 = F.conv2d(act.double(), w.double(), b.double(), padding=1)
        if mode in ('res', 'pro_res'):
            y64 = y64 + res.double()
        y32 = F.conv2d(act, w, b, padding=1) + (res if mode in ('res', 'pro_res') else 0)
        got = y1.cpu().permute(0, 3, 1, 2).double()
        err16, err32 = (got - y64).abs().max().item(), (y32.double() - y64).abs().max().item()
        assert err16 <= max(4 * err32, 2e-6 * y64.abs().max().item()), (err16, err32)
        cols = y1.double().reshape(M, Cout)
        assert (st1[:, 0].double().sum(0) - cols.sum(0)).abs().max().item() <= 1e-4 * max(1.0, cols.sum(0).abs().max().item())
        assert (st1[:, 1].double().sum(0) - (cols * cols).sum(0)).abs().max().item() <= 1e-4 * (cols * cols).sum(0).max().item()


@pytest.mark.parametrize('case', [(20, 64, 64, 128, 128), (3, 64, 64, 64, 128)])
def test_conv3_stream_kernel_with_the_weight_ring_filled_by_lds_dma_is_reproducible(case):
    """The 16x16x32 form streams its weights global -> LDS directly (buffer_load ... lds, csrc/conv3s.hip `dmaB`), with hand-counted
    s_waitcnt vmcnt in front of the barriers — loads hipcc does not track.  A misplaced wait is a race, and a race shows as
    launches that disagree: 12 launches over 2560 patches (five tiles per persistent workgroup, the ring wrapping through every tile
    change) / 384 patches (fewer tiles than workgroups), all bit-identical, and equal to the fp64 convolution to the fp32 bar."""
    from dsnt import _lib
    from dsnt._lib import ptr, call
    N, H, W, Cin, Cout = case
    dev = torch.device('cuda:0')
    g = _geom(N, H, W, Cin, Cout, 3, 3, 1, 1, 1)
    x = synthetic.tensor('dmax', (N, Cin, H, W), seed=2)
    w = synthetic.tensor('dmaw', (Cout, Cin, 3, 3), seed=2, scale=(2.0 / (Cin * 9)) ** 0.5)
    xd = _nhwc(x).to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    n = wd.numel()
    strm = torch.empty(2 * n, dtype=torch.float16, device=dev)
    wb = torch.zeros(64, device=dev)
    table = torch.tensor([[wd.data_ptr(), strm.data_ptr(), wb.data_ptr(), n, n, Cout, Cin]], dtype=torch.int64).to(dev)
    call('dsnt_f16_prep_weights', ptr(table), 1, 7)
    ab = torch.zeros(64, device=dev)
    ab[5] = x.abs().max().item() * 2.0
    outs = []
    for _ in range(12):
        y = torch.full((N, H, W, Cout), float('nan'), device=dev)
        stats = torch.full((N * H * W // 128, 2, Cout), float('nan'), device=dev)
        call('dsnt_conv_fwd_f16x3_stream', ptr(xd), ptr(strm), n, ptr(wb), ptr(ab), None, ptr(y), None, None, 0, None, None,
             ptr(stats), C.byref(g), None, None)
        outs.append((y, stats))
    torch.cuda.synchronize()
    for y, st in outs[1:]:
        assert torch.equal(y, outs[0][0]) and torch.equal(st, outs[0][1])
    y64 = F.conv2d(x.double(), w.double(), None, padding=1)
    y32 = F.conv2d(x, w, None, padding=1)
    got = outs[0][0].cpu().permute(0, 3, 1, 2).double()
    err16, err32 = (got - y64).abs().max().item(), (y32.double() - y64).abs().max().item()
    assert err16 <= max(4 * err32, 2e-6 * y64.abs().max().item()), (err16, err32)


def test_conv3_stream_kernel_16x16x32_form_in_every_mode():
    """The 16x16x32 MFMA form of csrc/conv3s.hip is the default for the forward launches only; DSNT_X_C3_MF16=7 puts the
    data-gradient modes (BatchNorm-backward epilogue, folded BatchNorm backward) on it too.  The switch is read once per
    process, so the stream tests run again in a child with it set."""
    import subprocess
    import sys
    env = dict(os.environ, DSNT_X_C3_MF16='7')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-p', 'no:cacheprovider',
                        '-k', 'test_conv3_stream_kernel[ or folded_in'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize('case', [c for c in STREAM_CASES if c[3] in (64, 128)])
@pytest.mark.parametrize('relu,big_mean', [(1, False), (0, False), (1, True)])
def test_conv3_stream_data_gradient_with_the_batchnorm_backward_folded_in(case, relu, big_mean):
    """dsnt_conv_dgrad_f16x3_stream_apply (csrc/conv3s.hip MODE 4: the BatchNorm backward of the layer BEHIND a 3x3 convolution
    formed in the operand load of that convolution's data gradient; /root/reference/src/dsnt/hourglass.py:36-40) against the
    two launches it replaces — dsnt_bn_act_bwd_apply, then dsnt_conv_fwd_f16x3_stream on its output:
      * dy_out (every pixel exactly once, halo pixels never) equals the apply kernel's dy to fp32 rounding of the affine
        form P dz + R y + S (its conditioning is a scale / shift BatchNorm's: eps |mean| / std — `big_mean` puts the
        BatchNorm input at mean 20 std);
      * the launch's own output (masked dz of the BatchNorm in front, its two sums per patch, max |dz|) equals the
        two-launch path's to the same rounding carried through the convolution;
      * DSNT_CONV_SHARE_CHIP changes no bit."""
    from dsnt import _lib
    from dsnt._lib import ptr, call, BnBwdEpilogue, BnBwdApply
    N, H, W, Cin, Cout = case            # of the data-gradient launch: Cin = channels of dL/dy, Cout = channels of dL/dx
    dev = torch.device('cuda:0')
    tag = 's4' + '_'.join(map(str, case)) + str(relu) + str(big_mean)
    g = _geom(N, H, W, Cin, Cout, 3, 3, 1, 1, 1)
    M = N * H * W
    w = synthetic.tensor(tag + 'w', (Cout, Cin, 3, 3), seed=1, scale=(2.0 / (Cin * 9)) ** 0.5)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    n = wd.numel()
    strm = torch.empty(2 * n, dtype=torch.float16, device=dev)
    wb = torch.zeros(64, device=dev)
    table = torch.tensor([[wd.data_ptr(), strm.data_ptr(), wb.data_ptr(), n, n, Cout, Cin]], dtype=torch.int64).to(dev)
    call('dsnt_f16_prep_weights', ptr(table), 1, 7)
    # the BatchNorm behind the convolution: its input y, its vectors, the reduced coefficients; dz behind it (ReLU-masked)
    mu = synthetic.tensor(tag + 'mu', (Cin,), seed=2, scale=20.0 if big_mean else 0.5).to(dev)
    std = (synthetic.tensor(tag + 'sd', (Cin,), seed=3, kind='uniform').abs() + 0.5).to(dev)
    yin = (_nhwc(synthetic.tensor(tag + 'y', (N, Cin, H, W), seed=4)).to(dev) * std + mu).contiguous()
    dz = _nhwc(synthetic.tensor(tag + 'dz', (N, Cin, H, W), seed=5)).to(dev)
    dz = (dz * (_nhwc(synthetic.tensor(tag + 'mk', (N, Cin, H, W), seed=6)).to(dev) > 0)).contiguous()
    invstd = (1.0 / std).contiguous()
    gamma = (synthetic.tensor(tag + 'ga', (Cin,), seed=7, kind='uniform') + 0.3).to(dev)
    scale = (gamma * invstd).contiguous()
    shift = torch.zeros(Cin, device=dev)
    xhat = (yin.view(M, Cin) - mu) * invstd
    coef = torch.cat([dz.view(M, Cin).mean(0), (dz.view(M, Cin) * xhat).mean(0)]).contiguous()
    # the BatchNorm in front of the convolution (the epilogue)
    xin = _nhwc(synthetic.tensor(tag + 'xin', (N, Cout, H, W), seed=8)).to(dev)
    mean = synthetic.tensor(tag + 'm', (Cout,), seed=9, scale=0.2).to(dev)
    istd = (synthetic.tensor(tag + 'i', (Cout,), seed=10, kind='uniform').abs() + 0.5).to(dev)
    bsc = (synthetic.tensor(tag + 'bs', (Cout,), seed=11, kind='uniform') + 0.2).to(dev)
    bsh = synthetic.tensor(tag + 'bh', (Cout,), seed=12, scale=0.3).to(dev)
    bnb = BnBwdEpilogue(ptr(xin), ptr(bsc), ptr(bsh), ptr(mean), ptr(istd), relu)
    tiles = M // 128
    # reference: the apply launch, then the stream kernel on its output
    dy = torch.full((N, H, W, Cin), float('nan'), device=dev)
    call('dsnt_bn_act_bwd_apply', ptr(dz), ptr(yin), ptr(scale), ptr(shift), ptr(mu), ptr(invstd), ptr(coef), 0, ptr(dy), 0, M, Cin)
    dy64 = (scale.double() * (dz.view(M, Cin).double() - coef[:Cin].double() -
                              (yin.view(M, Cin).double() - mu.double()) * invstd.double() * coef[Cin:].double())).view(N, H, W, Cin)
    ab = torch.zeros(64, device=dev)
    ab[5] = float(dy64.abs().max()) * 2.5
    outs = []
    for fold, flags in ((False, 0), (True, 0), (True, 2)):
        out = torch.full((N, H, W, Cout), float('nan'), device=dev)
        stats = torch.full((tiles, 2, Cout), float('nan'), device=dev)
        amax = torch.zeros(64, device=dev)
        tail = _lib.BnTail()
        tail.amax = amax.data_ptr()
        dyo = torch.full((N, H, W, Cin), float('nan'), device=dev)
        if fold:
            ap = BnBwdApply(ptr(yin), ptr(scale), ptr(mu), ptr(invstd), ptr(coef))
            call('dsnt_conv_dgrad_f16x3_stream_apply', ptr(dz), C.byref(ap), ptr(dyo), ptr(strm), n, ptr(wb), ptr(ab), ptr(out),
                 ptr(stats), flags, C.byref(g), C.byref(bnb), C.byref(tail))
        else:
            call('dsnt_conv_fwd_f16x3_stream', ptr(dy), ptr(strm), n, ptr(wb), ptr(ab), None, ptr(out), None, None, 0, None, None,
                 ptr(stats), C.byref(g), C.byref(bnb), C.byref(tail))
        outs.append((out, stats, amax, dyo))
    torch.cuda.synchronize()
    (o0, s0, a0, _), (o1, s1, a1, d1), (o2, s2, a2, d2) = outs
    assert torch.equal(o1, o2) and torch.equal(d1, d2) and torch.equal(a1.max(), a2.max())      # the share flag changes no bit
    assert not bool(torch.isnan(d1).any()) and not bool(torch.isnan(o1).any()) and not bool(torch.isnan(s1).any())
    # dy_out against fp64, held to the apply kernel's own distance (x 4) plus the affine form's conditioning
    sc_dy = float(dy64.abs().max())
    e_ref, e_new = float((dy.double() - dy64).abs().max()), float((d1.double() - dy64).abs().max())
    cond = float((mu.abs() * invstd).max())
    assert e_new <= max(4 * e_ref, 2e-7 * (1 + cond) * sc_dy), (e_new, e_ref, cond)
    # the convolution's output: same mask, values to the rounding of dy carried through 9 Cin products
    so = float(o0.abs().max())
    tol = (4e-7 * (1 + cond)) * so
    assert (o0 - o1).abs().max().item() <= tol, ((o0 - o1).abs().max().item(), tol)
    if relu:
        assert float((o1 == 0).float().mean()) > 0.2 and float(((o0 == 0) != (o1 == 0)).float().mean()) <= 1e-5
    assert abs(a0.max().item() - a1.max().item()) <= tol
    for k in range(2):
        b0, b1 = s0[:, k].double().sum(0), s1[:, k].double().sum(0)
        assert (b0 - b1).abs().max().item() <= 4e-6 * (1 + cond) * max(1.0, s0[:, k].double().abs().sum(0).max().item())


def test_share_chip_flag_of_the_streaming_1x1_kernel_changes_no_bit():
    """DSNT_CONV_SHARE_CHIP (bit 1 of in_relu) on dsnt_conv_fwd_f16x3_ex: the streaming 1x1 kernel (>= 65536 rows) starts half as
    many persistent workgroups; output, statistics partials and amax slot content are bit-identical."""
    from dsnt import _lib
    from dsnt._lib import ptr, call
    dev = torch.device('cuda:0')
    N, H, W, Cin, Cout = 16, 64, 64, 128, 256
    g = _geom(N, H, W, Cin, Cout, 1, 1, 1, 0, 1)
    x = synthetic.tensor('shx', (N, H, W, Cin), seed=1).to(dev)
    w = synthetic.tensor('shw', (Cout, 1, 1, Cin), seed=2, scale=0.1).to(dev)
    b = synthetic.tensor('shb', (Cout,), seed=3, scale=0.1).to(dev)
    sc = (synthetic.tensor('shs', (Cin,), seed=4, kind='uniform').abs() + 0.5).to(dev)
    sh = synthetic.tensor('shh', (Cin,), seed=5, scale=0.3).to(dev)
    res = synthetic.tensor('shr', (N, H, W, Cout), seed=6).to(dev)
    planes = torch.empty(2 * w.numel(), dtype=torch.float16, device=dev)
    wb, ab = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    call('dsnt_amax', ptr(w), w.numel(), ptr(wb))
    call('dsnt_split_f16x2', ptr(w), ptr(planes), w.numel(), w.numel(), ptr(wb))
    ab.fill_(float(torch.relu(x * sc + sh).max()) * 2)
    M = N * H * W
    outs = []
    for flag in (1, 1 | 2):
        y = torch.full((N, H, W, Cout), float('nan'), device=dev)
        st = torch.full((M // 128, 2, Cout), float('nan'), device=dev)
        call('dsnt_conv_fwd_f16x3_ex', ptr(x), ptr(planes), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y), ptr(sc), ptr(sh), flag,
             ptr(res), None, ptr(st), C.byref(g), None, None)
        outs.append((y, st))
    torch.cuda.synchronize()
    assert not bool(torch.isnan(outs[1][0]).any()) and not bool(torch.isnan(outs[1][1]).any())
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_conv3_stream_refusals():
    """dsnt_conv_fwd_f16x3_stream refuses what the kernel does not carry: a second residual, ticket counters, other
    geometries (the caller keeps dsnt_conv_fwd_f16x3_ex for those)."""
    from dsnt import _lib
    from dsnt._lib import ptr
    dev = torch.device('cuda:0')
    ok = _lib.fn('dsnt_conv_fwd_stream_ok')
    assert ok(C.byref(_geom(2, 16, 32, 64, 64, 3, 3, 1, 1, 1))) == 1
    assert ok(C.byref(_geom(2, 16, 16, 64, 64, 3, 3, 1, 1, 1))) == 1          # 8 x 16 patches
    for bad in [(2, 12, 16, 64, 64, 3, 3, 1, 1, 1), (2, 6, 32, 64, 64, 3, 3, 1, 1, 1), (2, 16, 32, 48, 64, 3, 3, 1, 1, 1), (2, 16, 24, 64, 64, 3, 3, 1, 1, 1),
                (2, 16, 32, 64, 96, 3, 3, 1, 1, 1), (2, 16, 32, 64, 64, 1, 1, 1, 0, 1), (2, 16, 32, 256, 64, 3, 3, 1, 1, 1),
                (2, 16, 32, 64, 64, 3, 3, 1, 2, 2)]:
        assert ok(C.byref(_geom(*bad))) == 0, bad
    g = _geom(2, 16, 32, 64, 64, 3, 3, 1, 1, 1)
    x = torch.zeros(2, 16, 32, 64, device=dev)
    y, r = torch.empty_like(x), torch.zeros_like(x)
    planes = torch.zeros(2 * 64 * 9 * 64, dtype=torch.float16, device=dev)
    wb = torch.ones(64, device=dev)
    f = _lib.fn('dsnt_conv_fwd_f16x3_stream')
    assert f(ptr(x), ptr(planes), 64 * 9 * 64, ptr(wb), ptr(wb), None, ptr(y), None, None, 0, ptr(r), ptr(r), None,
             C.byref(g), None, None, None) != 0
    assert b'stream' in _lib.load().dsnt_last_error()
    torch.cuda.synchronize()


# (N, H, W, Cin [BN'd input channels], Cout [channels of the conv whose data gradient is taken], k)
BNB_CASES = [
    (4, 64, 64, 128, 128, 3),     # 3x3 halo-tile kernel (split paths) / 128x128 implicit GEMM (fp32 path)
    (4, 64, 64, 256, 128, 1),     # reducing 1x1 of a Bottleneck: data gradient 128 -> 256
    (4, 64, 64, 128, 256, 1),     # expanding 1x1: data gradient 256 -> 128
    (2, 8, 8, 128, 128, 3),       # 128 rows: the K-split kernel on the fp32 path
    (2, 16, 16, 128, 128, 3),     # 512 rows: 32x128 tiles / K-split
    (16, 64, 64, 256, 128, 1),    # 65536 rows: the streaming 1x1 kernel (fp16x3 path), data gradient 128 -> 256
    (16, 64, 64, 128, 256, 1),    # ... 256 -> 128
]


@pytest.mark.parametrize('case', BNB_CASES)
@pytest.mark.parametrize('path', ['f32', 'bf16x6', 'f16x3'])
@pytest.mark.parametrize('with_amax', [False, True])
def test_bn_backward_epilogue_with_relu_vs_autograd(case, path, with_amax):
    """ONE layer x -> BatchNorm(train) -> ReLU -> conv, backward through the production chain with the ReLU ON:
    data-gradient launch with the BN-backward epilogue (ReLU mask + the two per-channel sums) -> finalise ->
    apply, against torch autograd (fp64).  A single layer has no flip amplification, so the bars are tight
    (2e-5 of the scale, as for the plain convolutions): inputs are nudged so that no pre-activation sits within
    5e-4 of the ReLU kink, where two fp32 evaluations may legitimately disagree.  with_amax: the launch also leaves max|dz|
    (dsnt_out_bounds.amax beside the epilogue: what dsnt_bn_bwd_finalize_bound makes the bound of a folded BatchNorm backward from)."""
    from dsnt import _lib
    from dsnt._lib import ptr, call, BnBwdEpilogue
    N, H, W, Cin, Cout, k = case
    dev = torch.device('cuda:0')
    tag = 'bnb' + '_'.join(map(str, case))
    pad = k // 2
    M = N * H * W
    x = synthetic.tensor(tag + 'x', (N, Cin, H, W), seed=21) * 1.3 + 0.2
    gamma = synthetic.tensor(tag + 'g', (Cin,), seed=21, kind='uniform') * 0.5 + 1.0
    beta = synthetic.tensor(tag + 'b', (Cin,), seed=21, scale=0.2)
    w = synthetic.tensor(tag + 'w', (Cout, Cin, k, k), seed=21, scale=(2.0 / (Cin * k * k)) ** 0.5)
    gy = synthetic.tensor(tag + 'gy', (N, Cout, H, W), seed=22)
    for _ in range(4):       # keep every pre-activation away from the kink (batch statistics move a little: iterate)
        z = F.batch_norm(x.double(), None, None, gamma.double(), beta.double(), True, 0.0, 1e-5)
        near = z.abs() < 1e-3
        if not near.any():
            break
        std = x.double().var((0, 2, 3), unbiased=False, keepdim=True).add(1e-5).sqrt()
        x = (x.double() + torch.where(near, torch.sign(z) * 4e-3 * std / gamma.double().view(1, -1, 1, 1),
                                      torch.zeros_like(z))).float()
    z = F.batch_norm(x.double(), None, None, gamma.double(), beta.double(), True, 0.0, 1e-5)
    assert float(z.abs().min()) >= 5e-4
    # reference: fp64 autograd through the layer
    xr = x.double().requires_grad_()
    gr, br = gamma.double().requires_grad_(), beta.double().requires_grad_()
    a = F.relu(F.batch_norm(xr, None, None, gr, br, True, 0.0, 1e-5))
    F.conv2d(a, w.double(), None, padding=pad).backward(gy.double())

    xd, gyd = _nhwc(x).to(dev), _nhwc(gy).to(dev)
    gd, bd = gamma.to(dev), beta.to(dev)
    tiles = (M + 127) // 128
    part = torch.empty(tiles, 2, Cin, device=dev)
    call('dsnt_bn_stats', ptr(xd), ptr(part), M, Cin)
    mean, invstd, scale, shift = (torch.empty(Cin, device=dev) for _ in range(4))
    call('dsnt_bn_finalize', ptr(part), tiles, M, Cin, ptr(gd), ptr(bd), None, None, 0.1, 1e-5, 1,
         ptr(mean), ptr(invstd), ptr(scale), ptr(shift))
    # data gradient of the conv: forward kernel on dY with tap-flipped, transposed weights, Cout -> Cin channels
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    wdg = torch.empty(Cin, k, k, Cout, device=dev)
    call('dsnt_conv_pack_dgrad', ptr(wd), ptr(wdg), Cout, k, k, Cin)
    g = _geom(N, H, W, Cout, Cin, k, k, 1, pad, 1)
    bnb = BnBwdEpilogue(ptr(xd), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), 1)
    dz = torch.empty(N, H, W, Cin, device=dev)
    dgamma, dbeta = torch.empty(Cin, device=dev), torch.empty(Cin, device=dev)
    coef = torch.empty(2, Cin, device=dev)
    bm = _lib.fn('dsnt_conv_fwd_bm')(C.byref(g)) if path == 'f32' else 128
    tail, dz_amax = None, torch.zeros(64, device=dev)
    if with_amax:
        tail_s = _lib.BnTail()
        tail_s.amax = dz_amax.data_ptr()
        tail = C.byref(tail_s)
    if path == 'f32':
        tiles_d = (M + bm - 1) // bm
        part_d = torch.zeros(tiles_d, 2, Cin, device=dev)
        call('dsnt_conv_fwd_ex', ptr(gyd), ptr(wdg), None, ptr(dz), None, None, 0, None, None, ptr(part_d),
             C.byref(g), C.byref(bnb), tail)
    else:
        if not _lib.fn('dsnt_conv_bf16x6_ok')(C.byref(g)):
            pytest.skip('geometry not supported by the split-precision kernels')
        tiles_d = (M + 127) // 128
        part_d = torch.zeros(tiles_d, 2, Cin, device=dev)
        if path == 'bf16x6':
            planes = torch.empty(3 * wdg.numel(), dtype=torch.bfloat16, device=dev)
            call('dsnt_split_bf16x3', ptr(wdg), ptr(planes), wdg.numel())
            call('dsnt_conv_fwd_bf16x6_ex', ptr(gyd), ptr(planes), wdg.numel(), None, ptr(dz), None, None, 0, None,
                 None, ptr(part_d), C.byref(g), C.byref(bnb), tail)
        else:
            wb, ab = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
            call('dsnt_amax', ptr(wdg), wdg.numel(), ptr(wb))
            call('dsnt_amax', ptr(gyd), gyd.numel(), ptr(ab))
            planes = torch.empty(2 * wdg.numel(), dtype=torch.float16, device=dev)
            call('dsnt_split_f16x2', ptr(wdg), ptr(planes), wdg.numel(), wdg.numel(), ptr(wb))
            call('dsnt_conv_fwd_f16x3_ex', ptr(gyd), ptr(planes), wdg.numel(), ptr(wb), ptr(ab), None, ptr(dz), None,
                 None, 0, None, None, ptr(part_d), C.byref(g), C.byref(bnb), tail)
    # dz = dL/d(bn output) masked by the ReLU: autograd's gradient at the BN output
    dz_ref = torch.autograd.grad(F.conv2d(F.relu(zz := z.clone().requires_grad_()), w.double(), None, padding=pad),
                                 zz, gy.double())[0]
    tol = 2e-5
    s_dz = dz_ref.abs().max().item()
    assert (dz.cpu().permute(0, 3, 1, 2).double() - dz_ref).abs().max().item() <= tol * s_dz
    assert float((dz == 0).float().mean()) > 0.2          # the mask really is on
    if with_amax:
        # max|dz| exactly, and the finalise launch that also leaves the bound of dx = scale (dz - c0 - xhat c1): it dominates dx
        assert float(dz_amax.max()) == float(dz.abs().max())
        bound = torch.zeros(64, device=dev)
        call('dsnt_bn_bwd_finalize_bound', ptr(part_d), tiles_d, M, Cin, ptr(dgamma), ptr(dbeta), 0, ptr(coef), ptr(scale),
             ptr(dz_amax), ptr(bound))
        ref = [torch.empty_like(dgamma), torch.empty_like(dbeta), torch.empty_like(coef)]
        call('dsnt_bn_bwd_finalize', ptr(part_d), tiles_d, M, Cin, ptr(ref[0]), ptr(ref[1]), 0, ptr(ref[2]))
        for got_, want_ in zip((dgamma, dbeta, coef), ref):
            assert torch.equal(got_, want_)
    else:
        call('dsnt_bn_bwd_finalize', ptr(part_d), tiles_d, M, Cin, ptr(dgamma), ptr(dbeta), 0, ptr(coef))
    dx = torch.empty(N, H, W, Cin, device=dev)
    call('dsnt_bn_act_bwd_apply', ptr(dz), ptr(xd), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(coef), 0,
         ptr(dx), 0, M, Cin)
    assert (dgamma.cpu().double() - gr.grad).abs().max().item() <= tol * max(1.0, gr.grad.abs().max().item())
    assert (dbeta.cpu().double() - br.grad).abs().max().item() <= tol * max(1.0, br.grad.abs().max().item())
    s_dx = xr.grad.abs().max().item()
    assert (dx.cpu().permute(0, 3, 1, 2).double() - xr.grad).abs().max().item() <= tol * s_dx
    if with_amax:
        assert float(dx.abs().max()) <= float(bound.max()) <= 4096.0 * float(dx.abs().max())


@pytest.mark.parametrize('shape', [(2, 8, 8, 128, 128, 3), (2, 8, 8, 256, 128, 1), (4, 16, 16, 128, 256, 1), (2, 4, 4, 128, 128, 3)],
                         ids=['ksplit_3x3', 'ksplit_1x1_256', 'tiled_1x1_1024rows', 'ksplit_4x4'])
def test_batchnorm_finalised_in_the_consumers_prologue(shape):
    """dsnt_conv_fwd_pro (the BatchNorm of the A operand finalised in the convolution's own prologue: the 8x8 / 4x4
    hourglass levels) == dsnt_bn_finalize + dsnt_conv_fwd_ex: the four BatchNorm vectors to fp32 rounding of the same
    fp64 sums (another summation order), the running statistics moved exactly once, the convolution output to 1e-6 of
    its scale; and dsnt_bn_act_bwd_apply_pro == dsnt_bn_bwd_finalize + dsnt_bn_act_bwd_apply (dgamma / dbeta, their
    accumulate form, dx)."""
    from dsnt import _lib
    from dsnt._lib import ptr, call, BnPrologue
    dev = torch.device('cuda:0')
    N, H, W, Cin, Cout, k = shape
    g = _geom(N, H, W, Cin, Cout, k, k, 1, k // 2, 1)
    M = N * H * W
    tag = 'pro' + '_'.join(map(str, shape))
    x = synthetic.tensor(tag + 'x', (N, H, W, Cin), seed=31).to(dev)
    w = (synthetic.tensor(tag + 'w', (Cout, k, k, Cin), seed=31) * 0.05).to(dev)
    b = (synthetic.tensor(tag + 'b', (Cout,), seed=31) * 0.1).to(dev)
    gamma = (synthetic.tensor(tag + 'g', (Cin,), seed=31, kind='uniform') + 1.5).to(dev)
    beta = (synthetic.tensor(tag + 'be', (Cin,), seed=31) * 0.2).to(dev)
    tiles = (M + 31) // 32          # 32-row tiles, as the small-M convolution kernels write them
    part = torch.empty(tiles, 2, Cin, device=dev)
    xr = x.view(-1, Cin)
    for t in range(tiles):          # per-tile sums as a producer's epilogue leaves them
        blk = xr[t * 32:(t + 1) * 32]
        part[t, 0], part[t, 1] = blk.sum(0), (blk * blk).sum(0)
    assert _lib.fn('dsnt_conv_fwd_pro_ok')(C.byref(g), tiles, Cin) == 1
    bm = _lib.fn('dsnt_conv_fwd_bm')(C.byref(g))
    otiles = (M + bm - 1) // bm

    def vectors():
        return [torch.full((Cin,), float('nan'), device=dev) for _ in range(4)]
    # reference: separate launches
    v_ref, rm_ref, rv_ref = vectors(), torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev)
    call('dsnt_bn_finalize', ptr(part), tiles, M, Cin, ptr(gamma), ptr(beta), ptr(rm_ref), ptr(rv_ref), 0.1, 1e-5, 1,
         *[ptr(v) for v in v_ref])
    y_ref, st_ref = torch.empty(N, H, W, Cout, device=dev), torch.zeros(otiles, 2, Cout, device=dev)
    call('dsnt_conv_fwd_ex', ptr(x), ptr(w), ptr(b), ptr(y_ref), ptr(v_ref[2]), ptr(v_ref[3]), 1, None, None, ptr(st_ref),
         C.byref(g), None, None)
    # fused
    v, rm, rv = vectors(), torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev)
    pro = BnPrologue(ptr(part), tiles, Cin, M, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 0.1, 1e-5, *[ptr(t_) for t_ in v])
    y, st = torch.empty(N, H, W, Cout, device=dev), torch.zeros(otiles, 2, Cout, device=dev)
    call('dsnt_conv_fwd_pro', ptr(x), ptr(w), ptr(b), ptr(y), C.byref(pro), 1, None, None, ptr(st), C.byref(g), None)
    for a_, b_ in zip(v, v_ref):
        assert (a_ - b_).abs().max().item() <= 2e-6 * max(1.0, b_.abs().max().item())
    assert (rm - rm_ref).abs().max().item() <= 1e-6 and (rv - rv_ref).abs().max().item() <= 1e-6      # moved exactly once
    assert (y - y_ref).abs().max().item() <= 1e-6 * y_ref.abs().max().item()
    assert (st - st_ref).abs().max().item() <= 1e-5 * st_ref.abs().max().item()
    # errors: too many channels / tiles for the prologue
    assert _lib.fn('dsnt_conv_fwd_pro_ok')(C.byref(g), 16384 // Cin + 1, Cin) == 0
    # ---- backward: apply with the backward finalise in its prologue
    da = synthetic.tensor(tag + 'da', (N, H, W, Cin), seed=32).to(dev)
    bt = (M + 127) // 128
    bpart = torch.empty(bt, 2, Cin, device=dev)
    call('dsnt_bn_act_bwd_reduce', ptr(da), ptr(x), ptr(v_ref[2]), ptr(v_ref[3]), ptr(v_ref[0]), ptr(v_ref[1]), 1, ptr(bpart), M, Cin)
    for accp in (0, 1):
        dg_ref, db_ref = torch.full((Cin,), 0.25, device=dev), torch.full((Cin,), -0.5, device=dev)
        dg, db = dg_ref.clone(), db_ref.clone()
        coef_ref, coef = torch.empty(2, Cin, device=dev), torch.empty(2, Cin, device=dev)
        dx_ref, dx = torch.ones(N, H, W, Cin, device=dev), torch.ones(N, H, W, Cin, device=dev)
        call('dsnt_bn_bwd_finalize', ptr(bpart), bt, M, Cin, ptr(dg_ref), ptr(db_ref), accp, ptr(coef_ref))
        call('dsnt_bn_act_bwd_apply', ptr(da), ptr(x), ptr(v_ref[2]), ptr(v_ref[3]), ptr(v_ref[0]), ptr(v_ref[1]), ptr(coef_ref),
             1, ptr(dx_ref), accp, M, Cin)
        call('dsnt_bn_act_bwd_apply_pro', ptr(da), ptr(x), ptr(v_ref[2]), ptr(v_ref[3]), ptr(v_ref[0]), ptr(v_ref[1]), ptr(bpart),
             bt, ptr(dg), ptr(db), accp, ptr(coef), 1, ptr(dx), accp, M, Cin, None)
        for a_, b_ in ((dg, dg_ref), (db, db_ref), (coef, coef_ref), (dx, dx_ref)):
            assert (a_ - b_).abs().max().item() <= 2e-6 * max(1.0, b_.abs().max().item())
    # ---- the out-of-place forms: dx = base + value, `base` a tensor of its own that stays intact (dsnt_bn_act_bwd_apply_base /
    # _pro_base) == the in-place accumulation started from a copy of base, bit for bit; max |dx| in the bound slots
    base = synthetic.tensor(tag + 'base', (N, H, W, Cin), seed=33).to(dev)
    keep = base.clone()
    want = base.clone()
    call('dsnt_bn_act_bwd_apply', ptr(da), ptr(x), ptr(v_ref[2]), ptr(v_ref[3]), ptr(v_ref[0]), ptr(v_ref[1]), ptr(coef_ref),
         1, ptr(want), 1, M, Cin)
    got, am = torch.empty_like(base), torch.zeros(64, device=dev)
    call('dsnt_bn_act_bwd_apply_base', ptr(da), ptr(x), ptr(v_ref[2]), ptr(v_ref[3]), ptr(v_ref[0]), ptr(v_ref[1]), ptr(coef_ref),
         1, ptr(base), ptr(got), M, Cin, ptr(am))
    assert torch.equal(got, want) and torch.equal(base, keep) and am.max().item() == want.abs().max().item()
    dg, db, coef, got2 = torch.zeros(Cin, device=dev), torch.zeros(Cin, device=dev), torch.empty(2, Cin, device=dev), torch.empty_like(base)
    call('dsnt_bn_act_bwd_apply_pro_base', ptr(da), ptr(x), ptr(v_ref[2]), ptr(v_ref[3]), ptr(v_ref[0]), ptr(v_ref[1]), ptr(bpart),
         bt, ptr(dg), ptr(db), 0, ptr(coef), 1, ptr(base), ptr(got2), M, Cin, None)
    assert (got2 - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item()) and torch.equal(base, keep)
    assert _lib.fn('dsnt_bn_act_bwd_apply_base')(ptr(da), ptr(x), ptr(v_ref[2]), ptr(v_ref[3]), ptr(v_ref[0]), ptr(v_ref[1]),
                                                 ptr(coef_ref), 1, ptr(got), ptr(got), M, Cin, None, None) != 0      # base must differ from dx


@pytest.mark.parametrize('path', ['f32', 'bf16x6', 'f16x3'])
@pytest.mark.parametrize('k,with_res,N', [(1, True, 2), (3, False, 2), (1, True, 64)],
                         ids=['k1_res', 'k3', 'k1_res_65536rows'])
def test_epilogue_leaves_the_bound_of_its_output(path, k, with_res, N):
    """dsnt_out_bounds.amax: the launch raises a 64-slot bound to max|y| of what it wrote (bias and residual included) —
    the fp16x3 operand bound of a consumer that reads y raw (skip projections / `lin` convolutions,
    hourglass.py:45-48,120-135).  Exactly the maximum (a max is order-independent), never lowered, and a second
    launch into the same slots keeps the larger value."""
    from dsnt import _lib
    from dsnt._lib import ptr, call, ConvGeom, BnTail
    dev = torch.device('cuda:0')
    torch.manual_seed(1234)
    H, Cin, Cout = 32, 64, 128           # N = 64: 65536 rows, the streaming 1x1 kernel on the fp16x3 path
    g = ConvGeom(N, H, H, Cin, H, H, Cout, k, k, 1, k // 2, 1)
    x = torch.randn(N, H, H, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) * 0.1
    b = torch.randn(Cout, device=dev)
    res = torch.randn(N, H, H, Cout, device=dev) * 3 if with_res else None
    y = torch.empty(N, H, H, Cout, device=dev)
    slots = torch.zeros(64, device=dev)
    t = BnTail()
    t.amax = slots.data_ptr()
    # ... and the bound of the operand an eval-mode BatchNorm + ReLU consumer will form from y (amax_bn)
    bsc, bsh = torch.randn(Cout, device=dev), torch.randn(Cout, device=dev)
    slots_bn = torch.zeros(64, device=dev)
    t.amax_bn, t.amax_scale, t.amax_shift, t.amax_relu = slots_bn.data_ptr(), bsc.data_ptr(), bsh.data_ptr(), 1

    def launch(xx):
        if path == 'f32':
            call('dsnt_conv_fwd_ex', ptr(xx), ptr(w), ptr(b), ptr(y), None, None, 0, ptr(res) if with_res else None, None,
                 None, C.byref(g), None, C.byref(t))
        elif path == 'bf16x6':
            planes = torch.empty(3 * w.numel(), dtype=torch.bfloat16, device=dev)
            call('dsnt_split_bf16x3', ptr(w), ptr(planes), w.numel())
            call('dsnt_conv_fwd_bf16x6_ex', ptr(xx), ptr(planes), w.numel(), ptr(b), ptr(y), None, None, 0,
                 ptr(res) if with_res else None, None, None, C.byref(g), None, C.byref(t))
        else:
            wb, ab = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
            call('dsnt_amax', ptr(w), w.numel(), ptr(wb))
            call('dsnt_amax', ptr(xx), xx.numel(), ptr(ab))
            planes = torch.empty(2 * w.numel(), dtype=torch.float16, device=dev)
            call('dsnt_split_f16x2', ptr(w), ptr(planes), w.numel(), w.numel(), ptr(wb))
            call('dsnt_conv_fwd_f16x3_ex', ptr(xx), ptr(planes), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y), None, None,
                 0, ptr(res) if with_res else None, None, None, C.byref(g), None, C.byref(t))
        torch.cuda.synchronize()

    if path != 'f32' and not _lib.fn('dsnt_conv_bf16x6_ok')(C.byref(g)):
        pytest.skip('geometry not supported by the split-precision kernels')
    launch(x)
    first = float(y.abs().max())
    assert float(slots.max()) == first and first > 0
    want_bn = float(torch.relu(y.double() * bsc.double() + bsh.double()).max())
    assert abs(float(slots_bn.max()) - want_bn) <= 1e-6 * want_bn and want_bn > 0
    launch(x * 0.25)                       # a (mostly) smaller output: the bound is never lowered
    second = max(first, float(y.abs().max()))
    assert float(slots.max()) == second
    launch(x * 4.0)                        # a larger one raises it
    assert float(slots.max()) == float(y.abs().max()) > second
    # a K-split-sized launch with amax set still reports (it is routed to the tile kernel)
    if path == 'f32':
        gs = ConvGeom(1, 8, 8, Cin, 8, 8, Cout, k, k, 1, k // 2, 1)
        xs = torch.randn(1, 8, 8, Cin, device=dev)
        ys = torch.empty(1, 8, 8, Cout, device=dev)
        slots.zero_()
        call('dsnt_conv_fwd_ex', ptr(xs), ptr(w), ptr(b), ptr(ys), None, None, 0, None, None, None, C.byref(gs), None,
             C.byref(t))
        torch.cuda.synchronize()
        assert float(slots.max()) == float(ys.abs().max())


def test_stem_as_space_to_depth_convolution():
    """The 7x7 / stride 2 / pad 3 stem (hourglass.py:106) as a 4x4 / stride 1 / pad 1 convolution on the space-to-depth image
    (dsnt_s2d_input + dsnt_s2d_weights): same result as F.conv2d on the original tensors, max|image| as the operand bound,
    and the gradient gather (back = 1) is the adjoint of the filter re-packing."""
    from dsnt._lib import ptr, call, ConvGeom, BnTail
    dev = torch.device('cuda:0')
    torch.manual_seed(7)
    N, Cc, H, W, Cout = 2, 3, 32, 48, 8
    x = torch.randn(N, Cc, H, W, device=dev)
    w = torch.randn(Cout, Cc, 7, 7, device=dev) * 0.1
    b = torch.randn(Cout, device=dev)
    want = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=3)
    w_ohwi = torch.zeros(Cout, 7, 7, 4, device=dev)
    w_ohwi[..., :Cc] = w.permute(0, 2, 3, 1)
    xs = torch.full((N, H // 2 + 1, W // 2 + 1, 16), float('nan'), device=dev)
    slots = torch.zeros(64, device=dev)
    t = BnTail()
    t.amax = slots.data_ptr()
    call('dsnt_s2d_input', ptr(x), ptr(xs), N, Cc, H, W, C.byref(t))
    w2 = torch.full((Cout, 4, 4, 16), float('nan'), device=dev)
    call('dsnt_s2d_weights', ptr(w_ohwi), ptr(w2), Cout, 0)
    torch.cuda.synchronize()
    assert float(slots.max()) == float(x.abs().max())
    assert float(xs[:, 0].abs().max()) == 0 and float(xs[:, :, 0].abs().max()) == 0 and not torch.isnan(xs).any()
    assert float(w2[:, 0, :, :8].abs().max()) == 0 and float(w2[:, :, 0, 0:4].abs().max()) == 0    # the added zero row / column
    g = ConvGeom(N, H // 2 + 1, W // 2 + 1, 16, H // 2, W // 2, Cout, 4, 4, 1, 1, 1)
    y = torch.empty(N, H // 2, W // 2, Cout, device=dev)
    call('dsnt_conv_fwd', ptr(xs), ptr(w2), ptr(b), ptr(y), None, None, 0, None, None, None, C.byref(g))
    torch.cuda.synchronize()
    err = (y.permute(0, 3, 1, 2).double() - want).abs().max().item()
    assert err <= 2e-5 * want.abs().max().item(), err
    # the one-launch per-step form: re-packed filter + bound + fp16 / bf16 planes, identical to the separate launches
    n = Cout * 256
    w2b = torch.empty_like(w2)
    p16, p6, bnd = (torch.empty(2 * n, dtype=torch.float16, device=dev), torch.empty(3 * n, dtype=torch.bfloat16, device=dev),
                    torch.zeros(64, device=dev))
    call('dsnt_s2d_weights_prep', ptr(w_ohwi), ptr(w2b), ptr(p16), ptr(p6), ptr(bnd), Cout)
    q16, q6, bnd2 = torch.empty_like(p16), torch.empty_like(p6), torch.zeros(64, device=dev)
    call('dsnt_split_bf16x3', ptr(w2), ptr(q6), n)
    call('dsnt_amax', ptr(w2), n, ptr(bnd2))
    call('dsnt_split_f16x2', ptr(w2), ptr(q16), n, n, ptr(bnd2))
    torch.cuda.synchronize()
    assert torch.equal(w2b, w2) and float(bnd.max()) == float(w2.abs().max()) == float(bnd2.max())
    assert torch.equal(p16.view(torch.int16), q16.view(torch.int16)) and torch.equal(p6.view(torch.int16), q6.view(torch.int16))
    # adjoint: <repack(w), g2> == <w, gather(g2)>
    g2 = torch.randn(Cout, 4, 4, 16, device=dev)
    back = torch.full((Cout, 7, 7, 4), float('nan'), device=dev)
    call('dsnt_s2d_weights', ptr(g2), ptr(back), Cout, 1)
    torch.cuda.synchronize()
    lhs = float((w2.double() * g2.double()).sum())
    rhs = float((w_ohwi.double() * back.double()).sum())
    assert abs(lhs - rhs) <= 1e-9 * max(1.0, abs(lhs)) and not torch.isnan(back).any()


def test_pack_of_all_data_gradient_weights_in_one_launch():
    """dsnt_conv_pack_dgrad_all (the per-step re-pack of every convolution's weights for its data gradient: wd[ci][R-1-r][S-1-s][co] =
    w[co][r][s][ci], fp32 + three bf16 planes; /root/reference/src/dsnt/hourglass.py:20-25 backward) — 32 x 32 tiles through LDS —
    against the index formula, for ragged channel counts, 1x1 / 3x3 / 7x7 filters and several convolutions in one table."""
    from dsnt._lib import ptr, call
    dev = torch.device('cuda:0')
    convs = [(128, 3, 3, 128), (256, 1, 1, 128), (16, 1, 1, 256), (256, 1, 1, 16), (64, 7, 7, 4), (48, 3, 3, 80), (36, 1, 1, 20)]
    srcs, rows, off_s, off_d = [], [], 0, 0
    for (co, r, s_, ci) in convs:
        n = co * r * s_ * ci
        srcs.append(torch.randn(n, device=dev))
        rows.append([off_s, off_d, co, r, s_, ci])
        off_s += n
        off_d += (n + 7) // 8 * 8
    params = torch.cat(srcs)
    total = off_d
    out = torch.full((total,), float('nan'), device=dev)
    planes = torch.zeros(3 * total, dtype=torch.bfloat16, device=dev)
    table = torch.tensor(rows, dtype=torch.int32, device=dev)
    call('dsnt_conv_pack_dgrad_all', ptr(table), len(convs), ptr(params), ptr(out), ptr(planes), total)
    torch.cuda.synchronize()
    for (co, r, s_, ci), w, row in zip(convs, srcs, rows):
        n = co * r * s_ * ci
        want = w.view(co, r, s_, ci).flip(1, 2).permute(3, 1, 2, 0).contiguous().view(-1)
        got = out[row[1]:row[1] + n]
        assert torch.equal(got, want), (co, r, s_, ci)
        p = [planes[k * total + row[1]:k * total + row[1] + n].float() for k in range(3)]
        assert torch.equal(p[0] + p[1] + p[2], want) or (p[0] + p[1] + p[2] - want).abs().max().item() <= 1e-7 * want.abs().max().item()
        assert torch.equal(p[0], want.to(torch.bfloat16).float())
