"""The skinny 1x1 convolutions of the intermediate supervision (csrc/skinny.hip) against torch fp64.

/root/reference/src/dsnt/hourglass.py:166-175: `score` (256 -> 16) produces a stack's heat-map logits, `score_` (16 -> 256) feeds
them back into the trunk (`x = x + fc_ + score_`).  With 16 channels on one side these convolutions and their data gradients are
streaming passes; round 6 runs the K = 16 forms (the `score_` forward with two residual inputs + the next BatchNorm's statistics,
and the data gradient of `score` with the BatchNorm-backward epilogue) on the vector ALU in fp32 instead of on 128-wide
matrix-core tiles.  Bars: the error against fp64 is at most 4 x that of torch's own fp32 convolution (the bar of every
split-precision kernel in tests/test_conv_gpu.py), statistics rows sum to the tensor's sums, the operand bounds cover the output,
and the launch is the one the kernel census of the models expects (entry point and contract unchanged)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from dsnt import synthetic

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _operands(N, H, Cin, Cout, f16):
    from dsnt._lib import ptr, call
    x = synthetic.tensor('skx%d' % Cin, (N, H, H, Cin), seed=3)
    w = synthetic.tensor('skw%d' % Cin, (Cout, Cin), seed=3, scale=(2.0 / Cin) ** 0.5)
    xd, wd = x.to(DEV), w.to(DEV)
    if f16:
        wb = torch.zeros(64, device=DEV)
        call('dsnt_amax', ptr(wd), wd.numel(), ptr(wb))
        planes = torch.empty(2 * wd.numel(), dtype=torch.float16, device=DEV)
        call('dsnt_split_f16x2', ptr(wd), ptr(planes), wd.numel(), wd.numel(), ptr(wb))
        ab = torch.full((64,), float(x.abs().max()) * 2.0, device=DEV)
        return x, w, xd, planes, wb, ab
    planes = torch.empty(3 * wd.numel(), dtype=torch.bfloat16, device=DEV)
    call('dsnt_split_bf16x3', ptr(wd), ptr(planes), wd.numel())
    return x, w, xd, planes, None, None


@pytest.mark.parametrize('f16', [True, False])
@pytest.mark.parametrize('nres', [0, 2])
def test_score_reinjection_forward_16_to_256(f16, nres):
    """`score_`: y = x W^T + b (+ res1 + res2), statistics rows of y per 128 rows, max |y| and max |relu(y s + h)| for the consumers."""
    from dsnt._lib import ptr, call, ConvGeom, BnTail
    N, H, Cin, Cout = 4, 32, 16, 256
    x, w, xd, planes, wb, ab = _operands(N, H, Cin, Cout, f16)
    b = synthetic.tensor('skb', (Cout,), seed=3, scale=0.1)
    r1, r2 = synthetic.tensor('skr1', (N, H, H, Cout), seed=3), synthetic.tensor('skr2', (N, H, H, Cout), seed=4)
    ts, th = synthetic.tensor('skts', (Cout,), seed=5, kind='uniform').abs() + 0.5, synthetic.tensor('skth', (Cout,), seed=6, scale=0.3)
    g = ConvGeom(N, H, H, Cin, H, H, Cout, 1, 1, 1, 0, 1)
    M = N * H * H
    y = torch.empty(N, H, H, Cout, device=DEV)
    stats = torch.zeros(M // 128, 2, Cout, device=DEV)
    amax, amax_bn = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    tsd, thd = ts.to(DEV), th.to(DEV)
    tail = BnTail()
    tail.amax, tail.amax_bn, tail.amax_scale, tail.amax_shift, tail.amax_relu = amax.data_ptr(), amax_bn.data_ptr(), tsd.data_ptr(), thd.data_ptr(), 1
    bd, r1d, r2d = b.to(DEV), r1.to(DEV), r2.to(DEV)
    res = (ptr(r1d), ptr(r2d)) if nres else (None, None)
    if f16:
        call('dsnt_conv_fwd_f16x3_ex', ptr(xd), ptr(planes), w.numel(), ptr(wb), ptr(ab), ptr(bd), ptr(y), None, None, 0,
             res[0], res[1], ptr(stats), C.byref(g), None, C.byref(tail))
    else:
        call('dsnt_conv_fwd_bf16x6_ex', ptr(xd), ptr(planes), w.numel(), ptr(bd), ptr(y), None, None, 0,
             res[0], res[1], ptr(stats), C.byref(g), None, C.byref(tail))
    x2 = x.reshape(M, Cin)
    extra = (r1 + r2).reshape(M, Cout) if nres else torch.zeros(M, Cout)
    y64 = x2.double() @ w.double().t() + b.double() + extra.double()
    y32 = x2 @ w.t() + b + extra
    got = y.cpu().reshape(M, Cout)
    scale = y64.abs().max().item()
    err, err32 = (got.double() - y64).abs().max().item(), (y32.double() - y64).abs().max().item()
    assert err <= max(4 * err32, 2e-6 * scale), (err, err32)
    s = stats.cpu().double().sum(0)
    assert (s[0] - y64.sum(0)).abs().max().item() <= 1e-5 * M ** 0.5 * scale
    assert (s[1] - (y64 * y64).sum(0)).abs().max().item() <= 1e-5 * (y64 * y64).sum(0).max().item()
    assert amax.max().item() == got.abs().max().item()
    want_bn = torch.relu(got * ts + th).abs().max().item()
    assert abs(amax_bn.max().item() - want_bn) <= 1e-6 * want_bn


@pytest.mark.parametrize('f16', [True, False])
@pytest.mark.parametrize('relu', [1, 0])
def test_score_data_gradient_16_to_256_with_batchnorm_backward_epilogue(f16, relu):
    """The data gradient of `score` (dY [M, 16] -> the gradient behind the BatchNorm + ReLU of the `fc` block in front of it):
    dz = (dY W) * [bn(x) > 0] written, and the two per-channel sums of the BatchNorm backward per 128 rows."""
    from dsnt._lib import ptr, call, ConvGeom, BnBwdEpilogue, BnTail
    N, H, Cin, Cout = 4, 32, 16, 256
    gy, wd_, gyd, planes, wb, ab = _operands(N, H, Cin, Cout, f16)
    gy = gy * 1e-3
    gyd = gy.to(DEV)
    if f16:
        ab.fill_(float(gy.abs().max()) * 2.0)
    xbn = synthetic.tensor('skxbn', (N, H, H, Cout), seed=7)
    sc, sh = synthetic.tensor('sksc', (Cout,), seed=8, kind='uniform').abs() + 0.5, synthetic.tensor('sksh', (Cout,), seed=9, scale=0.3)
    mu, isd = synthetic.tensor('skmu', (Cout,), seed=10, scale=0.1), synthetic.tensor('skis', (Cout,), seed=11, kind='uniform').abs() + 0.5
    g = ConvGeom(N, H, H, Cin, H, H, Cout, 1, 1, 1, 0, 1)
    M = N * H * H
    dz = torch.empty(N, H, H, Cout, device=DEV)
    stats = torch.zeros(M // 128, 2, Cout, device=DEV)
    amax = torch.zeros(64, device=DEV)
    tail = BnTail()
    tail.amax = amax.data_ptr()
    xd, scd, shd, mud, isdd = xbn.to(DEV), sc.to(DEV), sh.to(DEV), mu.to(DEV), isd.to(DEV)
    bnb = BnBwdEpilogue(ptr(xd), ptr(scd), ptr(shd), ptr(mud), ptr(isdd), relu)
    if f16:
        call('dsnt_conv_fwd_f16x3_ex', ptr(gyd), ptr(planes), wd_.numel(), ptr(wb), ptr(ab), None, ptr(dz), None, None, 0,
             None, None, ptr(stats), C.byref(g), C.byref(bnb), C.byref(tail))
    else:
        call('dsnt_conv_fwd_bf16x6_ex', ptr(gyd), ptr(planes), wd_.numel(), None, ptr(dz), None, None, 0,
             None, None, ptr(stats), C.byref(g), C.byref(bnb), C.byref(tail))
    x2 = xbn.reshape(M, Cout)
    raw64 = gy.reshape(M, Cin).double() @ wd_.double().t()
    mask = ((x2 * sc + sh) > 0) if relu else torch.ones(M, Cout, dtype=torch.bool)
    dz64 = raw64 * mask
    got = dz.cpu().reshape(M, Cout)
    scale = raw64.abs().max().item()
    err32 = ((gy.reshape(M, Cin) @ wd_.t()).double() - raw64).abs().max().item()
    assert (got.double() - dz64).abs().max().item() <= max(4 * err32, 2e-6 * scale)
    xhat = (x2.double() - mu.double()) * isd.double()
    s = stats.cpu().double().sum(0)
    assert (s[0] - dz64.sum(0)).abs().max().item() <= 1e-5 * M ** 0.5 * scale
    assert (s[1] - (dz64 * xhat).sum(0)).abs().max().item() <= 1e-5 * M ** 0.5 * scale * xhat.abs().max().item()
    assert amax.max().item() == got.abs().max().item()
