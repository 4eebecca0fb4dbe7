"""The one-pass backward of a 1x1 convolution (csrc/bwd1.hip, dsnt_conv1x1_bwd_f16x3) against fp64: what autograd runs as
backward-data + backward-filter of /root/reference/src/dsnt/hourglass.py:20,25 (conv1 / conv3 of a Bottleneck) and, in
the `apply` mode, the BatchNorm backward of the layer behind (hourglass.py:21,36-37), through the C ABI."""
import ctypes as C

import pytest
import torch

from dsnt import synthetic

pytestmark = pytest.mark.gpu

CASES = [
    # N, H, W, Cin, Cout
    (4, 64, 64, 256, 128),     # conv1 of a Bottleneck (n = 128, two 16-column tiles per wave)
    (4, 64, 64, 128, 256),     # conv3 (n = 256, one tile per wave)
    (4, 64, 64, 128, 128),     # conv1 of a 128-wide Bottleneck
    (5, 64, 64, 256, 128),     # 640 stages over 214 workgroups: the last one is short
    (16, 32, 32, 128, 256),
    (2, 128, 128, 64, 64),     # the 128 x 128 level: four-wave workgroups, two per CU
    (2, 128, 128, 64, 128),
    (4, 64, 64, 256, 256),     # the `fc` convolutions: two column chunks (gridDim.y) filling one slab
]


def _bound(value, dev):
    return torch.full((64,), float(value), device=dev)


def _reference(x, sc, sh, mu, istd, relu, dy, w):
    """fp64: dz_out, the two BatchNorm-backward sums, dW, db for y = conv1x1(relu?(x * sc + sh)) given dL/dy."""
    z = x * sc + sh
    act = torch.relu(z) if relu else z
    dx = dy @ w                                   # [M, Cin]
    if relu:
        dx = dx * (z > 0)
    xhat = (x - mu) * istd
    return dx, dx.sum(0), (dx * xhat).sum(0), dy.t() @ act, dy.sum(0), z


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('mode,relu', [('apply', 1), ('given', 1), ('apply', 0), ('raw', 0), ('raw_acc', 0)])
def test_conv1x1_backward_in_one_pass(case, mode, relu):
    from dsnt import _lib
    from dsnt._lib import ptr, call, ConvGeom, BnBwdEpilogue, BnBwdApply
    N, H, W, Cin, Cout = case
    dev = torch.device('cuda:0')
    tag = 'b1' + '_'.join(map(str, case)) + mode
    M = N * H * W
    g = ConvGeom(N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, 1)
    assert _lib.fn('dsnt_conv1x1_bwd_ok')(C.byref(g))
    x = synthetic.tensor(tag + 'x', (M, Cin), seed=1)
    gamma = synthetic.tensor(tag + 'ga', (Cin,), seed=1, kind='uniform').abs() + 0.5
    beta = synthetic.tensor(tag + 'be', (Cin,), seed=1, scale=0.3)
    mu = x.double().mean(0).float()
    istd = (1.0 / (x.double().var(0, unbiased=False) + 1e-5).sqrt()).float()
    sc = gamma * istd
    sh = beta - mu * sc
    raw = mode.startswith('raw')           # no BatchNorm in front of the convolution: act(x) = x, dL/dx written as it is
    if raw:
        sc, sh, mu, istd = torch.ones(Cin), torch.zeros(Cin), torch.zeros(Cin), torch.zeros(Cin)
    w = synthetic.tensor(tag + 'w', (Cout, Cin), seed=2) * 0.05
    if mode == 'apply':
        # dY = y_scale (dz - c0 - (y - y_mean) y_invstd c1): the BatchNorm backward of the layer behind, folded in
        y = synthetic.tensor(tag + 'y', (M, Cout), seed=3)
        dz = synthetic.tensor(tag + 'dz', (M, Cout), seed=4) * 1e-3
        dz = dz * (synthetic.tensor(tag + 'mk', (M, Cout), seed=5) > 0)          # masked, as a data-gradient epilogue leaves it
        y_mean = y.double().mean(0).float()
        y_istd = (1.0 / (y.double().var(0, unbiased=False) + 1e-5).sqrt()).float()
        y_scale = (synthetic.tensor(tag + 'yg', (Cout,), seed=6, kind='uniform').abs() + 0.5) * y_istd
        yhat = (y.double() - y_mean.double()) * y_istd.double()
        coef = torch.stack([dz.double().mean(0), (dz.double() * yhat).mean(0)]).float()
        dy64 = y_scale.double() * (dz.double() - coef[0].double() - yhat * coef[1].double())
    else:
        dy = synthetic.tensor(tag + 'dy', (M, Cout), seed=4) * 1e-3
        dy64 = dy.double()
    ref_dz, ref_s1, ref_s2, ref_dw, ref_db, z64 = _reference(x.double(), sc.double(), sh.double(), mu.double(), istd.double(),
                                                              relu, dy64, w.double())
    # device side
    xd, scd, shd, mud, isd = (t.to(dev) for t in (x, sc, sh, mu, istd))
    wdt = w.t().contiguous().to(dev)                                  # the data gradient's filter [Cin][Cout]
    wb = torch.zeros(64, device=dev)
    call('dsnt_amax', ptr(wdt), wdt.numel(), ptr(wb))
    planes = torch.empty(2 * wdt.numel(), dtype=torch.float16, device=dev)
    call('dsnt_split_f16x2', ptr(wdt), ptr(planes), wdt.numel(), wdt.numel(), ptr(wb))
    act_max = (torch.relu(z64) if relu else z64).abs().max().item()
    ab = _bound(act_max * 3.0, dev)
    gb = _bound(dy64.abs().max().item() * 5.0, dev)                  # loose, as the analytic bound of the engine is
    xs = BnBwdEpilogue(ptr(xd), None, None, None, None, 0) if raw else BnBwdEpilogue(ptr(xd), ptr(scd), ptr(shd), ptr(mud), ptr(isd), relu)
    flags = 1 if mode == 'raw_acc' else 0
    prev = synthetic.tensor(tag + 'pv', (M, Cin), seed=7) * 1e-3 if flags else None
    if mode == 'apply':
        yd, dzd = y.to(dev), dz.to(dev)
        ysd, ymd, yid, cfd = y_scale.to(dev), y_mean.to(dev), y_istd.to(dev), coef.contiguous().to(dev)
        ap = BnBwdApply(ptr(yd), ptr(ysd), ptr(ymd), ptr(yid), ptr(cfd))
        dyd, apref = dzd, C.byref(ap)
    else:
        dyd, apref = dy.to(dev), None
    splits = _lib.fn('dsnt_conv1x1_bwd_splits')(C.byref(g), 0)
    nws = _lib.fn('dsnt_conv1x1_bwd_ws_floats')(C.byref(g), 0)
    assert nws == splits * Cout * (Cin + 1) and 0 < splits <= 512
    ws = torch.full((nws,), float('nan'), device=dev)
    stats = torch.full((splits, 2, Cin), float('nan'), device=dev)
    dz_out = prev.to(dev) if flags else torch.full((M, Cin), float('nan'), device=dev)
    if flags:
        ref_dz = ref_dz + prev.double()
    amax = torch.zeros(64, device=dev)
    call('dsnt_conv1x1_bwd_f16x3', C.byref(xs), ptr(dyd), apref, ptr(planes), wdt.numel(), ptr(wb), ptr(ab), ptr(gb),
         ptr(dz_out), None if raw else ptr(stats), ptr(ws), ptr(amax), flags, C.byref(g))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ws).all()) and bool(torch.isfinite(dz_out).all())
    assert raw or bool(torch.isfinite(stats).all())
    # data gradient (elements whose pre-activation sits at the kink may take either side)
    got = dz_out.cpu().double()
    sure = (z64.abs() > 1e-5) if relu else torch.ones_like(z64, dtype=torch.bool)
    scale = ref_dz.abs().max().item()
    e = ((got - ref_dz) * sure).abs().max().item()
    assert e <= 2e-6 * scale, (e, scale)
    assert abs(amax.max().item() - dz_out.abs().max().item()) == 0.0
    # the BatchNorm-backward sums: one partial row per workgroup
    if not raw:
        s = stats.cpu().double().sum(0)
        unsure1 = ((got.abs() + ref_dz.abs()) * ~sure).sum(0)
        assert ((s[0] - ref_s1).abs() <= 1e-5 * ref_dz.abs().sum(0) + unsure1 + 1e-12).all()
        xhat = ((x.double() - mu.double()) * istd.double()).abs()
        assert ((s[1] - ref_s2).abs() <= 1e-5 * (ref_dz.abs() * xhat).sum(0) + unsure1 * xhat.max() + 1e-12).all()
    # weight and bias gradient through the table-driven slab reduction
    dw, db = torch.zeros(Cout, Cin, device=dev), torch.zeros(Cout, device=dev)
    table = torch.tensor([[ws.data_ptr(), dw.data_ptr(), db.data_ptr(), splits, Cout * Cin, Cout, 0]],
                         dtype=torch.int64).to(dev)
    call('dsnt_wgrad_reduce_all', ptr(table), 1, (Cout * Cin // 4 + (Cout + 3) // 4 + 63) // 64)
    ew = (dw.cpu().double() - ref_dw).abs().max().item()
    assert ew <= 3e-6 * ref_dw.abs().max().item(), (ew, ref_dw.abs().max().item())
    eb = (db.cpu().double() - ref_db).abs().max().item()
    assert eb <= 3e-6 * max(ref_db.abs().max().item(), dy64.abs().sum(0).max().item() * 1e-2), eb
    # a second launch writes the same bits (fixed summation order, no atomics on the results)
    ws2, stats2 = torch.empty_like(ws), torch.empty_like(stats)
    dz2 = prev.to(dev) if flags else torch.empty_like(dz_out)
    call('dsnt_conv1x1_bwd_f16x3', C.byref(xs), ptr(dyd), apref, ptr(planes), wdt.numel(), ptr(wb), ptr(ab), ptr(gb),
         ptr(dz2), None if raw else ptr(stats2), ptr(ws2), None, flags, C.byref(g))
    assert torch.equal(ws, ws2) and (raw or torch.equal(stats, stats2)) and torch.equal(dz_out, dz2)


def test_conv1x1_backward_refusals():
    from dsnt import _lib
    from dsnt._lib import ConvGeom
    ok = _lib.fn('dsnt_conv1x1_bwd_ok')
    assert not ok(C.byref(ConvGeom(4, 64, 64, 256, 64, 64, 16, 1, 1, 1, 0, 1)))       # 256 -> 16 (score): not built
    assert not ok(C.byref(ConvGeom(4, 64, 64, 128, 64, 64, 128, 3, 3, 1, 1, 1)))      # 3x3
    assert not ok(C.byref(ConvGeom(2, 16, 16, 256, 16, 16, 128, 1, 1, 1, 0, 1)))      # 512 rows
    g = ConvGeom(2, 16, 16, 256, 16, 16, 128, 1, 1, 1, 0, 1)
    rc = _lib.fn('dsnt_conv1x1_bwd_f16x3')(None, None, None, None, 0, None, None, None, None, None, None, None,
                                           0, C.byref(g), None)
    assert rc == 3


FWD_CASES = [
    # N, H, W, Cin, Cout
    (4, 64, 64, 256, 128),     # conv1 of a Bottleneck
    (4, 64, 64, 128, 256),     # conv3 (+ residual)
    (4, 64, 64, 128, 128),
    (2, 128, 128, 64, 64),     # four-wave workgroups
    (2, 128, 128, 64, 128),
    (4, 64, 64, 256, 256),     # two column chunks
    (5, 64, 64, 128, 256),     # 640 stages over 214 workgroups: the last one is short
]


@pytest.mark.parametrize('case', FWD_CASES)
@pytest.mark.parametrize('pro,res', [(True, True), (True, False), (False, True), (False, False)])
def test_conv1x1_forward_on_the_lds_staged_streaming_kernel(case, pro, res):
    """dsnt_conv1x1_fwd_f16x3 (csrc/fwd1.hip: conv1 / conv3 / shortcut / `fc` forward, hourglass.py:20,25,44-48,146-153) against
    fp64 and against dsnt_conv_fwd_f16x3_ex (the first streaming kernel) on the same operands: output, the one-row-per-workgroup
    statistics (summed: the same two sums), both operand-bound outputs, bit-reproducible; DSNT_CONV_SHARE_CHIP changes no value."""
    from dsnt import _lib
    from dsnt._lib import ptr, call, ConvGeom, BnTail
    N, H, W, Cin, Cout = case
    dev = torch.device('cuda:0')
    tag = 'f1' + '_'.join(map(str, case))
    M = N * H * W
    g = ConvGeom(N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, 1)
    assert _lib.fn('dsnt_conv1x1_fwd_ok')(C.byref(g))
    x = synthetic.tensor(tag + 'x', (M, Cin), seed=1)
    sc = synthetic.tensor(tag + 's', (Cin,), seed=1, kind='uniform').abs() + 0.5
    sh = synthetic.tensor(tag + 'h', (Cin,), seed=1, scale=0.3)
    w = synthetic.tensor(tag + 'w', (Cout, Cin), seed=2) * 0.05
    b = synthetic.tensor(tag + 'b', (Cout,), seed=2, scale=0.1)
    r = synthetic.tensor(tag + 'r', (M, Cout), seed=3)
    act = torch.relu(x.double() * sc.double() + sh.double()) if pro else x.double()
    ref = act @ w.double().t() + b.double() + (r.double() if res else 0.0)
    xd, scd, shd, wd, bd, rd = (t.to(dev) for t in (x, sc, sh, w, b, r))
    wb = torch.zeros(64, device=dev)
    call('dsnt_amax', ptr(wd), wd.numel(), ptr(wb))
    planes = torch.empty(2 * wd.numel(), dtype=torch.float16, device=dev)
    call('dsnt_split_f16x2', ptr(wd), ptr(planes), wd.numel(), wd.numel(), ptr(wb))
    ab = torch.full((64,), act.abs().max().item() * 3.0, device=dev)
    asc = (synthetic.tensor(tag + 'as', (Cout,), seed=4, kind='uniform').abs() + 0.5).to(dev)
    ash = synthetic.tensor(tag + 'ah', (Cout,), seed=4, scale=0.2).to(dev)

    def run(flags, name='dsnt_conv1x1_fwd_f16x3'):
        y = torch.full((M, Cout), float('nan'), device=dev)
        amax, amax_bn = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
        tl = BnTail()
        tl.amax, tl.amax_bn, tl.amax_scale, tl.amax_shift, tl.amax_relu = (amax.data_ptr(), amax_bn.data_ptr(), asc.data_ptr(),
                                                                            ash.data_ptr(), 1)
        common = (ptr(xd), ptr(planes), wd.numel(), ptr(wb), ptr(ab), ptr(bd), ptr(y), ptr(scd) if pro else None,
                  ptr(shd) if pro else None, 1 | flags, ptr(rd) if res else None)
        if name == 'dsnt_conv1x1_fwd_f16x3':
            rows = _lib.fn('dsnt_conv1x1_fwd_stats_rows')(C.byref(g), flags)
            stats = torch.full((rows, 2, Cout), float('nan'), device=dev)
            call(name, *common, ptr(stats), C.byref(g), C.byref(tl))
        else:
            rows = (M + 127) // 128
            stats = torch.full((rows, 2, Cout), float('nan'), device=dev)
            call(name, *common, None, ptr(stats), C.byref(g), None, C.byref(tl))
        torch.cuda.synchronize()
        return y, stats, amax, amax_bn, rows

    y, stats, amax, amax_bn, rows = run(0)
    assert 0 < rows <= 512 and bool(torch.isfinite(y).all()) and bool(torch.isfinite(stats).all())
    scale = ref.abs().max().item()
    e = (y.cpu().double() - ref).abs().max().item()
    assert e <= 2e-6 * scale, (e, scale)
    s = stats.cpu().double().sum(0)
    assert ((s[0] - ref.sum(0)).abs() <= 1e-5 * ref.abs().sum(0)).all()
    assert ((s[1] - (ref * ref).sum(0)).abs() <= 1e-5 * (ref * ref).sum(0)).all()
    assert float(amax.max()) == float(y.abs().max())
    want_bn = torch.relu(y * asc + ash).abs().max()
    assert abs(float(amax_bn.max()) - float(want_bn)) <= 1e-6 * float(want_bn)
    # the first streaming kernel on the same operands
    y0, stats0, amax0, _, _ = run(0, 'dsnt_conv_fwd_f16x3_ex')
    assert (y - y0).abs().max().item() <= 2e-6 * scale
    assert ((stats.double().sum(0) - stats0.double().sum(0)).abs() <= 1e-5 * stats0.double().abs().sum(0) + 1e-6).all()
    # bit-reproducible, and the share flag (fewer workgroups, other statistics rows) changes no output value
    y2, stats2, _, _, _ = run(0)
    assert torch.equal(y, y2) and torch.equal(stats, stats2)
    y3, stats3, _, _, rows3 = run(2)
    assert torch.equal(y, y3) and rows3 <= rows
    assert ((stats3.double().sum(0) - stats.double().sum(0)).abs() <= 1e-5 * stats.double().abs().sum(0) + 1e-6).all()


STEM_CASES = [
    # N, Ho (= Wo): the space-to-depth image is [N][Ho + 1][Wo + 1][16]
    (2, 128),      # the 256-pixel input of every BASELINE configuration: 256 tiles
    (3, 64),       # the 128-pixel goldens
    (9, 128),      # 1152 tiles on 512 persistent workgroups: 2-3 tiles each, ragged
    (1, 32),       # 8 tiles
]


@pytest.mark.parametrize('N,Ho', STEM_CASES)
@pytest.mark.parametrize('with_bias', [True, False])
def test_stem_forward_on_its_halo_kernel(N, Ho, with_bias):
    """dsnt_stem4_fwd_f16x3 (csrc/stem4.hip): the stem of /root/reference/src/dsnt/hourglass.py:157 (`conv1`, 7x7 / stride 2 / pad 3)
    in its space-to-depth form — a 4x4 / stride 1 / pad 1 convolution of the 16-channel image, 64 output channels — against the
    tiled fp16x3 kernel on the same operands (same K order, same products: bit-identical outputs), against torch in fp64 (the fp32
    bar), with its statistics (one row per workgroup: the column sums of y and y^2) and the bound of its output."""
    from dsnt import _lib
    from dsnt._lib import ptr, ConvGeom, BnTail
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    Hi = Ho + 1
    g = ConvGeom(N, Hi, Hi, 16, Ho, Ho, 64, 4, 4, 1, 1, 1)
    assert _lib.fn('dsnt_stem4_fwd_ok')(C.byref(g)) == 1
    tag = 'stem%d_%d' % (N, Ho)
    x = synthetic.tensor(tag + 'x', (N, Hi, Hi, 16), seed=1).to(dev)
    w = (synthetic.tensor(tag + 'w', (64, 4, 4, 16), seed=2) * 0.1).to(dev)
    b = synthetic.tensor(tag + 'b', (64,), seed=3).to(dev) if with_bias else None
    wb, ab = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    planes = torch.empty(2 * w.numel(), dtype=torch.float16, device=dev)
    assert _lib.fn('dsnt_amax')(ptr(w), w.numel(), ptr(wb), st) == 0
    assert _lib.fn('dsnt_amax')(ptr(x), x.numel(), ptr(ab), st) == 0
    assert _lib.fn('dsnt_split_f16x2')(ptr(w), ptr(planes), w.numel(), w.numel(), ptr(wb), st) == 0
    M = N * Ho * Ho
    rows = _lib.fn('dsnt_stem4_fwd_stats_rows')(C.byref(g))
    assert 0 < rows <= N * (Ho // 4) * (Ho // 32)
    y = torch.full((N, Ho, Ho, 64), float('nan'), device=dev)
    part = torch.full((rows, 2, 64), float('nan'), device=dev)
    am = torch.zeros(64, device=dev)
    tail = BnTail()
    tail.amax = am.data_ptr()
    assert _lib.fn('dsnt_stem4_fwd_f16x3')(ptr(x), ptr(planes), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y), ptr(part), C.byref(g),
                                           C.byref(tail), st) == 0
    # the tiled kernel on the same planes and bounds
    y_t = torch.empty_like(y)
    part_t = torch.empty((M + 127) // 128, 2, 64, device=dev)
    assert _lib.fn('dsnt_conv_fwd_f16x3_ex')(ptr(x), ptr(planes), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y_t), None, None, 0, None, None,
                                             ptr(part_t), C.byref(g), None, None, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, y_t)
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2),
                                     None if b is None else b.double(), stride=1, padding=1).permute(0, 2, 3, 1)
    assert ref.shape == y.shape
    assert (y.double() - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())
    s1, s2 = part[:, 0].double().sum(0), part[:, 1].double().sum(0)
    yd = y.double().reshape(-1, 64)
    assert (s1 - yd.sum(0)).abs().max().item() <= 1e-5 * max(1.0, yd.abs().sum(0).max().item())
    assert (s2 - (yd * yd).sum(0)).abs().max().item() <= 1e-5 * (yd * yd).sum(0).max().item()
    assert am.max().item() == y.abs().max().item()
    # without statistics / bound; refusals
    y2 = torch.empty_like(y)
    assert _lib.fn('dsnt_stem4_fwd_f16x3')(ptr(x), ptr(planes), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y2), None, C.byref(g), None, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(y2, y)
    bad = ConvGeom(N, Hi, Hi, 16, Ho, Ho, 128, 4, 4, 1, 1, 1)
    assert _lib.fn('dsnt_stem4_fwd_ok')(C.byref(bad)) == 0
    bad3 = ConvGeom(N, Ho, Ho, 16, Ho, Ho, 64, 3, 3, 1, 1, 1)
    assert _lib.fn('dsnt_stem4_fwd_ok')(C.byref(bad3)) == 0
    assert _lib.fn('dsnt_stem4_fwd_f16x3')(ptr(x), ptr(planes), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y2), None, C.byref(bad3), None, st) != 0


@pytest.mark.parametrize('N,Ho', STEM_CASES)
def test_stem_weight_gradient_on_its_own_kernel(N, Ho):
    """The weight gradient of the same convolution (csrc/stem4.hip: stem4_wgrad_kernel, reached through dsnt_conv_wgrad_f16x3): both
    operands read transposed from pixel-major LDS images, one slab and one bias partial per workgroup — against torch in fp64 (the
    weight-gradient bar of the other fp16x3 kernels), with and without the in-call reduction, accumulating, and the refusal of a
    BatchNorm prologue (the stem's operand is the raw image)."""
    from dsnt import _lib
    from dsnt._lib import ptr, ConvGeom
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    Hi = Ho + 1
    g = ConvGeom(N, Hi, Hi, 16, Ho, Ho, 64, 4, 4, 1, 1, 1)
    tag = 'stemw%d_%d' % (N, Ho)
    x = synthetic.tensor(tag + 'x', (N, Hi, Hi, 16), seed=1).to(dev)
    gy = (synthetic.tensor(tag + 'g', (N, Ho, Ho, 64), seed=2) * 1e-2).to(dev)
    ab, gb = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    assert _lib.fn('dsnt_amax')(ptr(x), x.numel(), ptr(ab), st) == 0
    assert _lib.fn('dsnt_amax')(ptr(gy), gy.numel(), ptr(gb), st) == 0
    splits = _lib.fn('dsnt_conv_wgrad_f16x3_splits')(C.byref(g), 0)
    nws = _lib.fn('dsnt_conv_wgrad_f16x3_ws_floats')(C.byref(g), 0)
    assert 0 < splits <= N * (Ho // 4) * (Ho // 32) and nws == splits * (64 * 256 + 64)
    assert _lib.fn('dsnt_conv_wgrad_ws_floats')(C.byref(g)) >= nws
    ws = torch.full((nws,), float('nan'), device=dev)
    dw, db = torch.empty(64, 4, 4, 16, device=dev), torch.empty(64, device=dev)
    wg = _lib.fn('dsnt_conv_wgrad_f16x3')
    assert wg(ptr(x), None, None, 0, ptr(gy), ptr(ws), ptr(dw), ptr(db), 0, ptr(ab), ptr(gb), C.byref(g), st) == 0
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.double().permute(0, 3, 1, 2), (64, 16, 4, 4), gy.double().permute(0, 3, 1, 2),
                                      stride=1, padding=1).permute(0, 2, 3, 1)
    refb = gy.double().sum((0, 1, 2))
    assert not torch.isnan(ws).any()
    assert (dw.double() - ref).abs().max().item() <= 4e-6 * ref.abs().max().item()
    assert (db.double() - refb).abs().max().item() <= 4e-6 * max(refb.abs().max().item(), gy.abs().sum().item() * 1e-3)
    # the slabs alone add up to the same gradient (what the engine's one reduction per bucket does), and accumulate adds
    slabs = ws[:splits * 64 * 256].view(splits, 64, 4, 4, 16).double().sum(0)
    assert (slabs - ref).abs().max().item() <= 4e-6 * ref.abs().max().item()
    assert wg(ptr(x), None, None, 0, ptr(gy), ptr(ws), ptr(dw), ptr(db), 1, ptr(ab), ptr(gb), C.byref(g), st) == 0
    torch.cuda.synchronize()
    assert (dw.double() - 2 * ref).abs().max().item() <= 8e-6 * ref.abs().max().item()
    sc, sh = torch.ones(16, device=dev), torch.zeros(16, device=dev)
    assert wg(ptr(x), ptr(sc), ptr(sh), 0, ptr(gy), ptr(ws), None, None, 0, ptr(ab), ptr(gb), C.byref(g), st) != 0
