"""ResNet backbone on the HIP path: new kernels against torch CPU ops, and
`ResNetHumanPoseModel` end to end (coords bar 1e-4, loss, every parameter gradient) against the oracle,
whose ResNet wrapper is pinned to the reference's in tests/test_oracle_vs_reference.py."""
import contextlib
import os

import pytest
import torch
import torch.nn.functional as F

from dsnt import synthetic

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize('N,Cc,H,W', [(2, 64, 16, 16), (1, 8, 7, 9), (3, 4, 1, 5), (2, 16, 112, 112)])
def test_maxpool3s2(N, Cc, H, W):
    from dsnt._lib import ptr, call
    x = synthetic.tensor('mp3', (N, Cc, H, W), seed=4).requires_grad_()
    y_ref = F.max_pool2d(x, 3, stride=2, padding=1)
    go = synthetic.tensor('mp3g', tuple(y_ref.shape), seed=5)
    y_ref.backward(go)
    Ho, Wo = y_ref.shape[-2:]
    xd = _nhwc(x.detach()).to(DEV)
    y = torch.empty(N, Ho, Wo, Cc, device=DEV)
    idx = torch.empty(N, Ho, Wo, Cc, dtype=torch.uint8, device=DEV)
    call('dsnt_maxpool3s2_fwd', ptr(xd), ptr(y), ptr(idx), N, H, W, Cc)
    assert torch.equal(y.cpu().permute(0, 3, 1, 2), y_ref.detach())
    god = _nhwc(go).to(DEV)
    dx = torch.full((N, H, W, Cc), 7.0, device=DEV)
    call('dsnt_maxpool3s2_bwd', ptr(god), ptr(idx), ptr(dx), 0, N, H, W, Cc)
    assert (dx.cpu().permute(0, 3, 1, 2) - x.grad).abs().max().item() <= 1e-6
    call('dsnt_maxpool3s2_bwd', ptr(god), ptr(idx), ptr(dx), 1, N, H, W, Cc)       # accumulate
    assert (dx.cpu().permute(0, 3, 1, 2) - 2 * x.grad).abs().max().item() <= 2e-6


def test_block_tail_and_zero_insert():
    from dsnt._lib import ptr, call
    M, Cc = 2 * 5 * 7, 32
    x = synthetic.tensor('tx', (M, Cc), seed=1).to(DEV)
    r = synthetic.tensor('tr', (M, Cc), seed=2).to(DEV)
    sc = (synthetic.tensor('ts', (Cc,), seed=3, kind='uniform').abs() + 0.5).to(DEV)
    sh = synthetic.tensor('th', (Cc,), seed=4, scale=0.3).to(DEV)
    y = torch.empty_like(x)
    for relu in (1, 0):
        call('dsnt_bn_add_act_fwd', ptr(x), ptr(sc), ptr(sh), ptr(r), relu, ptr(y), M, Cc)
        want = torch.addcmul(sh, x, sc) + r          # fma(x, sc, sh) + r
        want = want.clamp_min(0) if relu else want
        assert (y - want).abs().max().item() <= 1e-6
    g = synthetic.tensor('tg', (M, Cc), seed=5).to(DEV)
    dz = torch.empty_like(g)
    call('dsnt_relu_bwd', ptr(g), ptr(y.clamp_min(0)), ptr(dz), M * Cc)
    assert torch.equal(dz, torch.where(y > 0, g, torch.zeros_like(g)))
    # ... and with the BatchNorm's two reductions in the same pass (dsnt_bn_add_act_bwd_reduce): the same dz, bit for bit, and
    # the tile sums of dsnt_bn_act_bwd_reduce over it
    mu = synthetic.tensor('tm', (Cc,), seed=7, scale=0.2).to(DEV)
    inv = (synthetic.tensor('ti', (Cc,), seed=8, kind='uniform').abs() + 0.5).to(DEV)
    tiles = (M + 127) // 128
    for relu in (1, 0):
        dz2, part, part_ref = torch.empty_like(g), torch.empty(tiles, 2, Cc, device=DEV), torch.empty(tiles, 2, Cc, device=DEV)
        call('dsnt_bn_add_act_bwd_reduce', ptr(g), ptr(y), ptr(x), ptr(mu), ptr(inv), relu, ptr(dz2), ptr(part), M, Cc)
        want_dz = dz if relu else g
        assert torch.equal(dz2, want_dz)
        call('dsnt_bn_act_bwd_reduce', ptr(want_dz), ptr(x), ptr(sc), ptr(sh), ptr(mu), ptr(inv), 0, ptr(part_ref), M, Cc)
        assert torch.equal(part, part_ref)
    # zero stuffing == the scatter half of conv_transpose2d
    N, Ho, Wo, s = 2, 3, 4, 2
    dy = synthetic.tensor('zi', (N, Ho, Wo, 8), seed=6).to(DEV)
    Hs, Ws = 7, 8                                    # (Ho-1)*s+1 = 5 and 7, plus slack rows/cols of zeros
    out = torch.full((N, Hs, Ws, 8), 3.0, device=DEV)
    call('dsnt_zero_insert', ptr(dy), ptr(out), N, Ho, Wo, 8, Hs, Ws, s)
    want = torch.zeros(N, Hs, Ws, 8, device=DEV)
    want[:, 0:(Ho - 1) * s + 1:s, 0:(Wo - 1) * s + 1:s] = dy
    assert torch.equal(out, want)
    from dsnt import _lib
    assert _lib.fn('dsnt_zero_insert')(ptr(dy), ptr(out), N, Ho, Wo, 8, 4, 8, s, None) != 0   # too small


STRIDED_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil
    (2, 16, 16, 64, 128, 3, 2, 1, 1),      # layerN[0].conv1 of a BasicBlock / .conv2 of a Bottleneck
    (2, 16, 16, 64, 128, 1, 2, 0, 1),      # downsample[0]: three of the four phases have no tap
    (2, 32, 32, 4, 64, 7, 2, 3, 1),        # the stem (image gradient): 16 / 12 / 12 / 9 taps, 4 of 32 columns used
    (1, 15, 13, 32, 64, 3, 2, 1, 1),       # odd sizes: the phases have different pixel counts
    (3, 9, 9, 80, 48, 3, 2, 1, 1),         # ragged column tiles (80 = 2.5 x 32), Cout not a multiple of 32
    (1, 12, 12, 16, 32, 3, 3, 1, 1),       # stride 3
    (1, 16, 16, 16, 32, 3, 2, 2, 2),       # stride 2 with dilation 2: only the even taps of even pixels ... contribute
    (1, 17, 16, 8, 16, 5, 4, 2, 1),        # stride 4, 5x5
    (2, 8, 8, 256, 512, 3, 2, 1, 1),       # layer4 of resnet34 at 256 px: 32 rows per phase, K = 512 ... 2048
]


@pytest.mark.parametrize('case', STRIDED_CASES)
def test_native_strided_data_gradient(case):
    """dsnt_conv_dgrad_strided (csrc/dgrad_up.hip) against torch autograd on the CPU (fp32 kernel: 2e-5 of the scale, as
    the other fp32 convolution tests), its three epilogues, and against the path it replaces (dsnt_zero_insert + the
    stride-1 kernel) where that one applies."""
    import ctypes as C
    from dsnt import _lib
    from dsnt._lib import ptr, call, ConvGeom, BnBwdEpilogue, BnTail
    N, H, W, Cin, Cout, k, s, pad, dil = case
    tag = 'sd' + '_'.join(map(str, case))
    x = synthetic.tensor(tag + 'x', (N, Cin, H, W), seed=1).requires_grad_()
    w = synthetic.tensor(tag + 'w', (Cout, Cin, k, k), seed=1, scale=(2.0 / (Cin * k * k)) ** 0.5)
    y = F.conv2d(x, w, None, stride=s, padding=pad, dilation=dil)
    Ho, Wo = y.shape[-2:]
    gy = synthetic.tensor(tag + 'g', tuple(y.shape), seed=2)
    y.backward(gy)
    want = _nhwc(x.grad)
    scale = want.abs().max().item()
    g = ConvGeom(N, H, W, Cin, Ho, Wo, Cout, k, k, s, pad, dil)
    assert _lib.fn('dsnt_conv_dgrad_strided_ok')(C.byref(g)) == 1
    gyd = _nhwc(gy).to(DEV)
    wd = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    wdg = torch.empty(Cin, k, k, Cout, device=DEV)
    call('dsnt_conv_pack_dgrad', ptr(wd), ptr(wdg), Cout, k, k, Cin)
    dx = torch.full((N, H, W, Cin), 7.0, device=DEV)
    amax = torch.zeros(64, device=DEV)
    tl = BnTail()
    tl.amax = amax.data_ptr()
    call('dsnt_conv_dgrad_strided', ptr(gyd), ptr(wdg), ptr(dx), None, None, C.byref(g), None, C.byref(tl))
    assert (dx.cpu() - want).abs().max().item() <= 2e-5 * scale
    assert float(amax.max()) == float(dx.abs().max())                    # the bound of what was written
    first = dx.clone()
    call('dsnt_conv_dgrad_strided', ptr(gyd), ptr(wdg), ptr(dx), None, None, C.byref(g), None, None)
    assert torch.equal(dx, first)                                        # deterministic
    # accumulate in place
    call('dsnt_conv_dgrad_strided', ptr(gyd), ptr(wdg), ptr(dx), ptr(dx), None, C.byref(g), None, None)
    assert (dx.cpu() - 2 * want).abs().max().item() <= 4e-5 * scale
    # the path it replaces: zeros stuffed between the pixels of dY, stride-1 kernel on that
    pad_d = dil * (k - 1) - pad
    if pad_d >= 0 and Cout % 4 == 0:
        Hs, Ws = H + 2 * pad - dil * (k - 1), W + 2 * pad - dil * (k - 1)
        stuffed = torch.empty(N, Hs, Ws, Cout, device=DEV)
        call('dsnt_zero_insert', ptr(gyd), ptr(stuffed), N, Ho, Wo, Cout, Hs, Ws, s)
        gd = ConvGeom(N, Hs, Ws, Cout, H, W, Cin, k, k, 1, pad_d, dil)
        old = torch.empty(N, H, W, Cin, device=DEV)
        call('dsnt_conv_fwd', ptr(stuffed), ptr(wdg), None, ptr(old), None, None, 0, None, None, None, C.byref(gd))
        assert (old - first).abs().max().item() <= 2e-5 * scale
    # BatchNorm-backward epilogue: dz = dx * [scale x + shift > 0], per-tile (sum dz, sum dz xhat)
    xin = synthetic.tensor(tag + 'bx', (N, H, W, Cin), seed=3)
    sc = synthetic.tensor(tag + 'bs', (Cin,), seed=3, kind='uniform').abs() + 0.5
    sh = synthetic.tensor(tag + 'bh', (Cin,), seed=3, scale=0.3)
    mu = synthetic.tensor(tag + 'bm', (Cin,), seed=3, scale=0.2)
    inv = synthetic.tensor(tag + 'bi', (Cin,), seed=3, kind='uniform').abs() + 0.5
    tiles = _lib.fn('dsnt_conv_dgrad_strided_tiles')(C.byref(g))
    assert tiles == s * s * ((N * (-(-H // s)) * (-(-W // s)) + 31) // 32)
    xin_d, sc_d, sh_d, mu_d, inv_d = (t.to(DEV) for t in (xin, sc, sh, mu, inv))
    for relu in (1, 0):
        part = torch.full((tiles, 2, Cin), 5.0, device=DEV)
        dz = torch.empty(N, H, W, Cin, device=DEV)
        bnb = BnBwdEpilogue(ptr(xin_d), ptr(sc_d), ptr(sh_d), ptr(mu_d), ptr(inv_d), relu)
        call('dsnt_conv_dgrad_strided', ptr(gyd), ptr(wdg), ptr(dz), None, ptr(part), C.byref(g), C.byref(bnb), None)
        mask = (torch.addcmul(sh, xin, sc) > 0) if relu else torch.ones_like(xin, dtype=torch.bool)
        assert torch.equal(dz.cpu(), torch.where(mask, first.cpu(), torch.zeros(())))
        dzd = dz.cpu().double().reshape(-1, Cin)
        xh = ((xin - mu) * inv).double().reshape(-1, Cin)
        got = part.cpu().double().sum(0)
        assert (got[0] - dzd.sum(0)).abs().max().item() <= 1e-5 * max(1.0, dzd.abs().sum(0).max().item())
        assert (got[1] - (dzd * xh).sum(0)).abs().max().item() <= 1e-5 * max(1.0, (dzd * xh).abs().sum(0).max().item())
    # refusals
    f = _lib.fn('dsnt_conv_dgrad_strided')
    g1 = ConvGeom(N, H, W, Cin, H + 2 * pad - dil * (k - 1), W + 2 * pad - dil * (k - 1), Cout, k, k, 1, pad, dil)
    assert _lib.fn('dsnt_conv_dgrad_strided_ok')(C.byref(g1)) == 0                      # stride 1: the forward kernels
    assert f(ptr(gyd), ptr(wdg), ptr(dx), None, None, C.byref(g1), None, None, None) != 0
    assert f(ptr(gyd), ptr(wdg), ptr(dx), None, ptr(part), C.byref(g), None, None, None) != 0   # statistics without bnb
    assert f(ptr(gyd), ptr(wdg), ptr(dx), ptr(dx), ptr(part), C.byref(g), C.byref(bnb), None, None) != 0
    torch.cuda.synchronize()


class _SmoothResNet:
    """Both implementations without ReLU (see tests/test_model_gpu.py::_NoRelu for why)."""

    def __enter__(self):
        from dsnt_oracle import resnet as ores
        os.environ['DSNT_DEBUG_NO_RELU'] = '1'
        self.ores, self.saved = ores, ores.F

        class Shim:
            relu = staticmethod(lambda t: t)
        ores.F = Shim
        return self

    def __exit__(self, *a):
        os.environ.pop('DSNT_DEBUG_NO_RELU', None)
        self.ores.F = self.saved


def test_resnet_gradient_with_respect_to_the_input_image():
    """d loss / d image through the ResNet FCN (stem 7x7 / 2 data gradient by zero-stuffing, 3x3 / 2 max-pool, strided
    stage transitions) on the smooth network against the oracle: relative L2 <= 1e-3."""
    from dsnt.model import build_mpii_pose_model
    from dsnt_oracle import model as omodel
    import torch.nn as nn
    kw = dict(base='resnet18', dilate=2, truncate=0, output_strat='dsnt', reg='js')
    with _SmoothResNet():
        m = build_mpii_pose_model(**kw)
        o = omodel.build_mpii_pose_model(**kw)
        o.fcn[2] = nn.Identity()
        synthetic.fill_state_dict(m, seed=4)
        synthetic.fill_state_dict(o, seed=4)
        m.cuda().train()
        o.train()
        x, target, mask = synthetic.batch(2, size=128, seed=6, mask_p=0.8)
        xg = x.to(DEV).requires_grad_()
        m.forward_loss(m(xg), target.to(DEV), mask.to(DEV)).backward()
        xo = x.clone().requires_grad_()
        o.forward_loss(o(xo), target, mask).backward()
    e = (xg.grad.cpu().double() - xo.grad.double()).norm().item() / xo.grad.double().norm().item()
    assert xg.grad.shape == x.shape and e <= 1e-3, e


CASES = [
    # base, dilate, truncate, size, batch, mfma
    ('resnet18', 0, 0, 128, 4, 'f32'),
    ('resnet18', 0, 0, 128, 4, 'bf16x6'),
    ('resnet34', 0, 0, 256, 2, 'bf16x6'),      # BASELINE config 1's model and crop size
    ('resnet34', 0, 0, 256, 8, 'default'),     # BASELINE config 1 AS IS: batch 8, the production kernel selection (tests/test_model.py:39-63)
    ('resnet18', 2, 0, 64, 2, 'f32'),          # dilation surgery: strides removed, 3x3 convs dilated 2 and 4
    ('resnet18', 1, 1, 64, 2, 'bf16x6'),       # truncated + dilated
    ('resnet50', 0, 0, 256, 2, 'f32'),         # Bottleneck blocks (128 BN samples per channel at layer4)
]


@pytest.mark.parametrize('base,dilate,truncate,size,batch,mfma', CASES)
@pytest.mark.parametrize('smooth', [True, False])
def test_resnet_pose_model_vs_oracle(base, dilate, truncate, size, batch, mfma, smooth, monkeypatch):
    from dsnt.model import build_mpii_pose_model
    from dsnt_oracle import model as omodel
    import torch.nn as nn
    if mfma == 'default':
        monkeypatch.delenv('DSNT_MFMA', raising=False)
        monkeypatch.delenv('DSNT_BF16X6_MIN_ROWS', raising=False)
    else:
        monkeypatch.setenv('DSNT_MFMA', mfma)
    if mfma == 'bf16x6':
        monkeypatch.setenv('DSNT_BF16X6_MIN_ROWS', '0')
    kw = dict(base=base, dilate=dilate, truncate=truncate, output_strat='dsnt', reg='js')
    with (_SmoothResNet() if smooth else contextlib.nullcontext()):
        m = build_mpii_pose_model(**kw)
        o = omodel.build_mpii_pose_model(**kw)
        if smooth:
            o.fcn[2] = nn.Identity()
        assert list(m.state_dict().keys()) == list(o.state_dict().keys())
        synthetic.fill_state_dict(m, seed=4)
        synthetic.fill_state_dict(o, seed=4)
        m.cuda().train()
        o.train()
        x, target, mask = synthetic.batch(batch, size=size, seed=6, mask_p=0.8)
        out = m(x.to(DEV))
        loss = m.forward_loss(out, target.to(DEV), mask.to(DEV))
        loss.backward()
        out_o = o(x)
        loss_o = o.forward_loss(out_o, target, mask)
        loss_o.backward()
    hm = size // 32 * 2 ** max(dilate, truncate)
    assert out.shape == (batch, 16, 2) and m.heatmaps.shape == (batch, 16, hm, hm)
    assert (out.detach().cpu() - out_o.detach()).abs().max().item() <= 1e-4          # the north-star bar
    assert (m.heatmaps.detach().cpu() - o.heatmaps.detach()).abs().max().item() <= 1e-4
    assert abs(loss.item() - loss_o.item()) <= 1e-4 * max(1.0, abs(loss_o.item()))
    coords = m.compute_coords(out)
    assert coords.device.type == 'cpu' and coords.dtype == torch.float32
    # every parameter gradient
    floor = 1e-3 * max(q.grad.double().norm().item() for q in o.parameters())
    tol = 2e-3 if smooth else 0.25
    for (n, p), (_, q) in zip(m.named_parameters(), o.named_parameters()):
        e = (p.grad.cpu().double() - q.grad.double()).norm().item() / max(q.grad.double().norm().item(), floor)
        assert e <= tol, (n, e)
    fm = torch.cat([p.grad.cpu().reshape(-1) for p in m.parameters()]).double()
    fo = torch.cat([p.grad.reshape(-1) for p in o.parameters()]).double()
    cos = (fm @ fo / (fm.norm() * fo.norm())).item()
    assert cos >= (1 - 1e-6 if smooth else 0.995), cos
    # running statistics were updated like the oracle's
    for (n, b), (_, c) in zip(m.named_buffers(), o.named_buffers()):
        if 'running' in n:
            assert (b.cpu() - c).abs().max().item() <= 1e-4 * max(1.0, c.abs().max().item()), n
    if smooth:
        return
    # eval mode on realistic running statistics (one more train forward with momentum 1: after a single
    # momentum-0.1 update they are still mostly the initial 0 / 1 and the 18..50-layer eval network saturates)
    for mod in list(m.modules()) + list(o.modules()):
        if isinstance(mod, nn.BatchNorm2d):
            mod.momentum = 1.0
    with torch.no_grad():
        m(x.to(DEV)); o(x)
    m.eval(); o.eval()
    with torch.no_grad():
        ev, ev_o = m(x.to(DEV)), o(x)
    assert (ev.cpu() - ev_o).abs().max().item() <= 1e-4
    from dsnt import optim
    m.train()
    opt = optim.RMSprop(m, lr=1e-4)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt.step()
    assert all(not torch.equal(before[n], p.detach()) for n, p in m.named_parameters())   # tests/test_model.py:39-63


@pytest.mark.parametrize('mfma', ['f32', 'bf16x6'])
def test_resnet_backward_through_an_eval_mode_forward(mfma, monkeypatch):
    """model.eval() + backward on a ResNet backbone (bn_add_act tails, strided data gradients): frozen BatchNorm statistics,
    every gradient against the oracle on the smooth network."""
    from dsnt.model import build_mpii_pose_model
    from dsnt_oracle import model as omodel
    import torch.nn as nn
    monkeypatch.setenv('DSNT_MFMA', mfma)
    if mfma == 'bf16x6':
        monkeypatch.setenv('DSNT_BF16X6_MIN_ROWS', '0')
    kw = dict(base='resnet18', output_strat='dsnt', reg='js')
    with _SmoothResNet():
        m = build_mpii_pose_model(**kw)
        o = omodel.build_mpii_pose_model(**kw)
        o.fcn[2] = nn.Identity()
        synthetic.fill_state_dict(m, seed=8)
        synthetic.fill_state_dict(o, seed=8)
        m.cuda().train()
        o.train()
        x, target, mask = synthetic.batch(4, size=128, seed=9, mask_p=0.8)
        for mod in list(m.modules()) + list(o.modules()):
            if isinstance(mod, nn.BatchNorm2d):
                mod.momentum = 1.0
        with torch.no_grad():
            m(x.to(DEV)); o(x)
        m.eval(); o.eval()
        stats = {n: b.detach().clone() for n, b in m.named_buffers() if 'running' in n}
        out = m(x.to(DEV))
        loss = m.forward_loss(out, target.to(DEV), mask.to(DEV))
        loss.backward()
        out_o = o(x)
        loss_o = o.forward_loss(out_o, target, mask)
        loss_o.backward()
    assert (out.detach().cpu() - out_o.detach()).abs().max().item() <= 1e-4
    assert abs(loss.item() - loss_o.item()) <= 1e-4 * max(1.0, abs(loss_o.item()))
    floor = 1e-3 * max(q.grad.double().norm().item() for q in o.parameters())
    for (n, p), (_, q) in zip(m.named_parameters(), o.named_parameters()):
        e = (p.grad.cpu().double() - q.grad.double()).norm().item() / max(q.grad.double().norm().item(), floor)
        assert e <= 2e-3, (n, e)
    for n, b in m.named_buffers():
        if 'running' in n:
            assert torch.equal(b, stats[n]), n


def test_resnet_surface():
    from dsnt.model import build_mpii_pose_model
    m = build_mpii_pose_model(base='resnet34')
    assert m.heatmap_size == 7 and m.image_specs.size == 224 and m.output_strat == 'dsnt'
    assert m.hm_conv.in_channels == 512 and m.hm_conv.bias is None
    assert build_mpii_pose_model(base='resnet18', dilate=2).heatmap_size == 28
    assert build_mpii_pose_model(base='resnet50', truncate=1).hm_conv.in_channels == 1024
    with pytest.raises(RuntimeError, match='HIP device only'):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(Exception, match='unsupported base model type'):
        build_mpii_pose_model(base='resnet99')


@pytest.mark.parametrize('base,size', [('resnet18', 128), ('hg1', 64)])
def test_fc_output_strategy_vs_oracle(base, size, monkeypatch):
    """`output_strat='fc'` (model.py:222-223, 293-303 / 196-198): heat-maps -> Linear(H*W, 2) on HIP kernels;
    coords, loss and the gradients of out_fc and of the backbone against the oracle (smooth network)."""
    from dsnt.model import build_mpii_pose_model
    from dsnt_oracle import model as omodel
    import torch.nn as nn
    monkeypatch.setenv('DSNT_MFMA', 'f32')
    kw = dict(base=base, output_strat='fc', reg='js', preact='sigmoid')
    if base.startswith('resnet'):
        kw['dilate'] = 1        # 8x8 heat-maps from a 128-px input (heatmap_size is 7 * 2^dilate = 14 at 224 px)
    smooth = _SmoothResNet() if base.startswith('resnet') else contextlib.nullcontext()
    with smooth:
        m = build_mpii_pose_model(**kw)
        o = omodel.build_mpii_pose_model(**kw)
        if base.startswith('resnet'):
            o.fcn[2] = nn.Identity()
            hw = (size // 16) ** 2          # one stride removed by the surgery
        else:
            hw = (size // 4) ** 2
        for mod in (m, o):                  # the constructor sizes out_fc for the canonical crop (model.py:101,217-223)
            mod.out_fc = nn.Linear(hw, 2)
        assert list(m.state_dict().keys()) == list(o.state_dict().keys())
        synthetic.fill_state_dict(m, seed=8)
        synthetic.fill_state_dict(o, seed=8)
        m.cuda().train()
        o.train()
        x, target, mask = synthetic.batch(4, size=size, seed=9, mask_p=0.8)
        out = m(x.to(DEV))
        loss = m.forward_loss(out, target.to(DEV), mask.to(DEV))
        loss.backward()
        out_o = o(x)
        loss_o = o.forward_loss(out_o, target, mask)
        loss_o.backward()
    a = out[-1] if isinstance(out, list) else out
    b = out_o[-1] if isinstance(out_o, list) else out_o
    assert a.shape == (4, 16, 2) and (a.detach().cpu() - b.detach()).abs().max().item() <= 1e-4
    assert abs(loss.item() - loss_o.item()) <= 1e-4 * max(1.0, abs(loss_o.item()))
    floor = 1e-3 * max(q.grad.double().norm().item() for q in o.parameters())
    tol = 2e-3 if base.startswith('resnet') else 0.25        # the hourglass side keeps its ReLUs here
    for (n, p), (_, q) in zip(m.named_parameters(), o.named_parameters()):
        assert p.grad is not None, n
        e = (p.grad.cpu().double() - q.grad.double()).norm().item() / max(q.grad.double().norm().item(), floor)
        assert e <= (2e-3 if n.startswith('out_fc') else tol), (n, e)
    opt = torch.optim.RMSprop(m.parameters(), lr=1e-4)      # out_fc lives outside the flat arena: stock optimiser
    w0 = m.out_fc.weight.detach().clone()
    opt.step()
    assert not torch.equal(w0, m.out_fc.weight.detach())
