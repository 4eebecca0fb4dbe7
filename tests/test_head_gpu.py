"""HIP DSNT head vs the oracle and the reference's known answers (through the C ABI).

Tolerances: the reference's own test tolerance is 1e-5 (tests/common.py:72, double); the HIP
path computes in fp32, so values are held to 2e-6 absolute (coords) / 1e-5 relative (losses,
gradients) against the fp32 oracle and to 1e-5 against the reference's known answers.
"""
import numpy as np
import pytest
import torch

from dsnt import synthetic
import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def _head_inputs(dev):
    logits = (synthetic.tensor('head.logits', (4, 16, 64, 64), seed=11) * 3)
    target = synthetic.tensor('head.target', (4, 16, 2), seed=11, kind='uniform')
    mask = (synthetic.tensor('head.mask', (4, 16), seed=11, kind='uniform') > -0.6).float()
    return logits, target, mask


def test_known_answers(dev):
    import dsnt.nn as dn
    # tests/test_nn.py:10-50 (dsnt fwd/bwd)
    h = torch.zeros(1, 1, 5, 5)
    h[0, 0, 1, 3] = h[0, 0, 2, 2] = h[0, 0, 2, 4] = h[0, 0, 3, 3] = 0.1
    h[0, 0, 2, 3] = 0.6
    hv = h.to(dev).requires_grad_()
    out = dn.dsnt(hv)
    assert (out.detach().cpu() - torch.tensor([[[0.4, 0.0]]])).abs().max() <= 1e-5
    torch.nn.functional.mse_loss(out, torch.tensor([[[0.5, 0.5]]], device=dev)).backward()
    want = torch.tensor([[[[0.48 - 0.04 * c - 0.20 * r for c in range(5)] for r in range(5)]]])
    assert (hv.grad.cpu() - want).abs().max() <= 1e-5
    # batch-less (tests/test_nn.py:52-66)
    assert dn.dsnt(h[0].to(dev)).shape == (1, 2)
    # tests/test_nn.py:86-130 (euclid)
    a = torch.tensor([[[3.0, 4], [3, 4]], [[3, 4], [3, 4]]], device=dev, requires_grad=True)
    loss = dn.euclidean_loss(a, torch.zeros(2, 2, 2, device=dev))
    loss.backward()
    assert abs(loss.item() - 5.0) <= 1e-5
    assert (a.grad.cpu() - torch.tensor([0.15, 0.20]).expand(2, 2, 2)).abs().max() <= 1e-5
    o = torch.tensor([[[0.0, 0], [1, 1], [0, 0]], [[1, 1], [0, 0], [0, 0]]], device=dev)
    m = torch.tensor([[1.0, 0, 1], [0, 1, 1]], device=dev)
    assert abs(dn.euclidean_loss(o, torch.zeros(2, 3, 2, device=dev), m).item()) <= 1e-5
    # tests/test_nn.py:134-149 (thresholded softmax)
    got = dn.thresholded_softmax(torch.tensor([[2.0, 1, 3], [4, 0, 0]], device=dev), 1.5).cpu()
    want = torch.tensor([[0.26894142, 0, 0.73105858], [1, 0, 0]])
    assert (got - want).abs().max() <= 1e-5
    got = dn.thresholded_softmax(torch.tensor([2.0, 1, 3], device=dev), 1.5).cpu()
    assert (got - want[0]).abs().max() <= 1e-5
    # tests/test_nn.py:158-167 (make_gauss)
    g = dn.make_gauss(torch.tensor([0.0, 0.0], device=dev), 5, 5, sigma=0.4).cpu()
    assert abs(g[2, 2].item() - 0.1621) <= 1e-4 and abs(g[0, 0].item() - 0.0030) <= 1e-4
    assert abs(g[1, 2].item() - 0.0983) <= 1e-4 and abs(g.sum().item() - 1) <= 1e-5
    # tests/test_nn.py:204-224 (KL with mask)
    t = torch.zeros(2, 4, 4)
    t[0, 2, 3] = t[0, 3, 2] = 0.1; t[0, 3, 3] = 0.8
    t[1, 0, 0] = 0.8; t[1, 0, 1] = t[1, 1, 0] = 0.1
    kl = dn.kl_reg_loss(t.to(dev), torch.tensor([[1.0, 1], [0, 0]], device=dev), 1,
                        torch.tensor([1.0, 0], device=dev))
    assert abs(kl.item() - 1.2228811717796824) <= 1e-5


@pytest.mark.parametrize('fn,shift_mean', [('kl_reg_loss', True), ('mse_reg_loss', True),
                                           ('js_reg_loss', True), ('variance_reg_loss', False)])
def test_reg_loss_minimum(dev, fn, shift_mean):  # tests/test_nn.py:170-239
    import dsnt.nn as dn
    mean, std = torch.tensor([0.0, 0.0], device=dev), 0.4

    def calc(m, s):
        return getattr(dn, fn)(dn.make_gauss(m, 5, 5, sigma=s), mean, std, mask=None).item()

    lo = calc(mean, std)
    assert abs(lo) <= 1e-3
    assert calc(mean, std + 0.2) > lo + 1e-3 and calc(mean, std - 0.2) > lo + 1e-3
    if shift_mean:
        assert calc(mean + 0.1, std) > lo + 1e-3 and calc(mean - 0.1, std) > lo + 1e-3


def test_thresholded_softmax_grad(dev):  # tests/test_nn.py:140-154 (gradcheck, here vs oracle fp64)
    import dsnt.nn as dn
    from dsnt_oracle import nn as onn
    x = synthetic.tensor('ts.x', (3, 20), seed=4)
    g = synthetic.tensor('ts.g', (3, 20), seed=4)
    xd = x.to(dev).requires_grad_()
    dn.thresholded_softmax(xd, 0).backward(g.to(dev))
    xo = x.double().requires_grad_()
    onn.thresholded_softmax(xo, 0).backward(g.double())
    assert (xd.grad.cpu().double() - xo.grad).abs().max() <= 1e-6


@pytest.mark.parametrize('h,w,sigma', [(64, 64, 1 / 32), (5, 5, 0.4), (12, 20, 0.15)])
def test_make_gauss_grad_vs_oracle(dev, h, w, sigma):
    """`make_gauss` is differentiable in `coords` (nn.py:168-205): closed-form HIP backward vs the oracle's fp64
    autograd, directly and through a KL built from the compatibility helpers (bar 2e-5 relative to the gradient scale)."""
    import dsnt.nn as dn
    from dsnt_oracle import nn as onn
    mu = synthetic.tensor('mg.mu', (3, 16, 2), seed=9, kind='uniform') * 0.8
    G = synthetic.tensor('mg.g', (3, 16, h, w), seed=9)
    md = mu.to(dev).requires_grad_()
    out = dn.make_gauss(md, w, h, sigma)
    out.backward(G.to(dev))
    mo = mu.double().requires_grad_()
    oo = onn.make_gauss(mo, w, h, sigma)
    oo.backward(G.double())
    assert (out.detach().cpu().double() - oo.detach()).abs().max() <= 2e-6
    scale = float(mo.grad.abs().max())
    assert (md.grad.cpu().double() - mo.grad).abs().max() <= 2e-5 * scale
    # through a divergence, as a caller optimising the target means would use it
    p = torch.softmax(synthetic.tensor('mg.p', (3, 16, h * w), seed=9), -1).view(3, 16, h, w)
    md2 = mu.to(dev).requires_grad_()
    dn._kl_2d(p.to(dev), dn.make_gauss(md2, w, h, sigma)).sum().backward()
    mo2 = mu.double().requires_grad_()
    onn._kl_2d(p.double(), onn.make_gauss(mo2, w, h, sigma)).sum().backward()
    # (KL's 1/q amplifies fp32 rounding of the far Gaussian tail: hold the direction and the scale)
    gd, go = md2.grad.cpu().double().flatten(), mo2.grad.flatten()
    assert torch.isfinite(gd).all()
    assert float(gd @ go / (gd.norm() * go.norm())) > 1 - 1e-6


@pytest.mark.parametrize('reg', ['js', 'kl', 'mse', 'var'])
@pytest.mark.parametrize('h,w,sigma', [(64, 64, 2 / 64), (5, 5, 0.4), (12, 20, 0.15)])
def test_reg_losses_are_differentiable_in_the_target_means(dev, reg, h, w, sigma):
    """The reference builds the target Gaussian with make_gauss(mu_t) inside autograd (nn.py:219-271): kl / js / mse are
    differentiable in mu_t (and in the heat-maps at the same time); `var` never reads mu_t.  HIP (fused: the
    divergence's derivative in the target pixel composed with make_gauss's backward, `dsnt_reg_bwd_mu`) vs the oracle's
    fp64 autograd, with a mask, with mu_t broadcast over the batch, bar 2e-5 of the gradient scale (KL: its 1/q
    amplifies the fp32 rounding of the Gaussian's far tail — direction and scale are held instead)."""
    import dsnt.nn as dn
    from dsnt_oracle import nn as onn
    fn = {'js': 'js_reg_loss', 'kl': 'kl_reg_loss', 'mse': 'mse_reg_loss', 'var': 'variance_reg_loss'}[reg]
    p = torch.softmax(synthetic.tensor('rm.p', (3, 16, h * w), seed=11) * 2.0, -1).view(3, 16, h, w)
    mu = synthetic.tensor('rm.mu', (3, 16, 2), seed=11, kind='uniform') * 0.8
    mask = (synthetic.tensor('rm.m', (3, 16), seed=11, kind='uniform') > 0.3).float()
    for mu0, use_mask in ((mu, True), (mu[:1], False)):            # second: one set of means for the whole batch
        pd, md = p.to(dev).requires_grad_(), mu0.to(dev).requires_grad_()
        loss = getattr(dn, fn)(pd, md, sigma, mask.to(dev) if use_mask else None)
        po, mo = p.double().requires_grad_(), mu0.double().requires_grad_()
        loss_o = getattr(onn, fn)(po, mo, sigma, mask.double() if use_mask else None)
        assert abs(loss.item() - loss_o.item()) <= 2e-5 * max(1.0, abs(loss_o.item()))
        if reg == 'var':
            loss.backward()
            loss_o.backward()
            assert md.grad is not None and float(md.grad.abs().max()) == 0.0 and tuple(md.grad.shape) == tuple(mu0.shape)
            assert mo.grad is None or float(mo.grad.abs().max()) == 0.0
            continue
        loss.backward()
        loss_o.backward()
        assert tuple(md.grad.shape) == tuple(mu0.shape)
        gd, go = md.grad.cpu().double().flatten(), mo.grad.flatten()
        assert torch.isfinite(gd).all()
        if reg == 'kl':
            assert float(gd @ go / (gd.norm() * go.norm())) > 1 - 1e-6 and abs(float(gd.norm() / go.norm()) - 1) < 1e-3
        else:
            assert (gd - go).abs().max() <= 2e-5 * float(go.abs().max()), (reg, (gd - go).abs().max(), go.abs().max())
        ph, pho = pd.grad.cpu().double(), po.grad
        assert (ph - pho).abs().max() <= 2e-5 * float(pho.abs().max())


def test_ops_vs_oracle_and_golden(dev):
    import dsnt.nn as dn
    from dsnt_oracle import nn as onn, model as omodel
    g32 = gu.load('head_f32')
    logits, target, mask = _head_inputs(dev)
    ld = logits.to(dev).requires_grad_()
    td, md = target.to(dev), mask.to(dev)
    lo = logits.clone().requires_grad_()
    hm_o = omodel.hm_preact(lo, 'softmax')
    co_o = onn.dsnt(hm_o)
    hm = dn.hm_preact(ld, 'softmax')
    co = dn.dsnt(hm)
    assert (co.detach().cpu() - co_o.detach()).abs().max() <= 2e-6
    assert np.abs(co.detach().cpu().numpy() - g32['coords']).max() <= 2e-6   # golden (reference)
    gu.check_summary(g32, 'heatmaps', hm, 2e-6)
    eu = dn.euclidean_loss(co, td, md)
    eu_o = onn.euclidean_loss(co_o, target, mask)
    assert abs(eu.item() - eu_o.item()) <= 1e-5 and abs(eu.item() - float(g32['euclid'])) <= 1e-5
    for reg in ('js', 'kl', 'mse', 'var'):
        fn = {'js': 'js_reg_loss', 'kl': 'kl_reg_loss', 'mse': 'mse_reg_loss', 'var': 'variance_reg_loss'}[reg]
        r = getattr(dn, fn)(hm, td, 2.0 / 64, md)
        r_o = getattr(onn, fn)(hm_o, target, 2.0 / 64, mask)
        assert abs(r.item() - r_o.item()) <= 1e-5 * max(1, abs(r_o.item())), reg
        assert abs(r.item() - float(g32['reg_' + reg])) <= 1e-5 * max(1, abs(r_o.item())), reg
        coeff = 100.0 if reg == 'var' else 1.0
        gl, = torch.autograd.grad(eu + coeff * r, ld, retain_graph=True)
        gl_o, = torch.autograd.grad(eu_o + coeff * r_o, lo, retain_graph=True)
        scale = gl_o.abs().max().item()
        assert (gl.cpu() - gl_o).abs().max() <= 2e-5 * scale, reg
        gu.check_summary(g32, 'dlogits_' + reg, gl, 2e-5)
        ghm, = torch.autograd.grad(r, hm, retain_graph=True)
        gu.check_summary(g32, 'dhm_' + reg, ghm, 2e-5)
    assert abs(dn.euclidean_loss(co, td, None).item() - float(g32['euclid_nomask'])) <= 1e-5
    assert abs(dn.js_reg_loss(hm, td, 2.0 / 64, None).item() - float(g32['js_nomask'])) <= 1e-5
    for preact in ('thresholded_softmax', 'abs', 'relu', 'sigmoid'):
        y = dn.hm_preact(ld, preact)
        y_o = omodel.hm_preact(lo, preact)
        gu.check_summary(g32, 'preact_' + preact, y, 2e-6)
        gy = synthetic.tensor('pg', tuple(y.shape), seed=9)
        a, = torch.autograd.grad(y, ld, gy.to(dev))
        b, = torch.autograd.grad(y_o, lo, gy)
        assert (a.cpu() - b).abs().max() <= 2e-5 * max(1e-6, b.abs().max().item()), preact
    with pytest.raises(Exception, match='unrecognised heatmap preactivation'):
        dn.hm_preact(ld, 'tanh')


@pytest.mark.parametrize('reg', ['none', 'js', 'kl', 'mse', 'var'])
@pytest.mark.parametrize('use_mask', [True, False])
def test_fused_head_vs_oracle(dev, reg, use_mask):
    import dsnt.nn as dn
    from dsnt_oracle import nn as onn, model as omodel
    logits, target, mask = _head_inputs(dev)
    if not use_mask:
        mask = None
    coeff = 100.0 if reg == 'var' else 1.0
    ld = logits.to(dev).requires_grad_()
    hm, co = dn.head_forward(ld)
    loss = dn.head_loss(ld, hm.detach(), co.detach(), target.to(dev),
                        None if mask is None else mask.to(dev), reg, 2.0 / 64, coeff)
    loss.backward()
    lo = logits.double().requires_grad_()
    hm_o = omodel.hm_preact(lo, 'softmax')
    co_o = onn.dsnt(hm_o)
    t64 = target.double()
    m64 = None if mask is None else mask.double()
    loss_o = onn.euclidean_loss(co_o, t64, m64) + coeff * omodel.calculate_reg_loss(t64, m64, reg, hm_o, 1.0)
    loss_o.backward()
    assert (co.detach().cpu().double() - co_o.detach()).abs().max() <= 2e-6
    assert abs(loss.item() - loss_o.item()) <= 1e-5 * max(1, abs(loss_o.item()))
    scale = lo.grad.abs().max().item()
    assert (ld.grad.cpu().double() - lo.grad).abs().max() <= 2e-5 * scale


def test_fused_head_upstream_gradient_and_second_backward(dev):
    """The train-step kernel leaves d loss / d logits for an upstream gradient of 1; any other upstream value rescales
    it on the device, and a second backward through the same graph (retain_graph) falls back to dsnt_head_bwd."""
    import dsnt.nn as dn
    logits, target, mask = _head_inputs(dev)
    ld = logits.to(dev).requires_grad_()
    hm, co = dn.head_forward(ld)
    loss = dn.head_loss(ld, hm.detach(), co.detach(), target.to(dev), mask.to(dev), 'js', 2.0 / 64, 1.0)
    g1, = torch.autograd.grad(loss, ld, retain_graph=True)              # fused gradient, upstream 1
    g2, = torch.autograd.grad(loss * 2.5, ld, retain_graph=True)        # second pass: dsnt_head_bwd, upstream 2.5
    assert (g2 - 2.5 * g1).abs().max().item() <= 2e-5 * 2.5 * g1.abs().max().item()
    hm2, co2 = dn.head_forward(ld)
    loss2 = dn.head_loss(ld, hm2.detach(), co2.detach(), target.to(dev), mask.to(dev), 'js', 2.0 / 64, 1.0)
    g3, = torch.autograd.grad(loss2 * -0.75, ld)                        # fused gradient rescaled in place
    assert torch.equal(g3, g1 * -0.75)
    with torch.no_grad():                                               # no gradient wanted: value-only kernel, same loss
        l3 = dn.head_loss(ld, hm.detach(), co.detach(), target.to(dev), mask.to(dev), 'js', 2.0 / 64, 1.0)
    assert abs(l3.item() - loss.item()) <= 1e-5 * abs(loss.item())


@pytest.mark.parametrize('h,w', [(5, 5), (7, 7), (14, 14), (28, 28), (8, 8), (64, 48), (96, 96)])
def test_odd_shapes(dev, h, w):
    """Ragged / unaligned rows (ResNet heat-maps 7..28, H*W not a multiple of 4, > 4096)."""
    import dsnt.nn as dn
    from dsnt_oracle import nn as onn, model as omodel
    x = synthetic.tensor('odd', (3, 16, h, w), seed=h * 100 + w) * 2
    t = synthetic.tensor('oddt', (3, 16, 2), seed=7, kind='uniform')
    xd = x.to(dev).requires_grad_()
    hm, co = dn.head_forward(xd)
    loss = dn.head_loss(xd, hm.detach(), co.detach(), t.to(dev), None, 'js', 2.0 / w, 1.0)
    loss.backward()
    xo = x.double().requires_grad_()
    hm_o = omodel.hm_preact(xo, 'softmax')
    co_o = onn.dsnt(hm_o)
    loss_o = onn.euclidean_loss(co_o, t.double()) + onn.js_reg_loss(hm_o, t.double(), 2.0 / w)
    loss_o.backward()
    assert (co.detach().cpu().double() - co_o.detach()).abs().max() <= 3e-6
    assert abs(loss.item() - loss_o.item()) <= 2e-5 * max(1, abs(loss_o.item()))
    assert (xd.grad.cpu().double() - xo.grad).abs().max() <= 3e-5 * xo.grad.abs().max().item()


def test_head_large_property(dev):
    """Full-size batch (B=256): size-independent properties — heat-maps sum to 1, coords inside
    (-1, 1), d loss/d logits rows sum to 0 (softmax backward), finite everywhere."""
    import dsnt.nn as dn
    B = 256
    g = torch.Generator(device='cpu').manual_seed(5)
    x = (torch.randn(B, 16, 64, 64, generator=g) * 4).to(dev).requires_grad_()
    t = (torch.rand(B, 16, 2, generator=g) * 2 - 1).to(dev)
    hm, co = dn.head_forward(x)
    assert (hm.sum((-1, -2)) - 1).abs().max().item() <= 1e-5
    assert co.abs().max().item() < 1
    dn.head_loss(x, hm.detach(), co.detach(), t, None, 'js', 2.0 / 64, 1.0).backward()
    assert torch.isfinite(x.grad).all()
    assert x.grad.sum((-1, -2)).abs().max().item() <= 1e-6


def test_errors_are_loud(dev):
    import dsnt.nn as dn
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        dn.dsnt(torch.zeros(1, 1, 4, 4))
    with pytest.raises(RuntimeError, match='float32'):
        dn.dsnt(torch.zeros(1, 1, 4, 4, dtype=torch.float64, device=dev))
