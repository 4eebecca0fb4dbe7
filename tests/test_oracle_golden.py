"""Oracle vs the committed golden vectors (made from the reference by
tests/golden/make_golden.py).  Runs anywhere — this is the oracle's pin on the
GPU box, where /root/reference does not exist."""
import numpy as np
import pytest
import torch

from dsnt import synthetic
from dsnt_oracle import nn as onn, hourglass as ohg, model as omodel
from dsnt_oracle.evaluator import PCKhEvaluator
import golden_util as gu


@pytest.mark.parametrize('tag,dtype,tol', [('f32', torch.float32, 2e-6), ('f64', torch.float64, 1e-12)])
def test_head(tag, dtype, tol):
    g = gu.load('head_' + tag)
    logits = (synthetic.tensor('head.logits', (4, 16, 64, 64), seed=11) * 3).to(dtype).requires_grad_()
    target = synthetic.tensor('head.target', (4, 16, 2), seed=11, kind='uniform').to(dtype)
    mask = (synthetic.tensor('head.mask', (4, 16), seed=11, kind='uniform') > -0.6).to(dtype)
    hm = omodel.hm_preact(logits, 'softmax')
    coords = onn.dsnt(hm)
    assert np.abs(coords.detach().numpy() - g['coords']).max() <= tol
    gu.check_summary(g, 'heatmaps', hm, tol)
    eu = onn.euclidean_loss(coords, target, mask)
    assert abs(eu.item() - float(g['euclid'])) <= tol * 10
    for reg in ('js', 'kl', 'mse', 'var'):
        r = omodel.calculate_reg_loss(target, mask, reg, hm, 1.0)
        assert abs(r.item() - float(g['reg_' + reg])) <= tol * 10 * max(1, abs(r.item()))
        coeff = 100.0 if reg == 'var' else 1.0
        gl, = torch.autograd.grad(eu + coeff * r, logits, retain_graph=True)
        gu.check_summary(g, 'dlogits_' + reg, gl, tol)
    for preact in ('thresholded_softmax', 'abs', 'relu', 'sigmoid'):
        gu.check_summary(g, 'preact_' + preact, omodel.hm_preact(logits, preact), tol)


@pytest.mark.parametrize('kind', ['bottleneck', 'hourglass'])
def test_blocks(kind):
    g = gu.load(kind)
    m = ohg.Bottleneck(256, 128) if kind == 'bottleneck' else ohg.Hourglass(ohg.Bottleneck, 1, 128, 4)
    synthetic.fill_state_dict(m, seed=5)
    m.train()
    hw = 16 if kind == 'bottleneck' else 32
    x = synthetic.tensor(kind + '.x', (2, 256, hw, hw), seed=5).requires_grad_()
    y = m(x)
    y.backward(synthetic.tensor(kind + '.gy', (2, 256, hw, hw), seed=5))
    gu.check_summary(g, 'y', y, 1e-5)
    gu.check_summary(g, 'dx', x.grad, 1e-5)
    for n, p in m.named_parameters():
        want = float(g['gradnorm.' + n])
        assert abs(p.grad.double().norm().item() - want) <= 1e-4 * max(1.0, want), n


@pytest.mark.parametrize('base,size,reg,tag', [('hg1', 128, 'none', 'hg1_128'), ('hg2', 128, 'js', 'hg2_128'),
                                               ('hg8', 128, 'js', 'hg8_128')])
def test_end_to_end(base, size, reg, tag):
    g = gu.load(tag)
    m = omodel.build_mpii_pose_model(base=base, output_strat='dsnt', reg=reg)
    synthetic.fill_state_dict(m, seed=0)
    m.train()
    x, target, mask = synthetic.batch(2, size=size, seed=1, mask_p=0.9)
    outs = m(x)
    loss = m.forward_loss(outs, target, mask)
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * max(1, abs(loss.item()))
    for i, o in enumerate(outs):
        assert np.abs(o.detach().numpy() - g['coords%d' % i]).max() <= 1e-5
    for n, p in m.named_parameters():
        want = float(g['gradnorm.' + n])
        assert abs(p.grad.double().norm().item() - want) <= 1e-3 * max(1e-3, want), n
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 1.0
    with torch.no_grad():
        m(x)
    m.eval()
    with torch.no_grad():
        assert np.abs(m(x)[-1].numpy() - g['eval_coords']).max() <= 1e-4


def test_pckh():
    g = gu.load('pckh')
    _, target, mask = synthetic.batch(64, size=8, seed=3, mask_p=0.85)
    pred = target + synthetic.tensor('pckh.noise', (64, 16, 2), seed=3, scale=0.15)
    head, m, b = synthetic.pckh_inputs(64)
    ev = PCKhEvaluator(0.5)
    ev.add(torch.bmm(pred.double(), m) + b, torch.bmm(target.double(), m) + b, mask, head)
    for k, meter in ev.meters.items():
        assert abs(meter.value()[0] - float(g[k])) <= 1e-12, k
