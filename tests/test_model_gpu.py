"""HIP backbone + head vs the oracle and the golden vectors (made from the reference).

Bars (BASELINE.json north_star): joint coordinates within 1e-4 of the CPU path on identical
inputs; losses to 1e-4 relative; every parameter gradient to a few 1e-3 of its own max-norm
scale (fp32 accumulation-order differences through ~100 layers with batch-norm).
"""
import numpy as np
import pytest
import torch

from dsnt import synthetic
import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(params=['f32', 'bf16x6', 'f16x3'], autouse=True)
def mfma_path(request, monkeypatch):
    """Every model-level parity test runs twice: convolutions on the fp32 MFMA, and on the
    split-bf16 (bf16x6) matrix-core path forced on for ALL eligible convolutions (in production
    it is used from 16384 output rows up, which the small oracle-sized batches never reach)."""
    if request.param == 'f32':
        monkeypatch.setenv('DSNT_MFMA', 'f32')
    else:
        monkeypatch.setenv('DSNT_MFMA', 'bf16x6')
        monkeypatch.setenv('DSNT_BF16X6_MIN_ROWS', '0')
        # 'f16x3': two fp16 planes / three MFMAs wherever an operand bound exists, bf16x6 elsewhere
        monkeypatch.setenv('DSNT_SPLIT', 'f16x3' if request.param == 'f16x3' else 'bf16x6')
    return request.param


def _count_launches(module, name):
    progs = module._runner().programs.values()
    return sum(1 for p in progs for (_, _, n, _) in p.tape.fwd + p.tape.bwd if n == name)


def _rel_l2(got, want, floor=0.0):
    return (got.double() - want.double()).norm().item() / max(want.double().norm().item(), floor, 1e-30)


def _grads_close(m, o, tol, what):
    """Every parameter gradient: relative L2 error <= tol (floor = 1e-3 of the largest gradient
    norm: conv biases that feed a batch-norm have an exactly-zero true gradient, only noise)."""
    floor = 1e-3 * max(q.grad.double().norm().item() for q in o.parameters() if q.grad is not None)
    worst = (0.0, None)
    for (n, p), (_, q) in zip(m.named_parameters(), o.named_parameters()):
        assert p.grad is not None and p.grad.shape == q.grad.shape, n
        e = _rel_l2(p.grad.cpu(), q.grad, floor)
        worst = max(worst, (e, n))
        assert e <= tol, (what, n, e)
    return worst


class _NoRelu:
    """Context: run BOTH implementations without ReLU.  Two fp32 implementations of the same
    network can place a pre-activation that is within ~1e-7 of zero on opposite sides of the ReLU
    kink; that flips one mask element and moves gradients by O(1e-2) at the tiny batch sizes the
    CPU oracle can afford, although every kernel is exact.  The smooth network has no such kinks,
    so there the gradients must agree tightly — this is the proof that the backward kernels,
    accumulation and buffer-donation logic are right.  The ReLU network is then checked with a
    flip-tolerant bound."""

    def __enter__(self):
        import os
        import torch.nn.functional as F
        from dsnt_oracle import hourglass as ohg
        os.environ['DSNT_DEBUG_NO_RELU'] = '1'
        self.ohg, self.saved = ohg, ohg.F

        class Shim:
            relu = staticmethod(lambda t: t)
            max_pool2d = staticmethod(F.max_pool2d)
            interpolate = staticmethod(F.interpolate)
        ohg.F = Shim
        return self

    def __exit__(self, *a):
        import os
        os.environ.pop('DSNT_DEBUG_NO_RELU', None)
        self.ohg.F = self.saved

    @staticmethod
    def strip(oracle_model):
        import torch.nn as nn
        for seq in oracle_model.hg.fc:
            seq[2] = nn.Identity()


@pytest.mark.parametrize('kind', ['bottleneck', 'hourglass'])
@pytest.mark.parametrize('smooth', [True, False])
def test_blocks_vs_golden_and_oracle(kind, smooth):
    from dsnt import hourglass as dhg
    from dsnt_oracle import hourglass as ohg
    import contextlib
    g = gu.load(kind)
    hw = 16 if kind == 'bottleneck' else 32
    mk = (lambda mod: mod.Bottleneck(256, 128)) if kind == 'bottleneck' else \
        (lambda mod: mod.Hourglass(mod.Bottleneck, 1, 128, 4))
    with (_NoRelu() if smooth else contextlib.nullcontext()):
        m, o = mk(dhg), mk(ohg)
        synthetic.fill_state_dict(m, seed=5)
        synthetic.fill_state_dict(o, seed=5)
        assert list(m.state_dict().keys()) == list(o.state_dict().keys())
        m.cuda().train()
        o.train()
        x = synthetic.tensor(kind + '.x', (2, 256, hw, hw), seed=5)
        gy = synthetic.tensor(kind + '.gy', (2, 256, hw, hw), seed=5)
        xd = x.to(DEV).requires_grad_()
        y = m(xd)
        y.backward(gy.to(DEV))
        xo = x.clone().requires_grad_()
        yo = o(xo)
        yo.backward(gy)
    assert _rel_l2(y.detach().cpu(), yo.detach()) <= 1e-5
    tol = 2e-4 if smooth else 3e-2
    assert _rel_l2(xd.grad.cpu(), xo.grad) <= tol
    _grads_close(m, o, tol, kind)
    if not smooth:      # golden vectors come from the (ReLU) reference
        gu.check_summary(g, 'y', y, 2e-5)
        for n, p in m.named_parameters():
            want = float(g['gradnorm.' + n])
            assert abs(p.grad.double().norm().item() - want) <= 3e-2 * max(1.0, want), n
    for (n, b), (_, c) in zip(m.named_buffers(), o.named_buffers()):
        if 'running' in n:
            s_ = max(1.0, c.abs().max().item())
            assert (b.cpu() - c).abs().max().item() <= 1e-5 * s_, n
            if not smooth:
                assert np.abs(b.cpu().double().numpy() - g['buf.' + n]).max() <= 1e-5 * s_, n


@pytest.mark.parametrize('base,size,reg,tag', [('hg1', 128, 'none', 'hg1_128'),
                                               ('hg2', 128, 'js', 'hg2_128'),
                                               ('hg2', 256, 'js', 'hg2_256'),
                                               ('hg8', 128, 'js', 'hg8_128')])
def test_end_to_end_vs_golden(base, size, reg, tag, mfma_path):
    """Golden vectors made from the reference: coords of EVERY stack (bar 1e-4), loss, heat-maps, running
    statistics, eval-mode coords; gradient norms flip-tolerantly (see _NoRelu).  hg8 = experiments/hourglass.json
    (eight stacks, seven inter-stack remaps: hourglass.py:166-175)."""
    from dsnt.model import build_mpii_pose_model
    g = gu.load(tag)
    m = build_mpii_pose_model(base=base, output_strat='dsnt', reg=reg)
    synthetic.fill_state_dict(m, seed=0)
    m.cuda().train()
    x, target, mask = synthetic.batch(2, size=size, seed=1, mask_p=0.9)
    x, target, mask = x.to(DEV), target.to(DEV), mask.to(DEV)
    outs = m(x)
    loss = m.forward_loss(outs, target, mask)
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 1e-4 * max(1.0, abs(float(g['loss'])))
    for i, o in enumerate(outs):
        err = np.abs(o.detach().cpu().numpy() - g['coords%d' % i]).max()
        assert err <= 1e-4, ('coords', i, err)          # the north-star bar
        gu.check_summary(g, 'heatmaps%d' % i, m.heatmaps_array[i], 1e-4)
    coords = m.compute_coords(outs)
    assert coords.device.type == 'cpu' and coords.dtype == torch.float32 and coords.shape == (2, 16, 2)
    assert m.heatmaps is m.heatmaps_array[0] and m.heatmaps.shape == (2, 16, size // 4, size // 4)
    norms = {k[9:]: float(g[k]) for k in g.files if k.startswith('gradnorm.')}
    fl = 1e-3 * max(norms.values())
    bad = [(n, p.grad.double().norm().item(), norms[n]) for n, p in m.named_parameters()
           if abs(p.grad.double().norm().item() - norms[n]) > 0.15 * max(fl, norms[n])]
    assert not bad, bad[:5]
    tot_got = np.sqrt(sum(p.grad.double().norm().item() ** 2 for p in m.parameters()))
    tot_want = np.sqrt(sum(v * v for v in norms.values()))
    assert abs(tot_got - tot_want) <= 2e-2 * tot_want, (tot_got, tot_want)
    has64 = 'eval_coords_f64' in g.files     # hg8: the reference's fp64 run is stored beside its fp32 run (make_golden.py)
    for n, b in m.named_buffers():
        if 'running' in n:
            want = float(g['bufsum.' + n])
            if has64:       # no further from the fp64 truth than twice the fp32 reference is (floor: the usual bar)
                w64 = float(g['bufsum_f64.' + n])
                # (floor 2e-4: the innermost levels average 8 samples; measured 1.1e-4 on the fp16x3 path, whose
                # different roundings flip other ReLUs there, <= 1e-4 on the other two)
                assert abs(b.double().sum().item() - w64) <= max(2e-4 * max(1.0, abs(w64)), 2 * abs(want - w64)), n
            else:
                assert abs(b.double().sum().item() - want) <= 1e-4 * max(1.0, abs(want)), n
    for mod in m.modules():                               # see tests/golden/make_golden.py
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 1.0
    with torch.no_grad():
        m(x)
    m.eval()
    with torch.no_grad():
        ev = m(x)[-1].cpu().numpy()
    if has64:
        ref_err = np.abs(g['eval_coords'] - g['eval_coords_f64']).max()
        ev_err = np.abs(ev - g['eval_coords_f64']).max()
        assert ev_err <= max(1e-4, 2 * ref_err), (ev_err, ref_err)
        for i, o in enumerate(outs):             # train-mode coordinates against the fp64 truth as well
            assert np.abs(o.detach().cpu().numpy() - g['coords%d_f64' % i]).max() <= 1e-4, i
    else:
        ev_err = np.abs(ev - g['eval_coords']).max()
        assert ev_err <= 1e-4, ev_err
    n6 = _count_launches(m.hg, 'dsnt_conv_fwd_bf16x6') + _count_launches(m.hg, 'dsnt_conv_fwd_bf16x6_ex')
    n16 = _count_launches(m.hg, 'dsnt_conv_fwd_f16x3_ex')
    if mfma_path == 'bf16x6':
        assert n6 > 100 and n16 == 0
    elif mfma_path == 'f16x3':
        assert n16 > 30 and n6 > 0, (n16, n6)          # train-mode BN+ReLU operands on fp16x3; eval mode and raw operands on bf16x6
    else:
        assert n6 == 0 and n16 == 0


@pytest.mark.parametrize('smooth', [True, False])
def test_hg2_every_gradient_vs_oracle(smooth):
    """All 396 parameter gradients of hg2 + DSNT + JS against the CPU oracle (batch 4, 128 px)."""
    import contextlib
    from dsnt.model import build_mpii_pose_model
    from dsnt_oracle import model as omodel
    with (_NoRelu() if smooth else contextlib.nullcontext()):
        m = build_mpii_pose_model(base='hg2', output_strat='dsnt', reg='js')
        o = omodel.build_mpii_pose_model(base='hg2', output_strat='dsnt', reg='js')
        if smooth:
            _NoRelu.strip(o)
        synthetic.fill_state_dict(m, seed=3)
        synthetic.fill_state_dict(o, seed=3)
        m.cuda().train()
        o.train()
        x, target, mask = synthetic.batch(4, size=128, seed=2, mask_p=0.8)
        outs = m(x.to(DEV))
        loss = m.forward_loss(outs, target.to(DEV), mask.to(DEV))
        loss.backward()
        outs_o = o(x)
        loss_o = o.forward_loss(outs_o, target, mask)
        loss_o.backward()
    for a, b in zip(outs, outs_o):
        assert (a.detach().cpu() - b.detach()).abs().max().item() <= 1e-4
    assert abs(loss.item() - loss_o.item()) <= 1e-4 * max(1.0, abs(loss_o.item()))
    pm, po = dict(m.named_parameters()), dict(o.named_parameters())
    assert list(pm) == list(po) and len(pm) == 396
    worst = _grads_close(m, o, 1e-3 if smooth else 0.2, 'hg2')
    flat_m = torch.cat([p.grad.cpu().reshape(-1) for p in pm.values()]).double()
    flat_o = torch.cat([p.grad.reshape(-1) for p in po.values()]).double()
    cos = (flat_m @ flat_o / (flat_m.norm() * flat_o.norm())).item()
    assert cos >= (1 - 1e-7 if smooth else 0.999), (cos, worst)
    if smooth:
        return
    # mask_var=None path (tests/test_model.py:56) and torch's gradient accumulation semantics
    m.zero_grad()
    l2 = m.forward_loss(m(x.to(DEV)), target.to(DEV), None)
    l2.backward()
    g1 = {n: p.grad.clone() for n, p in pm.items()}
    l3 = m.forward_loss(m(x.to(DEV)), target.to(DEV), None)
    l3.backward()                                        # no zero_grad: torch accumulates
    for n, p in pm.items():
        assert _rel_l2(p.grad, 2 * g1[n], 1e-3 * g1[n].norm().item() + 1e-12) <= 1e-5, n


def test_gradients_with_the_slab_reduction_inside_every_launch(monkeypatch):
    """DSNT_DEFER_REDUCE=0 (each weight gradient reduces its own slabs instead of one launch per parameter bucket) gives
    the same gradients as the default schedule — including the stem's 7x7 filter, whose gradient is produced in the
    space-to-depth layout and gathered back by a launch that has to follow the reduction on either path."""
    from dsnt.model import build_mpii_pose_model

    def grads(env):
        if env is not None:
            monkeypatch.setenv('DSNT_DEFER_REDUCE', env)
        else:
            monkeypatch.delenv('DSNT_DEFER_REDUCE', raising=False)
        m = build_mpii_pose_model(base='hg1', output_strat='dsnt', reg='js')
        synthetic.fill_state_dict(m, seed=4)
        m.cuda().train()
        x, target, mask = synthetic.batch(2, size=128, seed=6, mask_p=0.9)
        loss = m.forward_loss(m(x.to(DEV)), target.to(DEV), mask.to(DEV))
        loss.backward()
        return {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    a, b = grads(None), grads('0')
    assert float(b['hg.conv1.weight'].abs().max()) > 0
    for n in a:
        scale = max(float(a[n].abs().max()), 1e-12)
        assert float((a[n] - b[n]).abs().max()) <= 2e-5 * scale + 1e-9, n


@pytest.mark.parametrize('base', ['hg1'])
def test_gradient_with_respect_to_the_input_image(base):
    """d loss / d image (the reference's autograd provides it for any input that requires a gradient; train.py never asks):
    a program traced for such an input keeps the stem as the plain 7x7 / stride 2 convolution and takes its data gradient
    through the zero-stuffed stride-1 kernels.  Smooth network (no ReLU) against the oracle: relative L2 <= 1e-3; and the
    parameter gradients of that program equal the ordinary program's to the same bar (another stem kernel, same numbers)."""
    from dsnt.model import build_mpii_pose_model
    from dsnt_oracle import model as omodel
    kw = dict(output_strat='dsnt', reg='js')
    with _NoRelu():
        m = build_mpii_pose_model(base=base, **kw)
        o = omodel.build_mpii_pose_model(base=base, **kw)
        _NoRelu.strip(o)
        synthetic.fill_state_dict(m, seed=5)
        synthetic.fill_state_dict(o, seed=5)
        m.cuda().train()
        o.train()
        x, target, mask = synthetic.batch(2, size=128, seed=4, mask_p=0.8)
        xg = x.to(DEV).requires_grad_()
        loss = m.forward_loss(m(xg), target.to(DEV), mask.to(DEV))
        loss.backward()
        g_in = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
        for p in m.parameters():
            p.grad = None
        loss2 = m.forward_loss(m(x.to(DEV)), target.to(DEV), mask.to(DEV))      # the ordinary program of the same shape
        loss2.backward()
        xo = x.clone().requires_grad_()
        lo = o.forward_loss(o(xo), target, mask)
        lo.backward()
    assert xg.grad is not None and xg.grad.shape == x.shape
    assert abs(loss.item() - lo.item()) <= 1e-4 * max(1.0, abs(lo.item()))
    assert _rel_l2(xg.grad.cpu(), xo.grad) <= 1e-3, _rel_l2(xg.grad.cpu(), xo.grad)
    floor = 1e-3 * max(v.double().norm().item() for v in g_in.values())
    for n, p in m.named_parameters():
        assert _rel_l2(g_in[n], p.grad, floor) <= 1e-3, n


def test_hg8_every_gradient_vs_oracle_on_the_smooth_network():
    """hg8 + DSNT + JS (BASELINE config 5's model; batch 2, 128 px) against the CPU oracle: coordinates of all eight
    stacks within 1e-4, the loss, and — with the ReLUs taken out on both sides — every one of its 1464 parameter gradients
    against the oracle IN FP64, held to an envelope the oracle draws itself (the fp32 oracle's own distance from the fp64
    oracle, parameter by parameter) instead of one constant:
      * the bulk: the MEDIAN of (this path's distance / the fp32 oracle's distance) <= 2.5 (measured 1.2 - 1.3 on all three
        matrix-core paths) — a 2x loss of accuracy anywhere in the common path fails here, which a constant bar set by the
        worst parameter (5e-3 in round 3) would hide;
      * >= 80 % of the parameters within max(2e-4, 16 x the fp32 oracle's distance);
      * every parameter <= 5e-3 and the flat gradient's cosine >= 1 - 1e-7.
    Why not the envelope for ALL parameters: the network is not smooth even without its ReLUs.  Its max-pools route the
    gradient by arg-max, and the fp64 forward of this input has 2x2 windows whose two largest values differ by 2.5e-7 ..
    4.5e-7 of the tensor's scale in stack 2 (pools 9 and 11: tools/diag_hg8_envelope.py prints them) — any fp32 evaluation
    whose rounding differs from torch-CPU's may take the other pixel.  The parameters behind such a window then sit at
    1e-3 .. 2.7e-3 on the fp32-MFMA and fp16x3 paths ALIKE (same quantiles to three digits, whatever the switches:
    finalisation, lanes, reduction order) and at <= 4.2e-4 on the bf16x6 path, whose products carry 6e-9 instead of 1e-7
    and stay on the oracle's side of every tie.  (Round 3 read this as a "chaotic amplifier"; it is an arg-max flip.)"""
    from dsnt.model import build_mpii_pose_model
    from dsnt_oracle import model as omodel
    K, FLOOR = 16.0, 2e-4
    with _NoRelu():
        m = build_mpii_pose_model(base='hg8', output_strat='dsnt', reg='js')
        o = omodel.build_mpii_pose_model(base='hg8', output_strat='dsnt', reg='js')
        o64 = omodel.build_mpii_pose_model(base='hg8', output_strat='dsnt', reg='js')
        _NoRelu.strip(o)
        _NoRelu.strip(o64)
        synthetic.fill_state_dict(m, seed=3)
        synthetic.fill_state_dict(o, seed=3)
        synthetic.fill_state_dict(o64, seed=3)
        o64.double()
        m.cuda().train()
        o.train()
        o64.train()
        x, target, mask = synthetic.batch(2, size=128, seed=2, mask_p=0.8)
        outs = m(x.to(DEV))
        loss = m.forward_loss(outs, target.to(DEV), mask.to(DEV))
        loss.backward()
        outs_o = o(x)
        loss_o = o.forward_loss(outs_o, target, mask)
        loss_o.backward()
        o64.forward_loss(o64(x.double()), target.double(), mask.double()).backward()
    assert len(outs) == 8 and len(outs_o) == 8
    for a, b in zip(outs, outs_o):
        assert (a.detach().cpu() - b.detach()).abs().max().item() <= 1e-4
    assert abs(loss.item() - loss_o.item()) <= 1e-4 * max(1.0, abs(loss_o.item()))
    pm, po, p64 = dict(m.named_parameters()), dict(o.named_parameters()), dict(o64.named_parameters())
    assert list(pm) == list(po) == list(p64)
    floor = 1e-3 * max(q.grad.norm().item() for q in p64.values())
    ratios, inside, worst_abs = [], 0, 0.0
    for n in pm:
        e_hip = _rel_l2(pm[n].grad.cpu(), p64[n].grad, floor)
        e_o32 = _rel_l2(po[n].grad, p64[n].grad, floor)
        ratios.append(e_hip / max(e_o32, 1e-9))
        inside += e_hip <= max(FLOOR, K * e_o32)
        worst_abs = max(worst_abs, e_hip)
    ratios.sort()
    worst_ratio = ratios[-1]
    assert ratios[len(ratios) // 2] <= 2.5, ratios[len(ratios) // 2]
    assert inside >= 0.8 * len(pm), (inside, len(pm))
    assert worst_abs <= 5e-3, worst_abs
    flat_m = torch.cat([p.grad.cpu().reshape(-1) for p in pm.values()]).double()
    flat_o = torch.cat([p.grad.reshape(-1) for p in p64.values()]).double()
    cos = (flat_m @ flat_o / (flat_m.norm() * flat_o.norm())).item()
    assert cos >= 1 - 1e-7, (cos, worst_abs, worst_ratio)


@pytest.mark.parametrize('kind', ['rmsprop', 'sgd'])
def test_ten_optimiser_steps_vs_oracle(mfma_path, kind):
    """TEN optimiser steps (hg1 + DSNT + JS, batch 4, 128 px) next to the CPU oracle stepping torch.optim from the
    same weights on the same batch.  Training is a chaotic map: two correct fp32 implementations separate (RMSprop's
    first steps are sign-like, g / (sqrt(0.01 g^2) + eps), so rounding noise in small gradients moves weights by
    10 lr; ReLU masks flip).  So the HIP path is held against an ENVELOPE the oracle draws itself: the oracle in fp32
    vs the oracle in fp64 along the same ten steps.  Bars: coordinates within the north-star 1e-4 on the first
    forward; afterwards the HIP path may drift from the fp64 trajectory at most 10x as far as the fp32 oracle has
    drifted so far (floor 1e-4; one trajectory is one sample of a chaotic process — measured on MI355X, all three
    matrix-core paths: the same size as the fp32 oracle's own drift, e.g. RMSprop step 1: 4e-3..9e-3 vs 1.1e-2);
    loss likewise.  kind: RMSprop lr 2.5e-4 / SGD lr 0.05 momentum 0.9 (train.py:88-99,314-326)."""
    import copy
    from dsnt.model import build_mpii_pose_model
    from dsnt import optim
    from dsnt_oracle import model as omodel
    m = build_mpii_pose_model(base='hg1', output_strat='dsnt', reg='js')
    o32 = omodel.build_mpii_pose_model(base='hg1', output_strat='dsnt', reg='js')
    synthetic.fill_state_dict(m, seed=6)
    synthetic.fill_state_dict(o32, seed=6)
    o64 = copy.deepcopy(o32).double()
    m.cuda().train()
    o32.train()
    o64.train()
    x, target, mask = synthetic.batch(4, size=128, seed=3, mask_p=0.9)
    xd, td, kd = x.to(DEV), target.to(DEV), mask.to(DEV)
    m.hg._runner().ensure(torch.device(DEV))
    if kind == 'rmsprop':
        opt = optim.RMSprop(m, lr=2.5e-4)
        mk = lambda mod: torch.optim.RMSprop(mod.parameters(), lr=2.5e-4)
    else:
        opt = optim.SGD(m, lr=0.05, momentum=0.9)
        mk = lambda mod: torch.optim.SGD(mod.parameters(), lr=0.05, momentum=0.9)

    def run(mod, op, xx, tt, kk, steps=10):
        cs, ls = [], []
        for _ in range(steps):
            out = mod(xx)
            loss = mod.forward_loss(out, tt, kk)
            op.zero_grad()
            loss.backward()
            op.step()
            cs.append(out[-1].detach().cpu().double())
            ls.append(float(loss.detach()))
        return cs, ls
    c_hip, l_hip = run(m, opt, xd, td, kd)
    c32, l32 = run(o32, mk(o32), x, target, mask)
    c64, l64 = run(o64, mk(o64), x.double(), target.double(), mask.double())
    d_hip = [(a - b).abs().max().item() for a, b in zip(c_hip, c64)]
    d_32 = [(a - b).abs().max().item() for a, b in zip(c32, c64)]
    e_hip = [abs(a - b) / abs(b) for a, b in zip(l_hip, l64)]
    e_32 = [abs(a - b) / abs(b) for a, b in zip(l32, l64)]
    print('10-step drift vs the fp64 oracle (%s, %s):\n  coords HIP   %s\n  coords fp32  %s\n  loss HIP     %s\n  loss fp32    %s\n  loss %s'
          % (mfma_path, kind, ' '.join('%.1e' % v for v in d_hip), ' '.join('%.1e' % v for v in d_32),
             ' '.join('%.1e' % v for v in e_hip), ' '.join('%.1e' % v for v in e_32), ' '.join('%.4f' % v for v in l64)))
    assert d_hip[0] <= 1e-4, d_hip
    env_c, env_l = 0.0, 0.0
    for i in range(10):
        env_c, env_l = max(env_c, d_32[i]), max(env_l, e_32[i])
        assert d_hip[i] <= max(1e-4, 10 * env_c), (i, d_hip, d_32)
        assert e_hip[i] <= max(1e-5, 10 * env_l), (i, e_hip, e_32)
    assert l_hip[-1] < 0.7 * l_hip[0]                   # and it trains
    if kind != 'rmsprop':
        return
    # the optimiser state is checkpointable in torch's format and resumes bit-exactly (train.py:364,492)
    sd = opt.state_dict()
    assert len(sd['state']) == len(list(m.parameters())) and float(sd['state'][0]['step']) == 10.0
    before = m.hg.arena.params.clone()
    out = m(xd)
    loss = m.forward_loss(out, td, kd)
    opt.zero_grad()
    loss.backward()
    g = m.hg.arena.grads.clone()
    opt.step()
    after = m.hg.arena.params.clone()
    m.hg.arena.params.copy_(before)
    opt2 = optim.RMSprop(m, lr=1.0)
    opt2.load_state_dict(sd)
    m.hg.arena.grads.copy_(g)
    opt2.step()
    assert torch.equal(m.hg.arena.params, after)


def test_optimiser_state_interoperates_with_torch_optim_on_hg2():
    """The flat optimiser's state_dict numbers parameters as torch.optim.RMSprop(model.parameters()) would (train.py:314-326,
    364, 492).  hg2's arena is ordered bucket by bucket (stem, stack 0, stack 1) while model.parameters() runs
    hg.*, res.*, fc.*, score.*, ...: after three steps the state loads into a stock RMSprop over the same parameters
    with every square_avg on its own parameter, one more step of either optimiser gives the same weights, and the
    stock optimiser's state loads back."""
    from dsnt.model import build_mpii_pose_model
    from dsnt import optim
    m = build_mpii_pose_model(base='hg2', output_strat='dsnt', reg='js')
    synthetic.fill_state_dict(m, seed=8)
    m.cuda().train()
    x, target, mask = synthetic.batch(2, size=64, seed=4, mask_p=0.9)
    xd, td, kd = x.to(DEV), target.to(DEV), mask.to(DEV)
    m.hg._runner().ensure(torch.device(DEV))
    arena = m.hg.arena
    params = list(m.parameters())
    arena_order = [id(p) for _, p, _, _ in arena.slots]
    assert arena_order != [id(p) for p in params], 'hg2: arena order and model.parameters() order should differ'
    opt = optim.RMSprop(m, lr=2.5e-4)
    assert [id(p) for p in opt.param_groups[0]['params']] == [id(p) for p in params]

    def backward():
        out = m(xd)
        loss = m.forward_loss(out, td, kd)
        opt.zero_grad()
        loss.backward()
    for _ in range(3):
        backward()
        opt.step()
    sd = opt.state_dict()
    assert len(sd['state']) == len(params)
    # every entry has its parameter's shape and equals the arena's state of THAT parameter
    by_id = {id(p): name for name, p, _, _ in arena.slots}
    for i, p in enumerate(params):
        sq = sd['state'][i]['square_avg']
        assert tuple(sq.shape) == tuple(p.shape), (i, sq.shape, p.shape)
        assert torch.equal(sq, arena.logical(opt.flat_state, by_id[id(p)]))
    stock = torch.optim.RMSprop(params, lr=2.5e-4)
    stock.load_state_dict(sd)
    for i, p in enumerate(params):
        assert torch.equal(stock.state[p]['square_avg'], sd['state'][i]['square_avg'])
    # one more step with the same gradients: stock optimiser vs the flat one
    backward()
    before = arena.params.clone()
    grads = arena.grads.clone()
    opt.step()
    flat_after = arena.params.clone()
    arena.params.copy_(before)
    arena.grads.copy_(grads)
    stock.step()
    assert (arena.params - flat_after).abs().max().item() <= 1e-6 * flat_after.abs().max().item()
    # and back: the stock optimiser's checkpoint resumes in the flat one
    sd2 = stock.state_dict()
    opt2 = optim.RMSprop(m, lr=1.0)
    opt2.load_state_dict(sd2)
    for i, p in enumerate(params):
        assert torch.equal(arena.logical(opt2.flat_state, by_id[id(p)]), sd2['state'][i]['square_avg'])
    assert opt2.param_groups[0]['lr'] == 2.5e-4
    bad = {'state': {0: {'step': torch.tensor(1.0), 'square_avg': torch.zeros(3)}}, 'param_groups': sd['param_groups']}
    with pytest.raises(ValueError):
        opt2.load_state_dict(bad)
    # the order marker: what this class writes says so; a checkpoint re-tagged as ARENA order (what revisions before the
    # model-order numbering wrote, without a marker) is re-numbered on load — the permutation swaps equal-shaped parameters
    # of stack 0 and stack 1, which no shape check could catch; an unknown marker is refused
    assert sd['param_groups'][0]['dsnt_order'] == 'model' and 'dsnt_order' not in opt.param_groups[0]
    name_of = {id(p): name for name, p, _, _ in arena.slots}
    slot_pos = {name: i for i, (name, _, _, _) in enumerate(arena.slots)}
    old = {'state': {slot_pos[name_of[id(p)]]: sd['state'][i] for i, p in enumerate(params)},
           'param_groups': [dict(sd['param_groups'][0], dsnt_order='arena')]}
    opt3 = optim.RMSprop(m, lr=1.0)
    opt3.load_state_dict(old)
    assert torch.equal(opt3.flat_state, opt.flat_state) or all(
        torch.equal(arena.logical(opt3.flat_state, by_id[id(p)]), sd['state'][i]['square_avg']) for i, p in enumerate(params))
    with pytest.raises(ValueError, match='unknown parameter order'):
        opt3.load_state_dict({'state': sd['state'], 'param_groups': [dict(sd['param_groups'][0], dsnt_order='bucket')]})


def test_data_parallel_world1_nccl():
    """`parallel.DataParallel` on the GPU (world size 1, RCCL): the bucket markers of the traced backward list fire in
    backward-completion order for hourglass AND ResNet models, attaching changes no gradient bit, and the 1/world
    averaging rides in the publish kernel (`publish_scale`)."""
    import os
    import socket
    import torch.distributed as dist
    from dsnt.model import build_mpii_pose_model
    from dsnt import parallel
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        for base, size, nb in (('hg2', 256, 3), ('resnet18', 128, 1)):
            m = build_mpii_pose_model(base=base, output_strat='fc' if base == 'hg2' else 'dsnt', reg='js')
            synthetic.fill_state_dict(m, seed=0)
            m.cuda().train()
            x, t, k = synthetic.batch(2, size=size, seed=1, mask_p=0.9)
            x, t, k = x.to(DEV), t.to(DEV), k.to(DEV)

            def grads():
                for p in m.parameters():
                    p.grad = None
                m.forward_loss(m(x), t, k).backward()
                return torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
            g0 = grads()
            dp = parallel.DataParallel(m)
            fired = []
            hook = dp.runner.bucket_hook
            dp.runner.bucket_hook = lambda kk: (fired.append(kk), hook(kk))
            g1 = grads()
            assert torch.equal(g0, g1), base
            assert fired == list(range(nb - 1, -1, -1)) and dp.reducer.last_late == [], (base, fired)
            assert len(dp.extra) == (2 if base == 'hg2' else 0)
            dp.runner.arena.publish_scale = 0.5
            g2 = grads()
            n_arena = sum(p.numel() for _, p, _, _ in dp.runner.arena.slots)
            arena_ids = {id(p) for _, p, _, _ in dp.runner.arena.slots}
            mask_arena = torch.cat([torch.full((p.numel(),), id(p) in arena_ids) for p in m.parameters()]).to(DEV)
            assert torch.equal(g2[mask_arena], 0.5 * g0[mask_arena]) and torch.equal(g2[~mask_arena], g0[~mask_arena])
            assert int(mask_arena.sum()) == n_arena
            dp.detach()
            assert torch.equal(grads(), g0)
    finally:
        dist.destroy_process_group()


def test_finalisation_in_the_consumers_prologue_gives_the_same_step(monkeypatch):
    """BatchNorm finalisation of few-tile statistics folded into the prologue of the consuming launch (the default:
    dsnt_conv_fwd_pro / dsnt_bn_act_bwd_apply_pro, csrc/bn_pro.h) against the stand-alone finalise launches
    (DSNT_FUSE_FINALIZE=0): same loss / coordinates / gradients / running statistics to fp32 rounding (the fp64 sums are
    added in another order), fewer launches, bit-reproducible."""
    from dsnt.model import build_mpii_pose_model

    def step(fuse):
        monkeypatch.setenv('DSNT_FUSE_FINALIZE', fuse)
        m = build_mpii_pose_model(base='hg2', output_strat='dsnt', reg='js')
        synthetic.fill_state_dict(m, seed=0)
        m.cuda().train()
        x, t, k = synthetic.batch(4, size=128, seed=1, mask_p=0.9)
        res = []
        for _ in range(2):
            for p in m.parameters():
                p.grad = None
            out = m(x.to(DEV))
            loss = m.forward_loss(out, t.to(DEV), k.to(DEV))
            loss.backward()
            res.append((loss.item(), out[-1].detach().clone(), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()))
        prog = [p for p in m.hg._runner().programs.values() if p.training][0]
        rs = {n: b.clone() for n, b in m.named_buffers() if 'running' in n}
        return res, prog.n_fwd + prog.n_bwd, rs
    (a0, a1), n_a, rs_a = step('0')
    (b0, b1), n_b, rs_b = step('1')
    assert n_a - n_b >= 20, (n_a, n_b)
    assert abs(a0[0] - b0[0]) <= 1e-6 * abs(a0[0]) and (a0[1] - b0[1]).abs().max().item() <= 2e-6
    assert (a0[2] - b0[2]).norm().item() <= 1e-4 * a0[2].norm().item()
    for n in rs_a:
        assert (rs_a[n] - rs_b[n]).abs().max().item() <= 1e-5 * max(1.0, rs_a[n].abs().max().item()), n
    assert b0[0] == b1[0] and torch.equal(b0[1], b1[1]) and torch.equal(b0[2], b1[2])


def test_the_stacks_heads_and_losses_in_one_launch_each_give_the_same_step(monkeypatch):
    """The stacks' heat-maps leave the hourglass in one slab (dsnt.hourglass.StackedOutputs) and `forward` / `forward_loss` run the
    DSNT head and the loss of ALL stacks as one launch each; DSNT_STACKED_OUTPUTS=0 is the per-stack path of rounds 1-4 (one
    output tensor, one head, one loss node per stack — model.py:299-307 / 273-297 of the reference, literally).  Same
    coordinates and gradients bit for bit (the per-row arithmetic is the same kernels'), the loss to fp32 rounding (one sum over
    all rows instead of a sum of per-stack sums).  A caller that takes the list apart — the loss of the last stack only, or the
    stacks in another order — falls back to per-stack losses on views of the same tensors."""
    from dsnt.model import build_mpii_pose_model
    from dsnt.hourglass import StackedOutputs

    def step(stacked, pick):
        monkeypatch.setenv('DSNT_STACKED_OUTPUTS', stacked)
        m = build_mpii_pose_model(base='hg2', output_strat='dsnt', reg='js')
        synthetic.fill_state_dict(m, seed=0)
        m.cuda().train()
        x, t, k = synthetic.batch(4, size=128, seed=1, mask_p=0.9)
        for p in m.parameters():
            p.grad = None
        hg_out = m.forward_part1(x.to(DEV))
        assert isinstance(hg_out, list) and len(hg_out) == 2 and isinstance(hg_out, StackedOutputs) == (stacked == '1')
        out = m.forward_part2(hg_out)
        assert isinstance(out, list) and len(out) == 2 and out[0].shape == (4, 16, 2)
        assert len(m.heatmaps_array) == 2 and m.heatmaps_array[0].shape == (4, 16, 32, 32)
        loss = m.forward_loss(pick(out), t.to(DEV), k.to(DEV))
        loss.backward()
        return (loss.item(), torch.stack([o.detach() for o in out]).clone(), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone(),
                torch.stack([h.detach() for h in m.heatmaps_array]).clone())

    for pick, exact in ((lambda o: o, False), (lambda o: [o[1]], True), (lambda o: [o[1], o[0]], True)):
        a, b = step('0', pick), step('1', pick)
        assert torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
        assert abs(a[0] - b[0]) <= (0.0 if exact else 2e-7) * abs(a[0]), (a[0], b[0])
        assert torch.equal(a[2], b[2]), (a[2] - b[2]).abs().max().item()


def test_optimizer_kernels():
    """Flat RMSprop / SGD-momentum kernels vs torch.optim on identical gradients (train.py:314-326)."""
    from dsnt._lib import ptr, call
    n = 100003
    p0 = synthetic.tensor('op', (n,), seed=8)
    grads = [synthetic.tensor('og%d' % i, (n,), seed=8) * (0.1 + i) for i in range(3)]
    for kind in ('rmsprop', 'sgd'):
        pt = p0.clone().requires_grad_()
        opt = torch.optim.RMSprop([pt], lr=2.5e-4) if kind == 'rmsprop' else \
            torch.optim.SGD([pt], lr=0.2, momentum=0.9)
        pd = p0.to(DEV).clone()
        state = torch.zeros(n, device=DEV)
        for i, g_ in enumerate(grads):
            pt.grad = g_.clone()
            opt.step()
            gd = g_.to(DEV)
            if kind == 'rmsprop':
                call('dsnt_rmsprop_step', ptr(pd), ptr(gd), ptr(state), n, 2.5e-4, 0.99, 1e-8, 0.0, 1.0)
            else:
                call('dsnt_sgd_step', ptr(pd), ptr(gd), ptr(state), n, 0.2, 0.9, 0.0, 1.0, 1 if i == 0 else 0)
        assert (pd.cpu() - pt.detach()).abs().max().item() <= 2e-6, kind


def test_several_forwards_before_the_first_backward():
    """The reference's autograd has no 'one forward in flight' rule (model.py:273-307): two forward passes of the same
    shape, then one backward through the sum of their losses, must give the gradient of that sum.  Each waiting forward gets
    its own traced program (its own saved activations); a dropped graph hands its program back; beyond the cap it raises."""
    import gc
    from dsnt.model import build_mpii_pose_model
    m = build_mpii_pose_model(base='hg1', output_strat='dsnt', reg='js')
    synthetic.fill_state_dict(m, seed=5)
    m.cuda().train()
    xa, ta, ka = (v.to(DEV) for v in synthetic.batch(2, size=64, seed=6, mask_p=0.9))
    xb, tb, kb = (v.to(DEV) for v in synthetic.batch(2, size=64, seed=7, mask_p=0.9))
    runner = m.hg._runner()

    def grads():
        torch.cuda.synchronize()
        return torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()

    for p in m.parameters():
        p.grad = None
    m.forward_loss(m(xa), ta, ka).backward()            # one after the other, accumulated into p.grad
    m.forward_loss(m(xb), tb, kb).backward()
    want = grads()
    assert sum(1 for p in runner.programs.values() if p.training) == 1
    for p in m.parameters():
        p.grad = None
    la = m.forward_loss(m(xa), ta, ka)                  # both forwards first
    lb = m.forward_loss(m(xb), tb, kb)
    assert sum(1 for p in runner.programs.values() if p.training) == 2
    (la + lb).backward()
    got = grads()
    assert (got - want).abs().max().item() <= 2e-6 * want.abs().max().item()
    assert not any(p.in_flight for p in runner.programs.values())
    # a graph that is dropped without a backward frees its program: no third one is traced
    lc = m.forward_loss(m(xa), ta, ka)
    del lc
    gc.collect()
    ld = m.forward_loss(m(xb), tb, kb)
    ld.backward()
    assert sum(1 for p in runner.programs.values() if p.training) == 2
    # and there is a cap
    keep = []
    with pytest.raises(RuntimeError, match='waiting for their backward'):
        for _ in range(runner.MAX_IN_FLIGHT + 1):
            keep.append(m(xa))
    # a forward that records nothing (train mode under no_grad) while another waits for its backward takes a program of its
    # own instead of overwriting the saved activations: the pending backward gives the gradient of ITS forward, and no
    # program is left marked in flight afterwards (round-4 advice: the flag could stick for good)
    del keep
    gc.collect()
    for p in m.parameters():
        p.grad = None
    m.forward_loss(m(xa), ta, ka).backward()
    want_a = grads()
    for p in m.parameters():
        p.grad = None
    la = m.forward_loss(m(xa), ta, ka)
    with torch.no_grad():
        m(xb)
    la.backward()
    assert (grads() - want_a).abs().max().item() <= 2e-6 * want_a.abs().max().item()
    assert not any(p.in_flight for p in runner.programs.values())


@pytest.mark.parametrize('smooth', [True, False])
def test_backward_through_an_eval_mode_forward(smooth, mfma_path):
    """model.eval() + loss.backward() (the reference's autograd allows it: model.py:273-307 has no mode check; fine-tuning on
    frozen BatchNorm statistics): forward on the running statistics, backward with dx = gamma invstd dz and dgamma / dbeta as the
    two sums — every gradient against the CPU oracle, the running statistics untouched, and the train-mode program of the same
    shape unaffected (its own traced program)."""
    import contextlib
    import torch.nn as nn
    from dsnt.model import build_mpii_pose_model
    from dsnt_oracle import model as omodel
    with (_NoRelu() if smooth else contextlib.nullcontext()):
        m = build_mpii_pose_model(base='hg2', output_strat='dsnt', reg='js')
        o = omodel.build_mpii_pose_model(base='hg2', output_strat='dsnt', reg='js')
        if smooth:
            _NoRelu.strip(o)
        synthetic.fill_state_dict(m, seed=5)
        synthetic.fill_state_dict(o, seed=5)
        m.cuda().train()
        o.train()
        x, target, mask = synthetic.batch(4, size=128, seed=7, mask_p=0.8)
        # realistic running statistics first: one train-mode forward with momentum 1
        for mod in list(m.modules()) + list(o.modules()):
            if isinstance(mod, nn.BatchNorm2d):
                mod.momentum = 1.0
        with torch.no_grad():
            m(x.to(DEV)); o(x)
        m.eval(); o.eval()
        stats = {n: b.detach().clone() for n, b in m.named_buffers() if 'running' in n}
        outs = m(x.to(DEV))
        assert all(t.requires_grad for t in outs)
        loss = m.forward_loss(outs, target.to(DEV), mask.to(DEV))
        loss.backward()
        outs_o = o(x)
        loss_o = o.forward_loss(outs_o, target, mask)
        loss_o.backward()
        for a, b in zip(outs, outs_o):
            assert (a.detach().cpu() - b.detach()).abs().max().item() <= 1e-4
        assert abs(loss.item() - loss_o.item()) <= 1e-4 * max(1.0, abs(loss_o.item()))
        # (the eval + backward program itself runs on bf16x6 under both split settings; under 'f16x3' the running statistics come
        # from an fp16x3 train-mode forward, 1e-7 away from the oracle's, and one max-pool near-tie of this input flips: 4e-3 on
        # the parameters in front of it — the envelope of test_hg8_every_gradient_vs_oracle_on_the_smooth_network)
        tight = 5e-3 if mfma_path == 'f16x3' else 1e-3
        worst = _grads_close(m, o, tight if smooth else 0.2, 'hg2 eval')
        pm, po = dict(m.named_parameters()), dict(o.named_parameters())
        flat_m = torch.cat([p.grad.cpu().reshape(-1) for p in pm.values()]).double()
        flat_o = torch.cat([p.grad.reshape(-1) for p in po.values()]).double()
        cos = (flat_m @ flat_o / (flat_m.norm() * flat_o.norm())).item()
        assert cos >= ((1 - 1e-5 if mfma_path == 'f16x3' else 1 - 1e-7) if smooth else 0.999), (cos, worst)
        for n, b in m.named_buffers():
            if 'running' in n:
                assert torch.equal(b, stats[n]), n               # an eval-mode forward updates nothing
        # eval mode under no_grad stays the forward-only program and gives the same coordinates
        with torch.no_grad():
            ev = m(x.to(DEV))
        assert not ev[-1].requires_grad and (ev[-1] - outs[-1].detach()).abs().max().item() <= 1e-4    # (fp16x3 vs bf16x6 products)
        progs = m.hg._runner().programs
        assert sum(1 for p in progs.values() if p.record and not p.training) == 1
        assert sum(1 for p in progs.values() if not p.record) == 1
        # ... and the train-mode step of the same shape is its own program, still right
        m.train(); o.train()
        m.zero_grad(); o.zero_grad()
        l1 = m.forward_loss(m(x.to(DEV)), target.to(DEV), mask.to(DEV))
        l1.backward()
        l1o = o.forward_loss(o(x), target, mask)
        l1o.backward()
        assert abs(l1.item() - l1o.item()) <= 1e-4 * max(1.0, abs(l1o.item()))
        _grads_close(m, o, 1e-3 if smooth else 0.2, 'hg2 train after eval')


def test_surface_and_errors():
    from dsnt.model import build_mpii_pose_model
    m = build_mpii_pose_model(base='hg', dilate=2, truncate=1)       # resnet kwargs filtered out
    assert m.output_strat == 'gauss' and m.hg.num_stacks == 2 and m.heatmap_size == 64
    assert m.image_specs.size == 256 and m.image_specs.subtract_mean
    assert build_mpii_pose_model(base='hg8', output_strat='dsnt').hg.num_stacks == 8
    for bad in ('vgg', 'hgx'):
        with pytest.raises(Exception, match='unsupported base model type'):
            build_mpii_pose_model(base=bad)
    m = build_mpii_pose_model(base='hg1', output_strat='dsnt')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.zeros(2, 3, 64, 64))
    m.cuda()
    with pytest.raises(RuntimeError, match='multiple of 64'):
        m(torch.zeros(2, 3, 96, 96, device=DEV))
    # inference.py:33-48: forward_part1 -> last stack -> forward_part2 on a bare 4-D tensor
    m.eval()
    with torch.no_grad():
        hm = m.forward_part1(torch.rand(2, 3, 64, 64, device=DEV))
        assert isinstance(hm, list) and hm[-1].shape == (2, 16, 16, 16)
        out = m.forward_part2(hm[-1][:1])
        assert len(out) == 1 and out[0].shape == (1, 16, 2)
    sd = m.state_dict()
    assert sd['hg.conv1.weight'].shape == (64, 3, 7, 7)
    assert sd['hg.layer1.0.conv2.weight'].shape == (64, 64, 3, 3)
    m2 = build_mpii_pose_model(base='hg1', output_strat='dsnt')
    m2.load_state_dict(sd)
    m2.cuda().eval()
    x = torch.rand(2, 3, 64, 64, device=DEV)
    with torch.no_grad():
        assert torch.equal(m(x)[0], m2(x)[0])
