"""The fp16x3 default is only safe while every operand bound dominates its operand (an fp16 overflow would be
fatal: csrc/conv.hip `pow2_scale` leaves 4x head-room and nothing else).  These tests walk a FULL-SIZE train step
launch by launch — hg2 batch 32 and hg8 batch 16 at 256 px (BASELINE configs 3 and 5), hg1 batch 32 (config 2) —
and, right before every fp16x3 launch, hold each bound slot against the tensor it must dominate AS IT IS AT THAT
MOMENT (gradient buffers are donated and accumulated into, so the end-of-step content is not what a consumer read):

* A operands seen through a train-mode BatchNorm(+ReLU): max|relu?(x scale + shift)| <= the analytic bound; raw A
  operands (skip projections, `lin` convolutions): max|x| <= the slot the producing launch's epilogue raised;
* weights (forward layout and the re-packed data-gradient layout): max|w| <= the bound the prep launch wrote;
* gradient operands (dY of data-gradient and weight-gradient launches): max|dY| <= the slot its writers raised.

Also: loss and every gradient finite, the device-side non-finite guard stays down, and a poisoned loss raises it,
blocks the optimiser and is reported by the asynchronous poll (train.py:360-371).
"""
import pytest
import torch

from dsnt import synthetic

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _train_step_with_probe(base, reg, batch):
    from dsnt.model import build_mpii_pose_model
    m = build_mpii_pose_model(base=base, output_strat='dsnt', reg=reg)
    synthetic.fill_state_dict(m, seed=0)
    m.to(DEV).train()
    x, t, k = synthetic.batch(batch, size=256, seed=1, mask_p=0.9)
    x, t, k = x.to(DEV), t.to(DEV), k.to(DEV)
    runner = m.hg._runner()
    # step 1 un-probed (traces the programs), step 2 probed
    loss = m.forward_loss(m(x), t, k)
    loss.backward()
    prog = [p for p in runner.programs.values() if p.training][0]
    uses = {}
    for entry, info in prog.tape.f16_uses:
        uses.setdefault(id(entry), []).append(info)
    checked = {'fwd': 0, 'wgrad': 0, 'dgrad': 0, 'bwd1': 0}
    worst = {'a': 0.0, 'w': 0.0, 'g': 0.0}
    bad = []

    def amax_a(u):
        v = u['x'] * u['sc'] + u['sh'] if u['sc'] is not None else u['x'].clone()      # raw operand: the producer's amax
        if u['relu']:
            v = v.clamp_(min=0)
        return float(v.abs().max())

    def probe(entry):
        infos = uses.get(id(entry))
        if not infos:
            return
        torch.cuda.synchronize()
        for u in infos:
            checked[u['kind']] += 1
            if 'x' in u:
                got, bound = amax_a(u), float(u['a_bound'].max())
                worst['a'] = max(worst['a'], got / bound)
                if not got <= bound:
                    bad.append((u['kind'], u['name'], 'A', got, bound))
            if 'w' in u:
                got, bound = float(u['w'].abs().max()), float(u['w_bound'].max())
                worst['w'] = max(worst['w'], got / bound)
                if not got <= bound:
                    bad.append((u['kind'], u['name'], 'W', got, bound))
            if 'g_apply' in u:
                # dsnt_conv1x1_bwd_f16x3 forms dL/dy = scale (dz - coef0 - (y - mean) invstd coef1) in registers: the analytic
                # bound dsnt_bn_bwd_finalize_bound left must dominate it
                a = u['g_apply']
                Cc = a['scale'].numel()
                dy = a['scale'] * (a['dz'].view(-1, Cc) - a['coef'][:Cc] -
                                   (a['y'].view(-1, Cc) - a['mean']) * a['invstd'] * a['coef'][Cc:])
                got, bound = float(dy.abs().max()), float(u['g_bound'].max())
                worst['gf'] = max(worst.get('gf', 0.0), got / max(bound, 1e-30))
                if not got <= bound:
                    bad.append((u['kind'], u['name'], 'dY (folded BatchNorm backward)', got, bound))
            if 'g' in u:
                got, bound = float(u['g'].abs().max()), float(u['g_bound'].max())
                worst['g'] = max(worst['g'], got / max(bound, 1e-30))
                if not got <= bound:
                    bad.append((u['kind'], u['name'], 'dY', got, bound))

    for p in m.parameters():
        p.grad = None
    runner.probe = probe
    try:
        out = m(x)
        loss = m.forward_loss(out, t, k)
        loss.backward()
    finally:
        runner.probe = None
    torch.cuda.synchronize()
    return m, loss, checked, worst, bad, prog


@pytest.mark.parametrize('base,reg,batch,min_uses', [('hg2', 'js', 32, (40, 30, 30)), ('hg8', 'js', 16, (120, 120, 120)),
                                                      ('hg1', 'none', 32, (20, 15, 15))])
def test_every_fp16x3_bound_dominates_its_operand(base, reg, batch, min_uses):
    m, loss, checked, worst, bad, prog = _train_step_with_probe(base, reg, batch)
    assert not bad, bad[:8]
    assert checked['fwd'] >= min_uses[0] and checked['wgrad'] + checked['bwd1'] >= min_uses[1] and \
        checked['dgrad'] + checked['bwd1'] >= min_uses[2] and checked['bwd1'] >= 10, checked
    # the analytic bound of a folded BatchNorm backward is loose (|xhat| <= sqrt(M)) but not absurd: within 2^12 of the operand
    assert 2.0 ** -12 < worst['gf'] <= 1.0, worst
    # exact bounds (weights, gradients) are tight; the analytic BatchNorm bound is loose but must not be absurd
    assert worst['w'] == 1.0 and 0.0 < worst['g'] <= 1.0 and 0.0 < worst['a'] <= 1.0, worst
    assert torch.isfinite(loss).item()
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert torch.isfinite(g).all().item() and float(g.norm()) > 0
    # every bound slot of the program is finite and non-negative
    t = prog.tape
    if t._amax_buf is not None:
        assert torch.isfinite(t._amax_buf[:t._amax_used]).all().item() and float(t._amax_buf[:t._amax_used].min()) >= 0.0


def test_nonfinite_guard_blocks_the_update_and_reports_asynchronously():
    from dsnt.model import build_mpii_pose_model
    from dsnt import optim
    from dsnt.guard import NanGuard, NonFiniteError
    m = build_mpii_pose_model(base='hg1', output_strat='dsnt', reg='js')
    synthetic.fill_state_dict(m, seed=0)
    m.to(DEV).train()
    x, t, k = synthetic.batch(2, size=128, seed=1)
    x, t, k = x.to(DEV), t.to(DEV), k.to(DEV)
    guard = NanGuard(torch.device(DEV))
    m.hg._runner().ensure(torch.device(DEV))
    opt = optim.RMSprop(m, lr=2.5e-4, guard=guard)

    def step(poison=None):
        out = m(x)
        loss = m.forward_loss(out, t, k)
        if poison is not None:
            loss = loss * poison
        guard.check(loss)
        opt.zero_grad()
        loss.backward()
        opt.step()
        guard.poll()

    before = m.hg.arena.params.clone()
    step()
    step()
    guard.sync()                                        # finite steps: flag down, weights move
    after = m.hg.arena.params.clone()
    assert not torch.equal(before, after) and guard.flag.tolist() == [0, 0]
    step(poison=float('nan'))                           # NaN loss: update skipped, poll reports at the latest one step on
    assert torch.equal(m.hg.arena.params, after)
    with pytest.raises(NonFiniteError, match='non-finite loss'):
        step()
        torch.cuda.synchronize()
        guard.poll()
    assert torch.equal(m.hg.arena.params, after)        # still blocked
    guard.reset()
    # a finite loss with a non-finite GRADIENT element: that element is skipped, the flag says so
    m.zero_grad()
    loss = m.forward_loss(m(x), t, k)
    loss.backward()
    m.hg.arena.grads[5] = float('inf')
    p5 = float(m.hg.arena.params[5])
    opt.step()
    assert float(m.hg.arena.params[5]) == p5 and not torch.equal(m.hg.arena.params, after)
    with pytest.raises(NonFiniteError, match='non-finite gradient'):
        guard.sync()
    assert torch.isfinite(m.hg.arena.params).all().item()


def test_eval_mode_fp16x3_bounds_come_from_the_producers():
    """Eval mode has no batch statistics to bound relu(bn(x)) analytically: the launch that produces x forms the
    consumer's operand from the (running-statistics) BatchNorm vectors and leaves its maximum (dsnt_out_bounds.amax_bn).
    Walk a batch-8 eval forward of hg2: every fp16x3 launch's A bound equals max|operand| at that moment (to fp32
    rounding) and most large convolutions are on fp16x3."""
    from dsnt.model import build_mpii_pose_model
    m = build_mpii_pose_model(base='hg2', output_strat='dsnt', reg='js')
    synthetic.fill_state_dict(m, seed=0)
    m.to(DEV).eval()
    x, _, _ = synthetic.batch(8, size=256, seed=1)
    x = x.to(DEV)
    runner = m.hg._runner()
    with torch.no_grad():
        ref = [o.clone() for o in m(x)]
    prog = [p for p in runner.programs.values() if not p.training][0]
    uses = {}
    for entry, info in prog.tape.f16_uses:
        uses.setdefault(id(entry), []).append(info)
    seen, bad = [0], []

    def probe(entry):
        for u in uses.get(id(entry), ()):
            torch.cuda.synchronize()
            v = u['x'].double()
            if u['sc'] is not None:
                v = v * u['sc'].double() + u['sh'].double()
                if u['relu']:
                    v = v.clamp_(min=0)
            got, bound = float(v.abs().max()), float(u['a_bound'].max())
            seen[0] += 1
            if not (got <= bound * (1 + 1e-6) and bound <= got * (1 + 1e-5) + 1e-30):
                bad.append((u['name'], got, bound))
            if not float(u['w'].abs().max()) <= float(u['w_bound'].max()):
                bad.append((u['name'], 'W'))

    runner.probe = probe
    try:
        with torch.no_grad():
            out = m(x)
    finally:
        runner.probe = None
    assert not bad, bad[:6]
    assert seen[0] >= 20, seen           # batch 8: the 64x64 level (the 32x32 level is below the split-precision row threshold)
    for a, b in zip(out, ref):
        assert torch.equal(a, b)           # probed (Python replay) and un-probed (C replay) runs agree bit for bit
