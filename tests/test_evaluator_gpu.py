"""Device PCKh evaluator vs the reference's known answers (tests/test_evaluator.py:8-39) and the
golden vector; flat optimiser on a real model vs torch.optim on the oracle."""
import pytest
import torch

from dsnt import synthetic
import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_pckh_known_answers():
    from dsnt.evaluator import PCKhEvaluator
    d = PCKhEvaluator.calculate_pckh_distance(torch.tensor([951.84, 580.64]), torch.tensor([804.0, 711]), 117.962)
    assert abs(float(d) - 1.6709) <= 1e-4
    ev = PCKhEvaluator(threshold=0.5)
    pred = torch.tensor([[[951.84, 580.64]], [[317.76, 406.75]], [[float('inf')] * 2]], device=DEV)
    target = torch.tensor([[[804.0, 711]], [[317, 412]], [[float('nan')] * 2]], device=DEV)
    ev.add(pred, target, torch.tensor([[1.0], [1], [0]], device=DEV),
           torch.tensor([117.962, 44.046, 78.481], device=DEV))
    assert ev.meters['all'].value()[0] == 0.5
    ev.reset()
    assert ev.meters['all'].value()[0] != ev.meters['all'].value()[0]       # NaN when empty


def test_pckh_golden_and_oracle():
    from dsnt.evaluator import PCKhEvaluator
    from dsnt_oracle.evaluator import PCKhEvaluator as OracleEval
    g = gu.load('pckh')
    _, target, mask = synthetic.batch(64, size=8, seed=3, mask_p=0.85)
    pred = target + synthetic.tensor('pckh.noise', (64, 16, 2), seed=3, scale=0.15)
    head, m, b = synthetic.pckh_inputs(64)
    ev = PCKhEvaluator(0.5)
    ev.add_normalized(pred.to(DEV), target, mask, head, m, b)      # back-projection on the device
    oe = OracleEval(0.5)
    oe.add(torch.bmm(pred.double(), m) + b, torch.bmm(target.double(), m) + b, mask, head)
    for k in ev.meters:
        assert abs(ev.meters[k].value()[0] - float(g[k])) <= 1e-6, k       # identical PCKh
        assert abs(ev.meters[k].value()[0] - oe.meters[k].value()[0]) <= 1e-6, k


def test_identical_pckh_given_identical_weights():
    """north_star: identical PCKh@0.5 for the HIP path and the CPU path on the same weights."""
    from dsnt.model import build_mpii_pose_model
    from dsnt.evaluator import PCKhEvaluator
    from dsnt_oracle import model as omodel
    from dsnt_oracle.evaluator import PCKhEvaluator as OracleEval
    m = build_mpii_pose_model(base='hg2', output_strat='dsnt', reg='js')
    o = omodel.build_mpii_pose_model(base='hg2', output_strat='dsnt', reg='js')
    synthetic.fill_state_dict(m, seed=0)
    synthetic.fill_state_dict(o, seed=0)
    m.cuda().train()
    o.train()
    x, target, mask = synthetic.batch(8, size=128, seed=6, mask_p=0.9)
    coords = m.compute_coords(m(x.to(DEV)))
    with torch.no_grad():
        coords_o = o.compute_coords(o(x))
    assert (coords - coords_o).abs().max().item() <= 1e-4
    head, tm, tb = synthetic.pckh_inputs(8)
    # synthetic "ground truth" near the predictions so that hits and misses both occur
    gt = coords_o + synthetic.tensor('gt.noise', (8, 16, 2), seed=6, scale=0.2)
    ev, oe = PCKhEvaluator(0.5), OracleEval(0.5)
    ev.add_normalized(coords.to(DEV), gt, mask, head, tm, tb)
    oe.add(torch.bmm(coords_o.double(), tm) + tb, torch.bmm(gt.double(), tm) + tb, mask, head)
    for k in ev.meters:
        a, b = ev.meters[k].value()[0], oe.meters[k].value()[0]
        assert a == b or (a != a and b != b), (k, a, b)
    assert 0.05 < ev.meters['all'].value()[0] < 0.95


@pytest.mark.parametrize('kind', ['rmsprop', 'sgd'])
def test_flat_optimizer_on_model(kind):
    """dsnt.optim (one kernel over the arena) == torch.optim on the same gradients, and it works
    as a drop-in `optimizer` in the train.py step order (zero_grad -> backward -> step)."""
    from dsnt.model import build_mpii_pose_model
    from dsnt import optim
    m = build_mpii_pose_model(base='hg1', output_strat='dsnt', reg='js')
    synthetic.fill_state_dict(m, seed=0)
    m.cuda().train()
    x, t, k = synthetic.batch(2, size=128, seed=1)
    x, t, k = x.to(DEV), t.to(DEV), k.to(DEV)
    out = m(x)
    m.forward_loss(out, t, k).backward()
    ref_params = [p.detach().clone().contiguous().requires_grad_() for p in m.parameters()]
    for rp, p in zip(ref_params, m.parameters()):
        rp.grad = p.grad.detach().clone().contiguous()
    if kind == 'rmsprop':
        opt, ropt = optim.RMSprop(m, lr=2.5e-4), torch.optim.RMSprop(ref_params, lr=2.5e-4)
    else:
        opt, ropt = optim.SGD(m, lr=0.2, momentum=0.9), torch.optim.SGD(ref_params, lr=0.2, momentum=0.9)
    opt.step()
    ropt.step()
    for rp, p in zip(ref_params, m.parameters()):
        assert (p.detach() - rp.detach()).abs().max().item() <= 1e-6
    # second step through the full loop; lr scheduler sees param_groups
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1], gamma=0.1)
    opt.zero_grad()
    m.forward_loss(m(x), t, k).backward()
    opt.step()
    sched.step()
    assert abs(opt.param_groups[0]['lr'] - (2.5e-5 if kind == 'rmsprop' else 0.02)) < 1e-12
    assert all(torch.isfinite(p).all() for p in m.parameters())
