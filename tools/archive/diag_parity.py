"""Print parity diagnostics (HIP path vs oracle / golden) without asserting."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
from dsnt import synthetic
import golden_util as gu
DEV = 'cuda:0'


def rel(a, b, floor=1e-12):
    return (a - b).abs().max().item() / max(b.abs().max().item(), floor)


def block(kind):
    from dsnt import hourglass as dhg
    from dsnt_oracle import hourglass as ohg
    hw = 16 if kind == 'bottleneck' else 32
    mk = (lambda mod: mod.Bottleneck(256, 128)) if kind == 'bottleneck' else (lambda mod: mod.Hourglass(mod.Bottleneck, 1, 128, 4))
    m, o = mk(dhg), mk(ohg)
    synthetic.fill_state_dict(m, seed=5); synthetic.fill_state_dict(o, seed=5)
    m.cuda().train(); o.train()
    x = synthetic.tensor(kind + '.x', (2, 256, hw, hw), seed=5)
    gy = synthetic.tensor(kind + '.gy', (2, 256, hw, hw), seed=5)
    xd = x.to(DEV).requires_grad_(); y = m(xd); y.backward(gy.to(DEV))
    xo = x.clone().requires_grad_(); yo = o(xo); yo.backward(gy)
    print(kind, 'y rel', rel(y.detach().cpu(), yo.detach()), 'dx rel', rel(xd.grad.cpu(), xo.grad))
    gmax = max(q.grad.abs().max().item() for q in o.parameters())
    worst = sorted(((p.grad.cpu() - q.grad).abs().max().item() / max(q.grad.abs().max().item(), 1e-3 * gmax), n)
                   for (n, p), (_, q) in zip(m.named_parameters(), o.named_parameters()))[-5:]
    print('  worst param grads', worst)
    wb = sorted(((b.cpu() - c).abs().max().item(), n) for (n, b), (_, c) in zip(m.named_buffers(), o.named_buffers()) if 'running' in n)[-3:]
    print('  worst buffers', wb)


def e2e(base, size, reg, bs=2, seed=0):
    from dsnt.model import build_mpii_pose_model
    from dsnt_oracle import model as omodel
    m = build_mpii_pose_model(base=base, output_strat='dsnt', reg=reg)
    o = omodel.build_mpii_pose_model(base=base, output_strat='dsnt', reg=reg)
    synthetic.fill_state_dict(m, seed=seed); synthetic.fill_state_dict(o, seed=seed)
    m.cuda().train(); o.train()
    x, t, k = synthetic.batch(bs, size=size, seed=1, mask_p=0.9)
    outs = m(x.to(DEV)); loss = m.forward_loss(outs, t.to(DEV), k.to(DEV)); loss.backward()
    oo = o(x); lo = o.forward_loss(oo, t, k); lo.backward()
    print(base, size, reg, 'coords err', [(a.detach().cpu() - b.detach()).abs().max().item() for a, b in zip(outs, oo)],
          'loss', loss.item(), lo.item())
    gmax = max(q.grad.abs().max().item() for q in o.parameters())
    worst = sorted(((p.grad.cpu() - q.grad).abs().max().item() / max(q.grad.abs().max().item(), 1e-3 * gmax), n)
                   for (n, p), (_, q) in zip(m.named_parameters(), o.named_parameters()))[-6:]
    print('  worst param grads', worst)
    torch.optim.SGD(m.parameters(), lr=0.01).step(); torch.optim.SGD(o.parameters(), lr=0.01).step()
    m.eval(); o.eval()
    with torch.no_grad():
        print('  eval coords err', (m(x.to(DEV))[-1].cpu() - o(x)[-1]).abs().max().item())


if __name__ == '__main__':
    block('bottleneck'); block('hourglass')
    e2e('hg1', 128, 'none'); e2e('hg2', 128, 'js'); e2e('hg2', 256, 'js'); e2e('hg2', 128, 'js', bs=4, seed=3)
