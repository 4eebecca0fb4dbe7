import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
from dsnt import synthetic
from dsnt import hourglass as dhg
from dsnt_oracle import hourglass as ohg
DEV = 'cuda:0'

def rel(a, b, floor=1e-12):
    return (a - b).abs().max().item() / max(b.abs().max().item(), floor)

def run(depth, hw, N=2):
    m = dhg.Hourglass(dhg.Bottleneck, 1, 128, depth); o = ohg.Hourglass(ohg.Bottleneck, 1, 128, depth)
    synthetic.fill_state_dict(m, seed=5); synthetic.fill_state_dict(o, seed=5)
    m.cuda().train(); o.train()
    x = synthetic.tensor('x', (N, 256, hw, hw), seed=5); gy = synthetic.tensor('gy', (N, 256, hw, hw), seed=5)
    xd = x.to(DEV).requires_grad_(); y = m(xd); y.backward(gy.to(DEV))
    xo = x.clone().requires_grad_(); yo = o(xo); yo.backward(gy)
    gmax = max(q.grad.abs().max().item() for q in o.parameters())
    errs = [((p.grad.cpu() - q.grad).abs().max().item() / max(q.grad.abs().max().item(), 1e-3 * gmax), n)
            for (n, p), (_, q) in zip(m.named_parameters(), o.named_parameters())]
    print('depth', depth, 'hw', hw, 'N', N, 'y', rel(y.detach().cpu(), yo.detach()), 'dx', rel(xd.grad.cpu(), xo.grad))
    for e, n in errs:
        if e > 1e-4 and 'bias' not in n.split('.')[-2:][0] or (e > 1e-3):
            print('   ', n, '%.2e' % e)

for depth, hw in [(1, 4), (1, 8), (1, 16), (2, 8), (2, 16), (1, 32)]:
    run(depth, hw)
run(1, 8, N=8)
