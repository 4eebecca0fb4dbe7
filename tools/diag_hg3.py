import os, sys
import torch, torch.nn as nn, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
from dsnt import synthetic
from dsnt import hourglass as dhg
from dsnt_oracle import hourglass as ohg
DEV = 'cuda:0'
def rel(a, b, floor=1e-12):
    return (a - b).abs().max().item() / max(b.abs().max().item(), floor)

class Comp(dhg.TapeModule):
    def __init__(self, kind):
        super().__init__()
        self.kind = kind
        self.b1 = dhg.Bottleneck(256, 128); self.b2 = dhg.Bottleneck(256, 128)
        self.out_channels = 256
    def trace(self, t, x, P):
        k = self.kind
        if k == 'A': return self.b1.trace(t, t.maxpool2(x), P)
        if k == 'B': return self.b2.trace(t, self.b1.trace(t, x, P), P)
        if k == 'C': return t.upsample2_add(x, self.b1.trace(t, t.maxpool2(x), P))
        if k == 'D': return t.upsample2_add(self.b2.trace(t, x, P), self.b1.trace(t, t.maxpool2(x), P))
        if k == 'E': return self.b2.trace(t, self.b1.trace(t, t.maxpool2(x), P), P)
class CompO(nn.Module):
    def __init__(self, kind):
        super().__init__()
        self.kind = kind
        self.b1 = ohg.Bottleneck(256, 128); self.b2 = ohg.Bottleneck(256, 128)
    def forward(self, x):
        k = self.kind
        if k == 'A': return self.b1(F.max_pool2d(x, 2, 2))
        if k == 'B': return self.b2(self.b1(x))
        if k == 'C': return x + F.interpolate(self.b1(F.max_pool2d(x, 2, 2)), scale_factor=2)
        if k == 'D': return self.b2(x) + F.interpolate(self.b1(F.max_pool2d(x, 2, 2)), scale_factor=2)
        if k == 'E': return self.b2(self.b1(F.max_pool2d(x, 2, 2)))

def run(kind, N, hw):
    m, o = Comp(kind), CompO(kind)
    synthetic.fill_state_dict(m, seed=5); synthetic.fill_state_dict(o, seed=5)
    m.cuda().train(); o.train()
    x = synthetic.tensor('x', (N, 256, hw, hw), seed=5)
    xd = x.to(DEV).requires_grad_(); y = m(xd)
    xo = x.clone().requires_grad_(); yo = o(xo)
    gy = synthetic.tensor('gy', tuple(yo.shape), seed=5)
    y.backward(gy.to(DEV)); yo.backward(gy)
    gmax = max(q.grad.abs().max().item() for q in o.parameters() if q.grad is not None)
    errs = sorted(((p.grad.cpu() - q.grad).abs().max().item() / max(q.grad.abs().max().item(), 1e-3 * gmax), n)
            for (n, p), (_, q) in zip(m.named_parameters(), o.named_parameters()) if q.grad is not None)
    print(kind, N, hw, 'y %.1e dx %.1e' % (rel(y.detach().cpu(), yo.detach()), rel(xd.grad.cpu(), xo.grad)), 'worst', ['%s %.1e' % (n, e) for e, n in errs[-4:]])

for kind in 'ABCDE':
    run(kind, 8, 8)
    run(kind, 2, 32)
    run(kind, 2, 16)
