import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
from dsnt import synthetic
from dsnt import hourglass as dhg
from dsnt._lib import ptr, call
from dsnt_oracle import hourglass as ohg
DEV = 'cuda:0'
def rel(a, b, floor=1e-12):
    return (a - b).abs().max().item() / max(b.abs().max().item(), floor)
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous()

for (N, H, W, Cc) in [(2, 32, 32, 256), (8, 8, 8, 256), (2, 16, 16, 256)]:
    x = synthetic.tensor('px', (N, Cc, H, W), seed=5).requires_grad_()
    y_ref = F.max_pool2d(x, 2, stride=2); gy = synthetic.tensor('pg', tuple(y_ref.shape), seed=5); y_ref.backward(gy)
    xd = nhwc(x.detach()).to(DEV); y = torch.empty(N, H // 2, W // 2, Cc, device=DEV)
    idx = torch.empty(N, H // 2, W // 2, Cc, dtype=torch.uint8, device=DEV)
    call('dsnt_maxpool2_fwd', ptr(xd), ptr(y), ptr(idx), N, H, W, Cc)
    dx = torch.empty_like(xd); gyd = nhwc(gy).to(DEV)
    call('dsnt_maxpool2_bwd', ptr(gyd), ptr(idx), ptr(dx), 0, N, H, W, Cc)
    go = synthetic.tensor('ug', (N, Cc, H, W), seed=6); god = nhwc(go).to(DEV)
    low = torch.zeros(N, Cc, H // 2, W // 2, requires_grad=True)
    (F.interpolate(low, scale_factor=2, mode='nearest') * go).sum().backward()
    dl = torch.empty(N, H // 2, W // 2, Cc, device=DEV)
    call('dsnt_upsample2_bwd', ptr(god), ptr(dl), 0, N, H, W, Cc)
    print((N, H, W, Cc), 'pool fwd', rel(y.cpu().permute(0, 3, 1, 2), y_ref.detach()), 'pool bwd', rel(dx.cpu().permute(0, 3, 1, 2), x.grad),
          'ups bwd', rel(dl.cpu().permute(0, 3, 1, 2), low.grad))

def bneck(N, hw):
    m = dhg.Bottleneck(256, 128); o = ohg.Bottleneck(256, 128)
    synthetic.fill_state_dict(m, seed=5); synthetic.fill_state_dict(o, seed=5)
    m.cuda().train(); o.train()
    x = synthetic.tensor('x', (N, 256, hw, hw), seed=5); gy = synthetic.tensor('gy', (N, 256, hw, hw), seed=5)
    xd = x.to(DEV).requires_grad_(); y = m(xd); y.backward(gy.to(DEV))
    xo = x.clone().requires_grad_(); yo = o(xo); yo.backward(gy)
    gmax = max(q.grad.abs().max().item() for q in o.parameters())
    errs = [((p.grad.cpu() - q.grad).abs().max().item() / max(q.grad.abs().max().item(), 1e-3 * gmax), n)
            for (n, p), (_, q) in zip(m.named_parameters(), o.named_parameters())]
    print('bottleneck N', N, 'hw', hw, 'y', rel(y.detach().cpu(), yo.detach()), 'dx', rel(xd.grad.cpu(), xo.grad), 'worst', sorted(errs)[-3:])
for N, hw in [(2, 16), (8, 4), (8, 8), (2, 32), (2, 8), (4, 16)]:
    bneck(N, hw)
