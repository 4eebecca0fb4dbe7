import os, sys
os.environ['DSNT_NO_DONATE'] = '1'
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
    def __init__(self):
        super().__init__()
        self.b1 = dhg.Bottleneck(256, 128); self.b2 = dhg.Bottleneck(256, 128); self.out_channels = 256
    def trace(self, t, x, P):
        return self.b2.trace(t, self.b1.trace(t, x, P), P)
class CompO(nn.Module):
    def __init__(self):
        super().__init__()
        self.b1 = ohg.Bottleneck(256, 128); self.b2 = ohg.Bottleneck(256, 128)
    def forward(self, x):
        return self.b2(self.b1(x))
N, hw = 2, 32
m, o = Comp(), CompO()
synthetic.fill_state_dict(m, seed=5); synthetic.fill_state_dict(o, seed=5)
m.cuda().train(); o.train()
saved = {}
def hook(name):
    def f(mod, inp, out):
        out.retain_grad(); saved[name] = out
    return f
for bn, b in (('b1', o.b1), ('b2', o.b2)):
    b.conv1.register_forward_hook(hook(bn + '.c1')); b.conv2.register_forward_hook(hook(bn + '.c2')); b.register_forward_hook(hook(bn + '.c3'))
x = synthetic.tensor('x', (N, 256, hw, hw), seed=5)
xd = x.to(DEV).requires_grad_(); y = m(xd)
xo = x.clone().requires_grad_(); yo = o(xo)
gy = synthetic.tensor('gy', tuple(yo.shape), seed=5)
y.backward(gy.to(DEV)); yo.backward(gy)
prog = list(m._runner().programs.values())[0]
names = ['input', 'b1.c1', 'b1.c2', 'b1.c3', 'b2.c1', 'b2.c2', 'b2.c3']
for a, n in zip(prog.tape.acts, names):
    if n == 'input':
        ref_v, ref_g = xo.detach(), xo.grad
    else:
        ref_v, ref_g = saved[n].detach(), saved[n].grad
    v = a.buf.cpu().permute(0, 3, 1, 2); g = a.grad.cpu().permute(0, 3, 1, 2)
    d = (g - ref_g).abs()
    print(n, tuple(a.buf.shape), 'val %.1e grad %.1e' % (rel(v, ref_v), rel(g, ref_g)), 'argmax', [int(i) for i in (d == d.max()).nonzero()[0]],
          'per-image err', [float(d[i].max()) for i in range(N)], 'row err(h) top5', sorted([(float(d[:, :, h].max()), h) for h in range(hw)])[-5:])
for (n, p), (_, q) in zip(m.named_parameters(), o.named_parameters()):
    print('   %-22s %.1e' % (n, rel(p.grad.cpu(), q.grad)))

# ---- isolate: conv2 dgrad of b1 on the oracle's data
import ctypes as C
from dsnt import _lib
from dsnt._lib import ptr, call, ConvGeom
o.zero_grad()
inp = {}
def pre(mod, args):
    args[0].retain_grad(); inp['a'] = args[0]
h = o.b1.conv2.register_forward_pre_hook(pre)
xo2 = x.clone().requires_grad_(); yo2 = o(xo2); yo2.backward(gy)
da_ref = inp['a'].grad                      # dL/d(conv2 input)
g2 = saved['b1.c2'].grad                    # dL/d(conv2 output)
w = o.b1.conv2.weight.detach()
gyd = g2.permute(0, 2, 3, 1).contiguous().to(DEV)
wd_ohwi = w.permute(0, 2, 3, 1).contiguous().to(DEV)
wdg = torch.empty(128, 3, 3, 128, device=DEV)
call('dsnt_conv_pack_dgrad', ptr(wd_ohwi), ptr(wdg), 128, 3, 3, 128)
gd = ConvGeom(N, hw, hw, 128, hw, hw, 128, 3, 3, 1, 1, 1)
da = torch.empty(N, hw, hw, 128, device=DEV)
call('dsnt_conv_fwd', ptr(gyd), ptr(wdg), None, ptr(da), None, None, 0, None, None, None, C.byref(gd))
d = (da.cpu().permute(0, 3, 1, 2) - da_ref).abs()
print('isolated dgrad: rel', d.max().item() / da_ref.abs().max().item(), 'argmax', [int(i) for i in (d == d.max()).nonzero()[0]])
print('bm', _lib.fn('dsnt_conv_fwd_bm')(C.byref(gd)))
