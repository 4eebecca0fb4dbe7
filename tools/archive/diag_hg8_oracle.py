"""hg8 smooth network (no ReLU), batch 2, 128 px: per-parameter gradient distance HIP vs oracle fp32 vs oracle fp64
(floor = 1e-3 of the largest gradient norm, as tests/test_model_gpu.py::_grads_close).  Diagnostic for the bar of
test_hg8_every_gradient_vs_oracle_on_the_smooth_network."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import test_model_gpu as T
from dsnt.model import build_mpii_pose_model
from dsnt_oracle import model as omodel
from dsnt import synthetic
dev = torch.device('cuda:0')
if os.environ.get('FORCE_GEMM6'):
    from dsnt import _lib
    _lib.load().dsnt_debug_force_gemm6(1)
def grads(o, x, t, k):
    outs = o(x); loss = o.forward_loss(outs, t, k); loss.backward()
    return {n: p.grad.detach().double().cpu() for n, p in o.named_parameters()}, loss.item()
with T._NoRelu():
    x, target, mask = synthetic.batch(2, size=128, seed=2, mask_p=0.8)
    m = build_mpii_pose_model(base='hg8', output_strat='dsnt', reg='js'); synthetic.fill_state_dict(m, seed=3); m.cuda().train()
    gm, lm = grads(m, x.to(dev), target.to(dev), mask.to(dev))
    res = {}
    for nt in (int(os.environ.get('NT', '0')) or torch.get_num_threads(), 1):
        torch.set_num_threads(nt)
        o = omodel.build_mpii_pose_model(base='hg8', output_strat='dsnt', reg='js'); T._NoRelu.strip(o); synthetic.fill_state_dict(o, seed=3); o.train()
        t0 = time.time(); res['o32_t%d' % nt], l32 = grads(o, x, target, mask); print('oracle fp32, %d threads: %.1f s, loss %.7f' % (nt, time.time() - t0, l32))
    torch.set_num_threads(int(os.environ.get('NT', '0')) or 16)
    o = omodel.build_mpii_pose_model(base='hg8', output_strat='dsnt', reg='js'); T._NoRelu.strip(o); synthetic.fill_state_dict(o, seed=3); o.double().train()
    t0 = time.time(); g64, l64 = grads(o, x.double(), target.double(), mask.double()); print('oracle fp64: %.1f s, loss %.9f; hip loss %.7f' % (time.time() - t0, l64, lm))
floor = 1e-3 * max(v.norm().item() for v in g64.values())
def worst(a, b):
    w = max(((a[n] - b[n]).norm().item() / max(b[n].norm().item(), floor), n) for n in a)
    return '%.3e %s' % w
print('HIP vs fp64      :', worst(gm, g64))
for k, v in res.items():
    print('%s vs fp64  :' % k, worst(v, g64))
    print('HIP vs %s   :' % k, worst(gm, v))
ks = list(res)
print('%s vs %s:' % (ks[0], ks[1]), worst(res[ks[0]], res[ks[1]]))
