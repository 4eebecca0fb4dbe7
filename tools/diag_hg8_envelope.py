"""Distribution behind tests/test_model_gpu.py::test_hg8_every_gradient_vs_oracle_on_the_smooth_network: per parameter, the
relative L2 distance of the HIP gradient and of the fp32 CPU oracle's from the fp64 oracle (smooth hg8, batch 2, 128 px).
Usage: DSNT_MFMA=f32|bf16x6 [DSNT_SPLIT=...] python tools/diag_hg8_envelope.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests'), ROOT]
import torch
from dsnt import synthetic
import test_model_gpu as T
from dsnt.model import build_mpii_pose_model
from dsnt_oracle import model as omodel
with T._NoRelu():
    m = build_mpii_pose_model(base='hg8', output_strat='dsnt', reg='js')
    o = omodel.build_mpii_pose_model(base='hg8', output_strat='dsnt', reg='js')
    o64 = omodel.build_mpii_pose_model(base='hg8', output_strat='dsnt', reg='js')
    T._NoRelu.strip(o); T._NoRelu.strip(o64)
    for mm in (m, o, o64):
        synthetic.fill_state_dict(mm, seed=3)
    o64.double(); m.cuda().train(); o.train(); o64.train()
    x, target, mask = synthetic.batch(2, size=128, seed=2, mask_p=0.8)
    m.forward_loss(m(x.cuda()), target.cuda(), mask.cuda()).backward()
    o.forward_loss(o(x), target, mask).backward()
    o64.forward_loss(o64(x.double()), target.double(), mask.double()).backward()
pm, po, p64 = dict(m.named_parameters()), dict(o.named_parameters()), dict(o64.named_parameters())
floor = 1e-3 * max(q.grad.norm().item() for q in p64.values())
rows = []
for n in pm:
    eh = T._rel_l2(pm[n].grad.cpu(), p64[n].grad, floor)
    eo = T._rel_l2(po[n].grad, p64[n].grad, floor)
    rows.append((eh, eo, n))
eh = torch.tensor([r[0] for r in rows]); eo = torch.tensor([r[1] for r in rows])
q = torch.tensor([0.5, 0.9, 0.99, 1.0])
print('path', os.environ.get('DSNT_MFMA'), os.environ.get('DSNT_SPLIT'), 'params', len(rows))
print('HIP    vs fp64: quantiles 50/90/99/100 %', [float('%.3g' % v) for v in torch.quantile(eh, q)])
print('fp32 o vs fp64: quantiles 50/90/99/100 %', [float('%.3g' % v) for v in torch.quantile(eo, q)])
ratio = eh / eo.clamp_min(1e-9)
print('ratio HIP / fp32 oracle: quantiles', [float('%.3g' % v) for v in torch.quantile(ratio, q)])
for r in sorted(rows, reverse=True)[:8]:
    print('  worst HIP %.3g (fp32 oracle %.3g) %s' % r)

# near-ties of the max-pools in the fp64 forward: (max - second max) of every 2x2 window, relative to the tensor's largest |value|
import torch.nn.functional as F
from dsnt_oracle import hourglass as ohg
gaps = []
class Shim:
    relu = staticmethod(lambda t: t)
    interpolate = staticmethod(F.interpolate)
    @staticmethod
    def max_pool2d(xx, *a, **k):
        N, C, H, W = xx.shape
        w = xx.view(N, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, C, H // 2, W // 2, 4)
        top2 = w.topk(2, dim=-1).values
        gap = (top2[..., 0] - top2[..., 1]) / xx.abs().max()
        gaps.append((tuple(xx.shape), gap.min().item(), int((gap < 1e-6).sum())))
        return F.max_pool2d(xx, *a, **k)
saved, ohg.F = ohg.F, Shim
with torch.no_grad():
    o64(x.double())
ohg.F = saved
print('max-pool near-ties in the fp64 forward (pool 0 = stem, then four per stack): index, input shape, smallest relative gap, windows below 1e-6')
for i, g in enumerate(gaps):
    if g[1] < 1e-6:
        print('  pool %2d (stack %d) %s  min gap %.2e  windows < 1e-6: %d' % (i, (i - 1) // 4, g[0], g[1], g[2]))
