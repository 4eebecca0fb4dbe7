"""hg8 smooth network: save (first call) or compare (second call) every parameter gradient and BatchNorm running statistic
of one step — two processes under different switches (e.g. DSNT_OFF=conv3s / unset): python tools/diag_hg8_ab.py /tmp/a.pt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
os.environ['DSNT_DEBUG_NO_RELU'] = '1'
from dsnt.model import build_mpii_pose_model
from dsnt import synthetic
dev = torch.device('cuda:0')
m = build_mpii_pose_model(base=os.environ.get('BASE', 'hg8'), output_strat='dsnt', reg='js')
synthetic.fill_state_dict(m, seed=3)
m.cuda().train()
x, target, mask = synthetic.batch(2, size=128, seed=2, mask_p=0.8)
outs = m(x.to(dev))
loss = m.forward_loss(outs, target.to(dev), mask.to(dev))
loss.backward()
torch.cuda.synchronize()
cur = {'grad.' + n: p.grad.detach().cpu() for n, p in m.named_parameters()}
cur.update({'buf.' + n: b.detach().cpu() for n, b in m.named_buffers() if 'running' in n})
path = sys.argv[1]
if not os.path.exists(path):
    torch.save(cur, path)
    print('saved', len(cur), 'loss', loss.item())
else:
    ref = torch.load(path)
    rows = []
    for i, (n, v) in enumerate(cur.items()):
        d = (v.double() - ref[n].double()).norm().item()
        s = ref[n].double().norm().item()
        rows.append((d / max(s, 1e-30), d, s, i, n))
    print('loss', loss.item(), ' differing tensors: %d of %d' % (sum(1 for r in rows if r[1] > 0), len(rows)))
    print('first differing running statistics (forward order):')
    k = 0
    for r in rows:
        if r[4].startswith('buf.') and r[1] > 0 and k < 8:
            print('   %.3e  |d| %.3e  |ref| %.3e  %s' % (r[0], r[1], r[2], r[4])); k += 1
    print('largest running-statistic differences:')
    for r in sorted((r for r in rows if r[4].startswith('buf.')), reverse=True)[:6]:
        print('   %.3e  |d| %.3e  |ref| %.3e  #%d %s' % r)
    floor = 1e-3 * max(r[2] for r in rows if r[4].startswith('grad.'))
    print('largest gradient differences relative to max(|ref|, floor %.3e), then in forward order those above 3e-4:' % floor)
    g = [(r[1] / max(r[2], floor), r[1], r[2], r[3], r[4]) for r in rows if r[4].startswith('grad.')]
    for r in sorted(g, reverse=True)[:8]:
        print('   %.3e  |d| %.3e  |ref| %.3e  %s' % (r[0], r[1], r[2], r[4]))
    print('   --')
    for r in [r for r in g if r[0] > 3e-4][-25:]:
        print('   %.3e  |d| %.3e  |ref| %.3e  #%d %s' % r)
