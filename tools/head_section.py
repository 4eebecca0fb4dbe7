"""Untraced time of the section between the two launch lists of a train step — heads, loss, and the autograd glue around them
(HIP events on the caller's stream): end of the forward list -> first launch of the backward list.
python3 tools/head_section.py [steps] [batch] [base]"""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt.model import build_mpii_pose_model
from dsnt import synthetic, optim
dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
base = sys.argv[3] if len(sys.argv) > 3 else 'hg2'
model = build_mpii_pose_model(base=base, output_strat='dsnt', reg='js')
synthetic.fill_state_dict(model, seed=0)
model.cuda().train()
x, target, mask = synthetic.batch(batch, size=256, seed=1, mask_p=1.0)
x, target, mask = x.to(dev), target.to(dev), mask.to(dev)
model.hg._runner().ensure(dev)
opt = optim.RMSprop(model, lr=2.5e-4)
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(steps + 2)]
cur = [None]
runner = model.hg._runner()
runner.probe = None
for s in range(steps + 2):
    e = ev[s]
    cur[0] = e
    e[0].record()
    outs = model.forward_part1(x)
    e[1].record()                                   # the forward list is enqueued (and its output copies)
    out = model.forward_part2(outs)
    loss = model.forward_loss(out, target, mask)
    e[2].record()
    opt.zero_grad()
    first = getattr(outs, 'stacked', None)
    if first is None:
        first = outs[0] if isinstance(outs, (list, tuple)) else outs
    first.register_hook(lambda g, e=e: (e[3].record(), g)[1])     # gradient of the first stack's heat-maps: late in the head backward
    loss.backward()
    e[4].record()
    opt.step()
    e[5].record()
torch.cuda.synchronize()
names = ['forward list', 'heads + loss (fwd)', 'head backward (to the first stack\'s gradient)', 'backward list', 'optimiser']
for k in range(5):
    v = [ev[s][k].elapsed_time(ev[s][k + 1]) for s in range(2, steps + 2)]
    print('%-48s %7.3f ms (median of %d; min %.3f max %.3f)' % (names[k], statistics.median(v), steps, min(v), max(v)))
v = [ev[s][0].elapsed_time(ev[s + 1][0]) for s in range(2, steps + 1)]
print('%-48s %7.3f ms (start to start)' % ('step', statistics.median(v)))
