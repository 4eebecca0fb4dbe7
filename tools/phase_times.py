"""Untraced phase times of a train step (HIP events on the caller's stream; the launch lists join their lanes before they return):
forward list | head + loss | backward (head + list) | optimiser.  python3 tools/phase_times.py [steps] [batch] [base]"""
import os, sys
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
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(steps + 2)]
for s in range(steps + 2):
    e = ev[s]
    e[0].record()
    out = model(x)
    e[1].record()
    loss = model.forward_loss(out, target, mask)
    e[2].record()
    opt.zero_grad()
    loss.backward()
    e[3].record()
    opt.step()
    e[4].record()
torch.cuda.synchronize()
import statistics
names = ['forward list', 'head + loss', 'backward', 'optimiser', 'step']
for k in range(4):
    v = [ev[s][k].elapsed_time(ev[s][k + 1]) for s in range(2, steps + 2)]
    print('%-14s %7.3f ms (median of %d; min %.3f max %.3f)' % (names[k], statistics.median(v), steps, min(v), max(v)))
v = [ev[s][0].elapsed_time(ev[s + 1][0]) for s in range(2, steps + 1)]
print('%-14s %7.3f ms (start to start)' % ('step', statistics.median(v)))
