"""Bare train loop (no bench extras) for rocprofv3: python3 tools/train_loop.py [steps] [batch] [base: hg2 | hg8 | hg1 | resnet34 ...] [reg: js | none | ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt.model import build_mpii_pose_model
from dsnt import synthetic, optim
dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
base = sys.argv[3] if len(sys.argv) > 3 else 'hg2'
reg = sys.argv[4] if len(sys.argv) > 4 else 'js'
model = build_mpii_pose_model(base=base, output_strat='dsnt', reg=reg)
synthetic.fill_state_dict(model, seed=0)
model.cuda().train()
x, target, mask = synthetic.batch(batch, size=256, seed=1, mask_p=1.0)
x, target, mask = x.to(dev), target.to(dev), mask.to(dev)
(model.hg if hasattr(model, 'hg') else model)._runner().ensure(dev)
opt = optim.RMSprop(model, lr=2.5e-4)
def step():
    out = model(x)
    loss = model.forward_loss(out, target, mask)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print('%.3f ms/step over %d steps (+2 warm-up), batch %d, %s (host issue %.3f ms/step)'
      % (1e3 * (time.perf_counter() - t0) / steps, steps, batch, base, 1e3 * (t1 - t0) / steps))
