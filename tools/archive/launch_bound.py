"""Is the train step limited by the CPU issuing launches?  Times the enqueue (CPU returns from step())
against the synchronised step, with and without lanes."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt.model import build_mpii_pose_model
from dsnt import synthetic, optim
dev = torch.device('cuda:0')
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = build_mpii_pose_model(base='hg2', output_strat='dsnt', reg='js')
synthetic.fill_state_dict(model, seed=0)
model.cuda().train()
x, target, mask = synthetic.batch(batch, size=256, seed=1, mask_p=1.0)
x, target, mask = x.to(dev), target.to(dev), mask.to(dev)
model.hg._runner().ensure(dev)
opt = optim.RMSprop(model, lr=2.5e-4)
def step():
    out = model(x)
    loss = model.forward_loss(out, target, mask)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss
for _ in range(3):
    step()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
print('batch %d: enqueue %.2f ms, step (isolated, synchronised) %.2f ms' % (batch, 1e3 * sorted(enq)[len(enq) // 2], 1e3 * sorted(tot)[len(tot) // 2]))
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print('back-to-back: %.2f ms/step' % (1e2 * (time.perf_counter() - t0)))
