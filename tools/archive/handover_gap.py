"""GPU time between the last kernel of the forward list and the first of the backward list (the DSNT head, the loss and
autograd's start-up sit in between): is the hand-over host-bound when nothing traces the process?
python tools/handover_gap.py [steps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt.model import build_mpii_pose_model
from dsnt import synthetic, optim
dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
model = build_mpii_pose_model(base='hg2', output_strat='dsnt', reg='js')
synthetic.fill_state_dict(model, seed=0)
model.cuda().train()
x, target, mask = synthetic.batch(32, size=256, seed=1, mask_p=1.0)
x, target, mask = x.to(dev), target.to(dev), mask.to(dev)
model.hg._runner().ensure(dev)
opt = optim.RMSprop(model, lr=2.5e-4)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
       torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
cur = [None]
model(x)                                         # trace
prog = [p for p in model.hg._runner().programs.values() if p.training][0]
_run = prog.tape.run
def run(lst, *a, **k):
    if lst is prog.tape.bwd and cur[0] is not None:
        cur[0].record()
    return _run(lst, *a, **k)
prog.tape.run = run
def step(i):
    e0, e1, e2, e3 = ev[i]
    e0.record()
    out = model(x)
    e1.record()                                  # after the forward list
    cur[0] = e2                                  # recorded by the patched Tape.run when the backward list starts
    loss = model.forward_loss(out, target, mask)
    opt.zero_grad()
    loss.backward()
    opt.step()
    e3.record()
for i in range(3):
    step(i)
torch.cuda.synchronize()
for i in range(steps):
    step(i)
torch.cuda.synchronize()
f = sum(e[0].elapsed_time(e[1]) for e in ev) / steps
h = sum(e[1].elapsed_time(e[2]) for e in ev) / steps
b = sum(e[2].elapsed_time(e[3]) for e in ev) / steps
print('forward list %.3f ms | hand-over (head fwd + loss + head bwd, ~0.06 ms of kernels) %.3f ms | backward list + optimiser %.3f ms' % (f, h, b))
