"""Where the HOST's time of a train step goes (cProfile over the bare loop; the device runs behind): python3 tools/host_profile.py [steps] [batch] [base]"""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt.model import build_mpii_pose_model
from dsnt import synthetic, optim
dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
base = sys.argv[3] if len(sys.argv) > 3 else 'hg2'
model = build_mpii_pose_model(base=base, output_strat='dsnt', reg='js')
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
for _ in range(3):
    step()
torch.cuda.synchronize()
# phase timing (host only: no synchronisation inside)
ph = {'forward': 0.0, 'loss': 0.0, 'zero_grad': 0.0, 'backward': 0.0, 'optim': 0.0}
t_all = time.perf_counter()
for _ in range(steps):
    t0 = time.perf_counter(); out = model(x)
    t1 = time.perf_counter(); loss = model.forward_loss(out, target, mask)
    t2 = time.perf_counter(); opt.zero_grad()
    t3 = time.perf_counter(); loss.backward()
    t4 = time.perf_counter(); opt.step()
    t5 = time.perf_counter()
    for k, v in zip(ph, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        ph[k] += v
host = time.perf_counter() - t_all
torch.cuda.synchronize()
wall = time.perf_counter() - t_all
print('host issue %.3f ms/step, wall %.3f ms/step' % (1e3 * host / steps, 1e3 * wall / steps))
print('  ' + ', '.join('%s %.3f' % (k, 1e3 * v / steps) for k, v in ph.items()))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
