"""Pure host cost of replaying the recorded launch lists (queues empty before every replay): python3 tools/host_replay.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "dsnt-pose2d_amd")]
from dsnt.model import build_mpii_pose_model
from dsnt import synthetic, optim
dev = torch.device('cuda:0')
model = build_mpii_pose_model(base='hg2', output_strat='dsnt', reg='js')
synthetic.fill_state_dict(model, seed=0); model.cuda().train()
x, target, mask = synthetic.batch(32, size=256, seed=1, mask_p=1.0)
x, target, mask = x.to(dev), target.to(dev), mask.to(dev)
r = model.hg._runner(); r.ensure(dev)
opt = optim.RMSprop(model, lr=2.5e-4)
for _ in range(3):
    out = model(x); loss = model.forward_loss(out, target, mask); opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize()
prog = [p for p in r.programs.values() if p.training][0]
tape = prog.tape
for nm, lst in (('fwd', tape.fwd), ('bwd', tape.bwd)):
    nl = sum(1 for e in lst if e[0] is not None); ns = sum(1 for e in lst if e[0] is None and e[2] == 'sync')
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); tape.run(lst); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    print(nm, 'launches', nl, 'syncs', ns, 'host ms / wall ms:', ' '.join('%.2f/%.2f' % t for t in ts))
