"""Inference throughput of the HIP path (f-2, reference inference.py:33-48): eval-mode forward of hg2 + DSNT,
batch B, with and without horizontal-flip test-time augmentation.   python tools/bench_infer.py [batch] [base]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt.model import build_mpii_pose_model
from dsnt import synthetic
from dsnt.inference import HFLIP_INDICES
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
base = sys.argv[2] if len(sys.argv) > 2 else 'hg2'
m = build_mpii_pose_model(base=base, output_strat='dsnt')
synthetic.fill_state_dict(m, seed=0)
m.cuda().eval()
x, _, _ = synthetic.batch(B, size=256, seed=1)
x = x.to(dev)
idx = HFLIP_INDICES.to(dev)
def plain():
    with torch.no_grad():
        return m.compute_coords if False else m(x)[-1]
def flipped():
    with torch.no_grad():
        xin = torch.cat([x, x.flip(-1)], 0)
        hm = m.forward_part1(xin)[-1]
        hm1, hm2 = hm[:B], hm[B:].flip(-1).index_select(-3, idx)
        return m.forward_part2([(hm1 + hm2) / 2])[-1]
for name, fn in (('plain', plain), ('flip-TTA', flipped)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print('%s %s batch %d: %.2f ms -> %.0f images/s' % (base, name, B, dt * 1e3, B / dt))
prog = [p for p in m.hg._runner().programs.values()][0]
print('launches per forward:', prog.n_fwd)
