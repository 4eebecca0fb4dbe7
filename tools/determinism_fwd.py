"""Train-mode forward repeated: every activation buffer must be bit-identical from run to run.  Prints the first
activations (creation order) that differ:  python tools/determinism_fwd.py [base] [batch] [repeats] [backward 0/1]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt.model import build_mpii_pose_model
from dsnt import synthetic
base = sys.argv[1] if len(sys.argv) > 1 else 'hg2'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
with_bwd = len(sys.argv) > 4 and sys.argv[4] == '1'
m = build_mpii_pose_model(base=base, output_strat='dsnt', reg='js')
synthetic.fill_state_dict(m, seed=0)
m.cuda().train()
x, t, k = synthetic.batch(batch, size=256, seed=1, mask_p=0.9)
x, t, k = x.cuda(), t.cuda(), k.cuda()
def run():
    out = m(x)
    if with_bwd:
        for p in m.parameters():
            p.grad = None
        m.forward_loss(out, t, k).backward()
    torch.cuda.synchronize()
run()
prog = [p for p in m.hg._runner().programs.values() if p.training][0]
acts = prog.tape.acts
def snap():
    run()
    s = [a.buf.clone() for a in acts]
    global stats
    stats = [a.stats[0].clone() if a.stats is not None else None for a in acts]
    g = [a.grad.clone() if (with_bwd and a.grad is not None) else None for a in acts]
    return s, g
ref, gref = snap()
sref = stats
for r in range(reps):
    cur, gcur = snap()
    bad = [(i, acts[i].name, tuple(acts[i].buf.shape), float((ref[i] - cur[i]).abs().max())) for i in range(len(acts)) if not torch.equal(ref[i], cur[i])]
    gbad = [(i, acts[i].name, float((gref[i] - gcur[i]).abs().max())) for i in range(len(acts)) if gref[i] is not None and gcur[i] is not None and not torch.equal(gref[i], gcur[i])]
    sbad = [(i, acts[i].name, tuple(sref[i].shape), float((sref[i] - stats[i]).abs().max())) for i in range(len(acts)) if sref[i] is not None and not torch.equal(sref[i], stats[i])]
    print('   statistics partials that differ:', sbad[:4])
    if sbad:
        i = sbad[0][0]
        pos = (sref[i] != stats[i]).nonzero()
        print('   act %d: %d entries differ: tiles %s kinds %s channels %s' % (i, len(pos), sorted(set(pos[:, 0].tolist()))[:20],
              sorted(set(pos[:, 1].tolist())), sorted(set(pos[:, 2].tolist()))[:40]))
        print('   ref', sref[i][pos[0, 0], :, pos[0, 2]].tolist(), 'now', stats[i][pos[0, 0], :, pos[0, 2]].tolist())
    print('repeat %d: %d of %d activations differ; first: %s | %d gradients differ; last-created: %s' % (r, len(bad), len(acts), bad[:3], len(gbad), gbad[-3:]))

# which entries of the statistics partials are wrong (against a recomputation from the stored activation)?
for i, a in enumerate(acts):
    if a.stats is None or a.M % 128:
        continue
    part = a.stats[0]
    tiles, C_ = part.shape[0], part.shape[2]
    rows = a.M // tiles
    yv = a.buf.view(tiles, rows, C_).double()
    want = torch.stack([yv.sum(1), (yv ** 2).sum(1)], 1)
    err = (part.double() - want).abs() / (want.abs() + 1.0)
    if float(err.max()) > 1e-4:
        badpos = (err > 1e-4).nonzero()
        print('act %d %s partial %s: %d wrong entries; tiles %s kinds %s channels %s..%s' % (
            i, a.name, tuple(part.shape), len(badpos), sorted(set(badpos[:, 0].tolist()))[:12], sorted(set(badpos[:, 1].tolist())),
            int(badpos[:, 2].min()), int(badpos[:, 2].max())))
