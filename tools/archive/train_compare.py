"""Train hg2 + DSNT + JS on a fixed synthetic batch for N steps (RMSprop lr 2.5e-4) and print the loss trajectory:
a practical check that the split-precision paths (DSNT_SPLIT=f16x3 | bf16x6, DSNT_MFMA=f32) train alike and stay finite
while weights and gradient magnitudes move.   python tools/train_compare.py [steps] [batch]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt.model import build_mpii_pose_model
from dsnt import synthetic, optim
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device('cuda:0')
model = build_mpii_pose_model(base='hg2', output_strat='dsnt', reg='js')
synthetic.fill_state_dict(model, seed=0)
model.cuda().train()
x, t, k = synthetic.batch(batch, size=256, seed=1, mask_p=0.9)
x, t, k = x.to(dev), t.to(dev), k.to(dev)
model.hg._runner().ensure(dev)
opt = optim.RMSprop(model, lr=2.5e-4)
out = []
for i in range(steps):
    loss = model.forward_loss(model(x), t, k)
    opt.zero_grad()
    loss.backward()
    opt.step()
    if i % (steps // 10) == 0 or i == steps - 1:
        out.append('%d:%.5f' % (i, loss.item()))
gn = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).norm().item()
print(os.environ.get('DSNT_SPLIT', 'f16x3') + '/' + os.environ.get('DSNT_MFMA', 'bf16x6'), ' '.join(out), '| final grad norm %.4g' % gn,
      '| finite', bool(torch.isfinite(torch.cat([p.detach().reshape(-1) for p in model.parameters()])).all()))
