"""The round-3 kernels against the kernels they replace, end to end: one hg2 + DSNT + JS train step (batch 16, 256 px: every
one of conv3s / gemm1 / wgrad3 / wgrad1 is eligible at the 64 x 64 level) in two processes — default, and with
DSNT_OFF=conv3s+gemm1+wgrad3+wgrad1 (the library reads the switch once per process) — must agree to fp32 rounding of another
summation order: loss, coordinates, every parameter gradient, running statistics."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys, torch
sys.path[:0] = [os.path.join(%(root)r, 'dsnt-pose2d_amd')]
from dsnt.model import build_mpii_pose_model
from dsnt import synthetic
m = build_mpii_pose_model(base=%(base)r, output_strat='dsnt', reg='js')
synthetic.fill_state_dict(m, seed=0)
m.cuda().train()
x, t, k = synthetic.batch(%(batch)d, size=%(size)d, seed=1, mask_p=0.9)
out = m(x.cuda())
loss = m.forward_loss(out, t.cuda(), k.cuda())
loss.backward()
prog = [p for p in (m.hg if hasattr(m, 'hg') else m)._runner().programs.values() if p.training][0]
names = [e[2] for e in prog.tape.fwd + prog.tape.bwd if e[0] is not None]
torch.save({'loss': loss.item(), 'coords': (out[-1] if isinstance(out, (list, tuple)) else out).detach().cpu(),
            'grads': {n: p.grad.detach().cpu() for n, p in m.named_parameters()},
            'running': {n: b.detach().cpu() for n, b in m.named_buffers() if 'running' in n},
            'stream_launches': names.count('dsnt_conv_fwd_f16x3_stream'),
            'strided_launches': names.count('dsnt_conv_dgrad_strided'), 'stuffed_launches': names.count('dsnt_zero_insert')},
           sys.argv[1])
'''


def _run(tmp_path, tag, off, base='hg2', batch=16, size=256):
    env = dict(os.environ)
    env.pop('DSNT_OFF', None)
    if off:
        env['DSNT_OFF'] = off
    path = str(tmp_path / (tag + '.pt'))
    subprocess.run([sys.executable, '-c', SCRIPT % {'root': ROOT, 'base': base, 'batch': batch, 'size': size}, path],
                   check=True, env=env, timeout=600)
    return torch.load(path)


def test_round3_kernels_agree_with_the_kernels_they_replace(tmp_path):
    new = _run(tmp_path, 'new', None)
    old = _run(tmp_path, 'old', 'conv3s+gemm1+wgrad3+wgrad1')
    assert new['stream_launches'] > 0 and old['stream_launches'] == 0          # the switch reached the engine too
    assert abs(new['loss'] - old['loss']) <= 2e-6 * abs(old['loss'])
    assert (new['coords'] - old['coords']).abs().max().item() <= 5e-6
    floor = 1e-3 * max(v.double().norm().item() for v in old['grads'].values())
    worst = max(((new['grads'][n].double() - v.double()).norm().item() / max(v.double().norm().item(), floor), n)
                for n, v in old['grads'].items())
    # (ReLU on: a mask bit may flip between two summation orders — the flip-tolerant bar of tests/test_model_gpu.py)
    assert worst[0] <= 3e-2, worst
    fn = torch.cat([v.reshape(-1) for v in new['grads'].values()]).double()
    fo = torch.cat([v.reshape(-1) for v in old['grads'].values()]).double()
    assert (fn @ fo / (fn.norm() * fo.norm())).item() >= 0.9999
    for n, v in old['running'].items():
        assert (new['running'][n] - v).abs().max().item() <= 1e-5 * max(1.0, v.abs().max().item()), n


def test_native_strided_data_gradient_agrees_with_zero_stuffing(tmp_path):
    """resnet18 + DSNT + JS, one train step: the stage transitions' data gradients on dsnt_conv_dgrad_strided (default) and, with
    DSNT_OFF=dgrad_up, on dsnt_zero_insert + the stride-1 kernels.  Only backward launches differ (no ReLU mask can flip), and
    the new kernel is exact fp32: tight bars."""
    new = _run(tmp_path, 'rn_new', None, 'resnet18', 8, 128)
    old = _run(tmp_path, 'rn_old', 'dgrad_up', 'resnet18', 8, 128)
    assert new['strided_launches'] == 6 and new['stuffed_launches'] == 0
    assert old['strided_launches'] == 0 and old['stuffed_launches'] == 6
    assert new['loss'] == old['loss'] and torch.equal(new['coords'], old['coords'])          # the forward is the same program
    floor = 1e-3 * max(v.double().norm().item() for v in old['grads'].values())
    worst = max(((new['grads'][n].double() - v.double()).norm().item() / max(v.double().norm().item(), floor), n)
                for n, v in old['grads'].items())
    assert worst[0] <= 1e-4, worst
