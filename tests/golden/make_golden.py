"""Generate golden vectors by running THE REFERENCE (imported from /root/reference).

Run in the build container only:  python tests/golden/make_golden.py
Outputs small .npz files next to this script.  A fixture is data: expected
outputs of the reference on deterministic inputs that `dsnt.synthetic`
regenerates anywhere (numpy PCG64), so no weights or images are stored.

Families:
  head_*        DSNT head + every regulariser on [4,16,64,64] logits (fp32 and fp64)
  bottleneck    Bottleneck(256,128) train-mode fwd/bwd at [2,256,16,16]
  hourglass     Hourglass(depth 4) fwd/bwd at [2,256,32,32]
  hg1_128 / hg2_128 / hg2_256 / hg8_128   end-to-end model: coords of EVERY stack, loss, per-parameter grad norms
                (hg8 = experiments/hourglass.json: eight stacks, seven inter-stack remaps, hourglass.py:166-175)
  pckh          PCKh on synthetic predictions (reference evaluator restated: torchnet absent)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd'), os.path.join(ROOT, 'tests'),
                os.path.join(ROOT, 'oracle')]

from dsnt import synthetic  # noqa: E402  (product-side deterministic data; no HIP needed)
import refimport  # noqa: E402

SAMPLES = 512


def sample_idx(n, name):
    r = np.random.Generator(np.random.PCG64([7, len(name), n % 65521]))
    return np.sort(r.choice(n, size=min(SAMPLES, n), replace=False))


def summarize(prefix, t, out):
    a = t.detach().double().reshape(-1).numpy()
    idx = sample_idx(a.size, prefix)
    out[prefix + '.idx'] = idx
    out[prefix + '.val'] = a[idx]
    out[prefix + '.sum'] = np.float64(a.sum())
    out[prefix + '.l2'] = np.float64(np.sqrt((a * a).sum()))


def head(ref_nn, ref_model, dtype, tag):
    out = {}
    logits = (synthetic.tensor('head.logits', (4, 16, 64, 64), seed=11) * 3).to(dtype)
    logits.requires_grad_()
    target = synthetic.tensor('head.target', (4, 16, 2), seed=11, kind='uniform').to(dtype)
    mask = (synthetic.tensor('head.mask', (4, 16), seed=11, kind='uniform') > -0.6).to(dtype)
    base = ref_model.HumanPoseModel()
    hm = base._hm_preact(logits, 'softmax')
    coords = ref_nn.dsnt(hm)
    out['coords'] = coords.detach().numpy()
    summarize('heatmaps', hm, out)
    eu = ref_nn.euclidean_loss(coords, target, mask)
    out['euclid'] = np.float64(eu.item())
    sigma = 2.0 * 1.0 / 64
    for reg in ('js', 'kl', 'mse', 'var'):
        fn = {'js': ref_nn.js_reg_loss, 'kl': ref_nn.kl_reg_loss, 'mse': ref_nn.mse_reg_loss,
              'var': ref_nn.variance_reg_loss}[reg]
        r = fn(hm, target, sigma, mask)
        out['reg_' + reg] = np.float64(r.item())
        coeff = 100.0 if reg == 'var' else 1.0
        g, = torch.autograd.grad(eu + coeff * r, logits, retain_graph=True)
        summarize('dlogits_' + reg, g, out)
        g, = torch.autograd.grad(r, hm, retain_graph=True)
        summarize('dhm_' + reg, g, out)
    g, = torch.autograd.grad(eu, logits, retain_graph=True)
    summarize('dlogits_none', g, out)
    out['euclid_nomask'] = np.float64(ref_nn.euclidean_loss(coords, target, None).item())
    out['js_nomask'] = np.float64(ref_nn.js_reg_loss(hm, target, sigma, None).item())
    for preact in ('thresholded_softmax', 'abs', 'relu', 'sigmoid'):
        summarize('preact_' + preact, base._hm_preact(logits, preact), out)
    np.savez_compressed(os.path.join(HERE, 'head_%s.npz' % tag), **out)


def block(ref_hg, kind):
    out = {}
    torch.manual_seed(0)
    if kind == 'bottleneck':
        m = ref_hg.Bottleneck(256, 128)
    else:
        m = ref_hg.Hourglass(ref_hg.Bottleneck, 1, 128, 4)
    synthetic.fill_state_dict(m, seed=5)
    m.train()
    hw = 16 if kind == 'bottleneck' else 32   # innermost hourglass level: 2x2 (BN over 8 values)
    x = synthetic.tensor(kind + '.x', (2, 256, hw, hw), seed=5).requires_grad_()
    gy = synthetic.tensor(kind + '.gy', (2, 256, hw, hw), seed=5)
    y = m(x)
    y.backward(gy)
    summarize('y', y, out)
    summarize('dx', x.grad, out)
    for n, p in m.named_parameters():
        out['gradnorm.' + n] = np.float64(p.grad.double().norm().item())
        out['gradsum.' + n] = np.float64(p.grad.double().sum().item())
    for n, b in m.named_buffers():
        if 'running' in n:
            out['buf.' + n] = b.double().numpy()
    np.savez_compressed(os.path.join(HERE, kind + '.npz'), **out)


def end_to_end(ref_model, base, size, reg, tag, batch=2):
    out = {}
    m = ref_model.build_mpii_pose_model(base=base, output_strat='dsnt', reg=reg)
    synthetic.fill_state_dict(m, seed=0)
    m.train()
    x, target, mask = synthetic.batch(batch, size=size, seed=1, mask_p=0.9)
    outs = m(x)
    loss = m.forward_loss(outs, target, mask)
    loss.backward()
    out['loss'] = np.float64(loss.item())
    for i, o in enumerate(outs):
        out['coords%d' % i] = o.detach().numpy()
        summarize('heatmaps%d' % i, m.heatmaps_array[i], out)
    for n, p in m.named_parameters():
        out['gradnorm.' + n] = np.float64(p.grad.double().norm().item())
    for n, b in m.named_buffers():
        if 'running' in n:
            out['bufsum.' + n] = np.float64(b.double().sum().item())
    # eval-mode coords (weights unchanged: an optimiser step would amplify ReLU-kink flips
    # between fp32 implementations).  One more training forward with momentum 1 makes the running
    # statistics equal this batch's statistics, so eval mode is in a well-conditioned regime
    # (after a single default-momentum update an untrained net saturates its heat-maps).
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 1.0
    with torch.no_grad():
        m(x)
    m.eval()
    with torch.no_grad():
        out['eval_coords'] = m(x)[-1].numpy()
    if base == 'hg8':
        # Eight stacks at batch 2 sit at the conditioning limit of fp32: the fp32 reference itself is 1.2e-4 (eval-mode
        # coordinates) / 9e-5 (running statistics) away from its own fp64 run.  The fp64 run of the reference is stored
        # too, so that a test can hold an fp32 implementation to "no further from the truth than twice the fp32
        # reference is" where a plain 1e-4 bar against the fp32 numbers would be a coin toss.
        m64 = ref_model.build_mpii_pose_model(base=base, output_strat='dsnt', reg=reg)
        synthetic.fill_state_dict(m64, seed=0)
        m64.double().train()
        x64 = x.double()
        outs64 = m64(x64)
        for i, o in enumerate(outs64):
            out['coords%d_f64' % i] = o.detach().numpy()
        for n, b in m64.named_buffers():
            if 'running' in n:
                out['bufsum_f64.' + n] = np.float64(b.double().sum().item())
        for mod in m64.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.momentum = 1.0
        with torch.no_grad():
            m64(x64)
        m64.eval()
        with torch.no_grad():
            out['eval_coords_f64'] = m64(x64)[-1].numpy()
    np.savez_compressed(os.path.join(HERE, tag + '.npz'), **out)


def pckh():
    from dsnt_oracle.evaluator import PCKhEvaluator  # torchnet absent: restated evaluator,
    out = {}                                          # itself pinned by the reference's tests
    _, target, mask = synthetic.batch(64, size=8, seed=3, mask_p=0.85)
    pred = target + synthetic.tensor('pckh.noise', (64, 16, 2), seed=3, scale=0.15)
    head, m, b = synthetic.pckh_inputs(64)
    op = torch.bmm(pred.double(), m) + b
    ot = torch.bmm(target.double(), m) + b
    ev = PCKhEvaluator(0.5)
    ev.add(op, ot, mask, head)
    for k, meter in ev.meters.items():
        out[k] = np.float64(meter.value()[0])
    np.savez_compressed(os.path.join(HERE, 'pckh.npz'), **out)


def main():
    ref = refimport.load_reference()
    assert ref is not None, 'run in the build container: /root/reference is required'
    ref_nn, ref_hg, ref_model = ref
    torch.set_num_threads(8)
    if len(sys.argv) > 1:                   # one end-to-end family only, e.g. `make_golden.py hg8_128 hg8 128 js`
        tag, base, size, reg = sys.argv[1:5]
        end_to_end(ref_model, base, int(size), reg, tag)
        print(tag, os.path.getsize(os.path.join(HERE, tag + '.npz')))
        return
    head(ref_nn, ref_model, torch.float32, 'f32')
    head(ref_nn, ref_model, torch.float64, 'f64')
    block(ref_hg, 'bottleneck')
    block(ref_hg, 'hourglass')
    end_to_end(ref_model, 'hg1', 128, 'none', 'hg1_128')
    end_to_end(ref_model, 'hg2', 128, 'js', 'hg2_128')
    end_to_end(ref_model, 'hg2', 256, 'js', 'hg2_256')
    end_to_end(ref_model, 'hg8', 128, 'js', 'hg8_128')
    pckh()
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == '__main__':
    main()
