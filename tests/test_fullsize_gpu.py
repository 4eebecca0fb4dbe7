"""BASELINE configs 3 and 2 at FULL size (hg2 + DSNT + JS and hg1 + DSNT, batch 32, 256x256 -> 64x64x16) and the 1-GPU leg of
config 4's strong-scaling curve (hg2 + DSNT + JS at GLOBAL batch 256 on one device) on the production path
(bf16x6 from 16384 rows up, grouped weight gradients, K-split kernels, two lanes): size-independent properties
instead of an oracle run (a CPU step at this size takes ~25 s per image batch of 8 on 128 threads).

* heat-maps are distributions and the coordinates are their expectation (dsnt/nn.py:49-78 of the reference);
* the loss equals the CPU oracle's head + loss evaluated on the SAME heat-maps / coordinates;
* the gradient is the derivative of the loss: central difference along the gradient direction;
* a step is deterministic (no atomics anywhere: bit-identical loss and gradients when repeated);
* eval-mode forward of the batch = forwards of its halves (replica semantics of inference).
"""
import pytest
import torch

from dsnt import synthetic

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module', params=[('hg2', 'js', 32), ('hg1', 'none', 32), ('hg2', 'js', 256)],
                ids=['hg2_js', 'hg1', 'hg2_js_b256'])
def setup(request):
    from dsnt.model import build_mpii_pose_model
    base, reg, batch = request.param
    m = build_mpii_pose_model(base=base, output_strat='dsnt', reg=reg)
    synthetic.fill_state_dict(m, seed=0)
    m.to(DEV).train()
    x, t, k = synthetic.batch(batch, size=256, seed=1, mask_p=0.9)
    yield m, x.to(DEV), t.to(DEV), k.to(DEV)
    del m
    torch.cuda.empty_cache()


def _step(m, x, t, k):
    for p in m.parameters():
        p.grad = None
    out = m(x)
    loss = m.forward_loss(out, t, k)
    loss.backward()
    return out, loss


def test_heatmaps_coords_and_loss(setup):
    from dsnt_oracle import nn as onn
    m, x, t, k = setup
    out, loss = _step(m, x, t, k)
    B = x.shape[0]
    assert len(out) == m.hg.num_stacks and out[0].shape == (B, 16, 2)
    total = 0.0
    for hm, coords in zip(m.heatmaps_array, out):
        assert hm.shape == (B, 16, 64, 64) and float(hm.detach().min()) >= 0.0
        assert (hm.double().sum((-1, -2)) - 1).abs().max().item() <= 1e-5
        xs = ((2 * torch.arange(64, device=DEV, dtype=torch.float64) - 63) / 64)
        ex = (hm.double().sum(-2) * xs).sum(-1)
        ey = (hm.double().sum(-1) * xs).sum(-1)
        assert (torch.stack([ex, ey], -1) - coords.double()).abs().max().item() <= 2e-6
        assert coords.abs().max().item() < 1.0
        # the oracle's loss on the same heat-maps and coordinates (CPU, fp32)
        hc, cc = hm.detach().cpu(), coords.detach().cpu()
        total += onn.euclidean_loss(cc, t.cpu(), k.cpu()).item()
        if m.reg == 'js':
            total += onn.js_reg_loss(hc, t.cpu(), 2.0 / 64, k.cpu()).item()
    assert abs(loss.item() - total) <= 1e-5 * abs(total)
    assert torch.equal(m.compute_coords(out), out[-1].detach().cpu())


def test_gradient_is_the_derivative_and_step_is_deterministic(setup):
    m, x, t, k = setup
    _, loss1 = _step(m, x, t, k)
    g1 = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
    _, loss2 = _step(m, x, t, k)
    g2 = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert loss1.item() == loss2.item() and torch.equal(g1, g2)          # no atomics, fixed reduction orders
    assert torch.isfinite(g1).all() and float(g1.norm()) > 0
    # central difference along the (normalised) gradient: dL = |g| * eps.  BatchNorm running statistics move
    # with every forward but do not enter a train-mode loss.
    params = list(m.parameters())
    d = [p.grad.clone() / g1.norm() for p in params]
    eps = 2.5e-4         # small: the loss is strongly curved along its own gradient at random initialisation
    with torch.no_grad():
        for p, dp in zip(params, d):
            p.add_(dp, alpha=eps)
        lp = m.forward_loss(m(x), t, k).item()
        for p, dp in zip(params, d):
            p.add_(dp, alpha=-2 * eps)
        lm = m.forward_loss(m(x), t, k).item()
        for p, dp in zip(params, d):
            p.add_(dp, alpha=eps)
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - float(g1.norm())) <= 0.03 * float(g1.norm()), (fd, float(g1.norm()))


def test_eval_forward_shards_like_replicas(setup):
    m, x, t, k = setup
    m.eval()
    try:
        with torch.no_grad():
            whole = m(x)[-1].clone()
            half = x.shape[0] // 2
            halves = torch.cat([m(x[:half])[-1].clone(), m(x[half:])[-1].clone()])
        # not bit-identical: eval-mode fp16x3 scales its operands by a power of two taken from the batch's own maximum
        # (dsnt_out_bounds.amax_bn), and the 16x16 level of a 16-image shard (4096 rows) falls below the split-precision row
        # threshold (8192) that the whole batch passes; both are fp32-rounding-level effects on coordinates in [-1, 1]
        # (measured: 2e-6 .. 6e-6 at batch 32; 6e-6 .. 1.4e-5 at batch 256, where the shards also differ in which 1x1 kernel
        # the 32x32 level runs on)
        err = (whole - halves).abs().max().item()
        assert err <= (1e-5 if x.shape[0] <= 32 else 3e-5), err
    finally:
        m.train()


@pytest.mark.parametrize('base,batch', [('hg2', 32), ('hg8', 16)], ids=['hg2_b32', 'hg8_b16'])
def test_every_activation_and_gradient_is_bit_reproducible(base, batch):
    """Ten forward/backward passes of the full-size step (hg2 batch 32: config 3; hg8 batch 16: config 5's per-GPU
    shard) from the same state: every activation buffer, every statistics partial and every gradient buffer of the
    launch lists is bit-identical from pass to pass — several lanes run concurrently, so this is the test that catches
    a missing lane dependency or a kernel that is only deterministic when it has the chip to itself (one was found this
    way: tools/determinism_fwd.py)."""
    from dsnt.model import build_mpii_pose_model
    m = build_mpii_pose_model(base=base, output_strat='dsnt', reg='js')
    synthetic.fill_state_dict(m, seed=0)
    m.to(DEV).train()
    x, t, k = synthetic.batch(batch, size=256, seed=1, mask_p=0.9)
    x, t, k = x.to(DEV), t.to(DEV), k.to(DEV)

    def run():
        for p in m.parameters():
            p.grad = None
        loss = m.forward_loss(m(x), t, k)
        loss.backward()
        torch.cuda.synchronize()
        return loss.item()

    run()
    prog = [p for p in m.hg._runner().programs.values() if p.training][0]
    acts = prog.tape.acts

    def snap():
        loss = run()
        return (loss, [a.buf.clone() for a in acts], [a.stats[0].clone() if a.stats is not None else None for a in acts],
                [a.grad.clone() if a.grad is not None else None for a in acts],
                torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone())

    ref = snap()
    for rep in range(9):
        cur = snap()
        assert cur[0] == ref[0], rep
        for kind in (1, 2, 3):
            bad = [(i, acts[i].name) for i in range(len(acts))
                   if ref[kind][i] is not None and not torch.equal(ref[kind][i], cur[kind][i])]
            assert not bad, (rep, ('activation', 'statistics', 'gradient')[kind - 1], bad[:4])
        assert torch.equal(ref[4], cur[4]), rep


def test_resnet34_batch8_properties():
    """BASELINE config 1 at its full size (resnet34 + DSNT + JS, 256 px, batch 8, dilate 0 -> 8 x 8 heat-maps;
    /root/reference/src/dsnt/model.py:79-201) on the production path: the same size-independent properties as above.
    (Its value check against the oracle at this very size is tests/test_resnet_gpu.py CASES.)"""
    from dsnt.model import build_mpii_pose_model
    from dsnt_oracle import nn as onn
    m = build_mpii_pose_model(base='resnet34', output_strat='dsnt', reg='js')
    synthetic.fill_state_dict(m, seed=0)
    m.to(DEV).train()
    x, t, k = synthetic.batch(8, size=256, seed=1, mask_p=0.9)
    x, t, k = x.to(DEV), t.to(DEV), k.to(DEV)

    def step():
        for p in m.parameters():
            p.grad = None
        out = m(x)
        loss = m.forward_loss(out, t, k)
        loss.backward()
        return out, loss

    out, loss1 = step()
    hm = m.heatmaps
    assert out.shape == (8, 16, 2) and hm.shape == (8, 16, 8, 8) and float(hm.detach().min()) >= 0.0
    assert (hm.double().sum((-1, -2)) - 1).abs().max().item() <= 1e-5
    xs = ((2 * torch.arange(8, device=DEV, dtype=torch.float64) - 7) / 8)
    ex = (hm.double().sum(-2) * xs).sum(-1)
    ey = (hm.double().sum(-1) * xs).sum(-1)
    assert (torch.stack([ex, ey], -1) - out.double()).abs().max().item() <= 2e-6
    hc, cc = hm.detach().cpu(), out.detach().cpu()
    total = onn.euclidean_loss(cc, t.cpu(), k.cpu()).item() + onn.js_reg_loss(hc, t.cpu(), 2.0 / 8, k.cpu()).item()
    assert abs(loss1.item() - total) <= 1e-5 * abs(total)
    g1 = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
    _, loss2 = step()
    g2 = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert loss1.item() == loss2.item() and torch.equal(g1, g2)
    assert torch.isfinite(g1).all() and float(g1.norm()) > 0
    params = list(m.parameters())
    d = [p.grad.clone() / g1.norm() for p in params]
    eps = 2.5e-4
    with torch.no_grad():
        for p, dp in zip(params, d):
            p.add_(dp, alpha=eps)
        lp = m.forward_loss(m(x), t, k).item()
        for p, dp in zip(params, d):
            p.add_(dp, alpha=-2 * eps)
        lm = m.forward_loss(m(x), t, k).item()
        for p, dp in zip(params, d):
            p.add_(dp, alpha=eps)
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - float(g1.norm())) <= 0.03 * float(g1.norm()), (fd, float(g1.norm()))
