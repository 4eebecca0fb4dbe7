"""HIP heat-map ("gauss") strategy vs the oracle and the reference's known answers (SURVEY 8 f-4:
`/root/reference/src/dsnt/util.py:129-198`, `/root/reference/src/dsnt/model.py:147-161, 247-269`).

Tolerances: target bumps 1e-6 absolute (device expf vs libm expf), losses 1e-5 relative, gradients 1e-5 relative
to the largest entry; decoding is index / sign arithmetic and must be exactly equal on the same heat-maps.
"""
import pytest
import torch
import torch.nn.functional as F

from dsnt import synthetic

pytestmark = pytest.mark.gpu

_CLIPPED = [[0.00000, 0.00000, 0.00000, 0.00000, 0.00000],
            [0.01111, 0.00674, 0.00150, 0.00012, 0.00000],
            [0.13534, 0.08208, 0.01832, 0.00150, 0.00000],
            [0.60653, 0.36788, 0.08208, 0.00674, 0.00000],
            [1.00000, 0.60653, 0.13534, 0.01111, 0.00000]]


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def test_known_answers(dev):
    from dsnt import util as du
    # tests/test_util.py:40-53
    coords = torch.tensor([[[-0.8, 0.8]]], device=dev)
    enc = du.encode_heatmaps(coords, 5, 5)
    assert (enc.cpu() - torch.tensor([[_CLIPPED]])).abs().max().item() <= 1e-5
    assert torch.equal(coords.cpu(), torch.tensor([[[-0.8, 0.8]]]))
    # tests/test_util.py:55-64
    hm = torch.tensor([[[[0.0, 0.9], [0.0, 0.1]]]], device=dev)
    assert (du.decode_heatmaps(hm).cpu() - torch.tensor([[[0.5, -0.5]]])).abs().max().item() <= 1e-7
    # tests/test_util.py:66-77
    hm = torch.tensor([[[[0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0], [0.0, 0.9, 0.1, 0.0], [0.0, 0.1, 0.0, 0.0]]]],
                      device=dev)
    assert (du.decode_heatmaps(hm, use_neighbours=True).cpu() - torch.tensor([[[-0.125, 0.375]]])).abs().max().item() <= 1e-7


@pytest.mark.parametrize('H,W,sigma', [(64, 64, 1.0), (16, 12, 1.25), (7, 9, 2.0)])
def test_encode_matches_oracle(dev, H, W, sigma):
    from dsnt import util as du
    from dsnt_oracle import util as ou
    coords = synthetic.tensor('hm.coords%d' % H, (5, 16, 2), seed=21, kind='uniform') * 1.15   # some off the map
    coords[0, 0] = torch.tensor([-1.0, 1.0]); coords[0, 1] = torch.tensor([0.0, 0.0])       # .5 pixel ties
    want = ou.encode_heatmaps(coords, W, H, sigma)
    got = du.encode_heatmaps(coords.to(dev), W, H, sigma).cpu()
    assert got.shape == want.shape == (5, 16, H, W)
    assert (got - want).abs().max().item() <= 1e-6
    assert torch.equal(got == 0, want == 0)


def test_mse_loss_and_gradient(dev):
    from dsnt import util as du
    from dsnt_oracle import util as ou
    hm = synthetic.tensor('hm.pred', (3, 16, 64, 64), seed=22) * 0.2
    coords = synthetic.tensor('hm.tgt', (3, 16, 2), seed=22, kind='uniform')
    ho = hm.clone().requires_grad_()
    lo = F.mse_loss(ho, ou.encode_heatmaps(coords, 64, 64, 1.0))
    (lo * 3.0).backward()
    hg = hm.to(dev).requires_grad_()
    lg = du.heatmap_mse_loss(hg, coords.to(dev), 1.0)
    (lg * 3.0).backward()
    assert abs(lg.item() - lo.item()) <= 1e-5 * abs(lo.item())
    assert (hg.grad.cpu() - ho.grad).abs().max().item() <= 1e-5 * ho.grad.abs().max().item()
    with pytest.raises(RuntimeError):
        du.heatmap_mse_loss(hg, coords[:2].to(dev), 1.0)
    with pytest.raises(RuntimeError):
        du.encode_heatmaps(coords, 64, 64)          # CPU tensor: no fallback


def test_decode_matches_oracle(dev):
    from dsnt import util as du
    from dsnt_oracle import util as ou
    hm = synthetic.tensor('hm.dec', (4, 16, 64, 64), seed=23)
    hm[0, 0].zero_()                                  # maximum not positive
    hm[0, 1] = -hm[0, 1].abs() - 1                    # all negative
    hm[0, 2, 0, 17] = 50.0                            # border row
    hm[0, 3, 20, 20] = 50.0; hm[0, 3, 20, 19] = hm[0, 3, 20, 21] = 1.0      # equal neighbours
    hm[0, 4, 30, 30] = hm[0, 4, 31, 5] = 50.0         # tie: first index wins
    hm[0, 5, 63, 63] = 50.0                           # corner
    for nb in (True, False):
        want = ou.decode_heatmaps(hm, use_neighbours=nb)
        got = du.decode_heatmaps(hm.to(dev), use_neighbours=nb).cpu()
        assert torch.equal(got, want)
    assert torch.equal(du.get_preds(hm.to(dev)).cpu(), ou.get_preds(hm))
    small = synthetic.tensor('hm.dec2', (2, 3, 16, 16), seed=24)
    assert torch.equal(du.decode_heatmaps(small.to(dev)).cpu(), ou.decode_heatmaps(small))


@pytest.mark.parametrize('base', ['hg2', 'resnet18'])
def test_gauss_strategy_train_step(dev, base):
    """The builder's default strategy for hourglass models (model.py:346): loss, every gradient direction and the
    decoded coordinates against the oracle."""
    from dsnt.model import build_mpii_pose_model
    from dsnt_oracle import model as omodel
    from dsnt_oracle import util as ou
    kw = {} if base.startswith('hg') else {'output_strat': 'gauss', 'dilate': 2}
    m = build_mpii_pose_model(base=base, **kw)
    o = omodel.build_mpii_pose_model(base=base, **kw)
    assert m.output_strat == o.output_strat == 'gauss'
    synthetic.fill_state_dict(m, seed=0)
    synthetic.fill_state_dict(o, seed=0)
    m.to(dev).train()
    o.train()
    size = 128 if base.startswith('hg') else 224
    x, t, k = synthetic.batch(2, size=size, seed=1, mask_p=0.9)
    out = m(x.to(dev))
    loss = m.forward_loss(out, t.to(dev), k.to(dev))
    loss.backward()
    oo = o(x)
    lo = o.forward_loss(oo, t, k)
    lo.backward()
    assert abs(loss.item() - lo.item()) <= 2e-5 * max(1.0, abs(lo.item()))
    gm = torch.cat([p.grad.reshape(-1).cpu() for p in m.parameters()]).double()
    go = torch.cat([p.grad.reshape(-1) for p in o.parameters()]).double()
    # (ReLU on, batch 2: one mask bit flipped between two correct fp32 implementations moves the flat gradient by ~1e-4 in
    # cosine — tests/test_model_gpu.py::_NoRelu; the flip-tolerant bar of the other ReLU-on tests)
    assert float(gm @ go / (gm.norm() * go.norm())) > 0.999
    last = out[-1] if isinstance(out, list) else out
    coords = m.compute_coords(out)
    assert coords.device.type == 'cpu' and coords.shape == (2, 16, 2)
    assert torch.equal(coords, ou.decode_heatmaps(last.detach().cpu()))       # same maps -> same decoding
    # against the oracle's own maps the arg-max may flip between near-equal pixels; most joints agree
    agree = (coords - o.compute_coords(oo)).abs().amax(-1) <= 1e-6
    assert agree.float().mean().item() >= 0.9
