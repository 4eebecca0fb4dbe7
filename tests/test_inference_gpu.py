"""Flip-augmented test-time prediction (reference src/dsnt/inference.py:12-68):
the HIP path's `dsnt.inference.generate_predictions` against the oracle's on the same weights, running
statistics and synthetic samples; tolerance = the 1e-4 coordinate bar scaled by the back-projection."""
import pytest
import torch

from dsnt import synthetic

pytestmark = pytest.mark.gpu


def _dataset(n, size, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        x, _, _ = synthetic.batch(1, size=size, seed=seed + i, mask_p=1.0)
        m = torch.eye(2, dtype=torch.float64) * (100.0 + 10 * i) + 3.0 * torch.rand(2, 2, generator=g, dtype=torch.float64)
        b = 200.0 * torch.rand(1, 2, generator=g, dtype=torch.float64)
        out.append({'input': x[0], 'transform_m': m, 'transform_b': b})
    return out


@pytest.mark.parametrize('base', ['hg1', 'hg2'])
@pytest.mark.parametrize('use_flipped', [True, False])
def test_generate_predictions_vs_oracle(base, use_flipped):
    from dsnt.model import build_mpii_pose_model
    from dsnt import inference
    from dsnt_oracle import model as omodel, inference as oinference
    m = build_mpii_pose_model(base=base, output_strat='dsnt', reg='js')
    o = omodel.build_mpii_pose_model(base=base, output_strat='dsnt', reg='js')
    synthetic.fill_state_dict(m, seed=0)
    # realistic running statistics: one train-mode forward with momentum 1, then share the state
    m.cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 1.0
    x, _, _ = synthetic.batch(2, size=128, seed=5, mask_p=1.0)
    with torch.no_grad():
        m(x.cuda())
    o.load_state_dict({k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    data = _dataset(3, 128, seed=11)
    bs = 1 if use_flipped else 2          # the ragged last batch (2 + 1) is part of the contract

    class Meter:
        total, n = 0.0, 0

        def add(self, v):
            self.total += v
            self.n += 1
    meter = Meter()
    got = inference.generate_predictions(m, data, use_flipped=use_flipped, batch_size=bs, time_meter=meter)
    want = oinference.generate_predictions(o, data, use_flipped=use_flipped, batch_size=bs)
    assert got.dtype == torch.float64 and got.shape == (3, 16, 2) and got.device.type == 'cpu'
    assert meter.n == (3 if use_flipped else 2) and meter.total > 0
    scale = max(float(d['transform_m'].abs().max()) for d in data)
    assert (got - want).abs().max().item() <= 2 * 1e-4 * scale
    assert not m.training


def test_flip_needs_batch_one():
    from dsnt import inference
    with pytest.raises(AssertionError, match='batch_size=1'):
        inference.generate_predictions(None, [], use_flipped=True, batch_size=2)
    assert inference.HFLIP_INDICES.tolist() == [5, 4, 3, 2, 1, 0, 6, 7, 8, 9, 15, 14, 13, 12, 11, 10]
    assert sorted(inference.HFLIP_INDICES.tolist()) == list(range(16))


@pytest.mark.parametrize('use_flipped', [True, False])
def test_generate_predictions_heatmap_strategy(use_flipped):
    """The same path with the builder's default `gauss` strategy: `forward_part2` passes the (flip-averaged)
    heat-maps through and `compute_coords` decodes their arg-max (model.py:268-269).  Arg-max of nearly equal
    pixels may flip between two fp32 implementations, so most joints must agree to the coordinate bar and
    every joint to within a heat-map pixel."""
    from dsnt.model import build_mpii_pose_model
    from dsnt import inference
    from dsnt_oracle import model as omodel, inference as oinference
    m = build_mpii_pose_model(base='hg2')
    o = omodel.build_mpii_pose_model(base='hg2')
    assert m.output_strat == 'gauss'
    synthetic.fill_state_dict(m, seed=0)
    m.cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 1.0
    x, _, _ = synthetic.batch(2, size=128, seed=5, mask_p=1.0)
    with torch.no_grad():
        m(x.cuda())
    o.load_state_dict({k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    data = _dataset(3, 128, seed=11)
    bs = 1 if use_flipped else 2
    got = inference.generate_predictions(m, data, use_flipped=use_flipped, batch_size=bs)
    want = oinference.generate_predictions(o, data, use_flipped=use_flipped, batch_size=bs)
    scale = max(float(d['transform_m'].abs().max()) for d in data)
    err = (got - want).abs().amax(-1)
    assert (err <= 2e-4 * scale).float().mean().item() >= 0.9
    assert err.max().item() <= 2.6 * (2.0 / 32) * scale          # one pixel + quarter-pixel shifts of a 32x32 map
