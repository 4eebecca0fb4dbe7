"""Pin the oracle to the reference itself, imported in the build container.

Skipped wherever `/root/reference` is absent (e.g. the GPU box): there the
committed golden vectors (`tests/golden`, `test_oracle_golden.py`) are the pin.
Checks forward, loss and EVERY parameter gradient, fp64 (tight) and fp32.
"""
import pytest
import torch

import refimport
from dsnt_oracle import nn as onn
from dsnt_oracle import model as omodel
from dsnt_oracle import hourglass as ohg

REF = refimport.load_reference()
pytestmark = pytest.mark.skipif(REF is None, reason='/root/reference not present')


def _pair(base, dtype, **kw):
    ref_nn, ref_hg, ref_model = REF
    torch.manual_seed(0)
    ref = ref_model.build_mpii_pose_model(base=base, output_strat='dsnt', **kw).to(dtype)
    mine = omodel.build_mpii_pose_model(base=base, output_strat='dsnt', **kw).to(dtype)
    assert list(ref.state_dict().keys()) == list(mine.state_dict().keys())
    mine.load_state_dict(ref.state_dict())
    return ref, mine


@pytest.mark.parametrize('base,reg,dtype,tol', [
    ('hg1', 'none', torch.float64, 1e-10), ('hg2', 'js', torch.float64, 1e-10),
    ('hg2', 'js', torch.float32, 2e-4), ('hg2', 'var', torch.float64, 1e-10),
    ('hg2', 'kl', torch.float64, 1e-10), ('hg2', 'mse', torch.float64, 1e-10)])
def test_hourglass_model_forward_loss_grads(base, reg, dtype, tol):
    ref, mine = _pair(base, dtype, reg=reg, reg_coeff=1.0)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(2, 3, 64, 64, generator=g).to(dtype)
    target = (torch.rand(2, 16, 2, generator=g) * 2 - 1).to(dtype)
    mask = (torch.rand(2, 16, generator=g) < 0.8).to(dtype)
    ref.train(); mine.train()
    out_r, out_m = ref(x), mine(x)
    assert len(out_r) == len(out_m)
    for a, b in zip(out_r, out_m):
        assert (a - b).abs().max() <= tol
    for a, b in zip(ref.heatmaps_array, mine.heatmaps_array):
        assert (a - b).abs().max() <= tol
    loss_r = ref.forward_loss(out_r, target, mask)
    loss_m = mine.forward_loss(out_m, target, mask)
    assert abs(loss_r.item() - loss_m.item()) <= tol * max(1, abs(loss_r.item()))
    loss_r.backward(); loss_m.backward()
    for (n, p), (_, q) in zip(ref.named_parameters(), mine.named_parameters()):
        scale = max(1.0, p.grad.abs().max().item())
        assert (p.grad - q.grad).abs().max() <= tol * scale, n
    for (n, p), (_, q) in zip(ref.named_buffers(), mine.named_buffers()):
        assert (p.double() - q.double()).abs().max() <= tol, n
    assert (ref.compute_coords(out_r) - mine.compute_coords(out_m)).abs().max() <= max(tol, 1e-6)
    assert mine.compute_coords(out_m).dtype == torch.float32
    assert (ref.heatmaps - mine.heatmaps).abs().max() <= tol


def test_mask_none_and_tensor_part2():
    ref, mine = _pair('hg1', torch.float64, reg='js')
    x = torch.rand(2, 3, 64, 64, dtype=torch.float64)
    t = torch.rand(2, 16, 2, dtype=torch.float64)
    lr = ref.forward_loss(ref(x), t, None)
    lm = mine.forward_loss(mine(x), t, None)
    assert abs(lr.item() - lm.item()) <= 1e-10
    # inference.py:47 hands forward_part2 a bare 4-D tensor of batch 1
    hm = torch.randn(1, 16, 8, 8, dtype=torch.float64)
    a, b = ref.forward_part2(hm), mine.forward_part2(hm)
    assert len(a) == len(b) == 1 and (a[0] - b[0]).abs().max() <= 1e-12


def test_builder_quirks():
    ref_nn, ref_hg, ref_model = REF
    # hg builder defaults to the gauss strategy; resnet-style kwargs are filtered out
    m = omodel.build_mpii_pose_model(base='hg', dilate=2, truncate=1)
    r = ref_model.build_mpii_pose_model(base='hg', dilate=2, truncate=1)
    assert m.output_strat == r.output_strat == 'gauss'
    assert m.hg.num_stacks == r.hg.num_stacks == 2
    assert omodel.build_mpii_pose_model(base='hg8').hg.num_stacks == 8
    for bad in ('vgg', 'hgx'):
        with pytest.raises(Exception, match='unsupported base model type'):
            omodel.build_mpii_pose_model(base=bad)
        with pytest.raises(Exception, match='unsupported base model type'):
            ref_model.build_mpii_pose_model(base=bad)
    assert m.image_specs.size == r.image_specs.size == 256
    assert m.heatmap_size == r.heatmap_size == 64


@pytest.mark.parametrize('dilate,truncate,hm', [(0, 1, 14), (2, 0, 28), (0, 0, 7), (1, 1, 14)])
def test_resnet_wrapper_matches_reference_wrapper(dilate, truncate, hm):
    """Drive the reference's ResNetHumanPoseModel with the oracle's ResNet on both sides
    (torchvision is absent: SURVEY.md §8c); shapes follow tests/test_model.py:11-37."""
    import copy
    from dsnt_oracle import resnet as oresnet
    ref_nn, ref_hg, ref_model = REF
    torch.manual_seed(0)
    base = oresnet.resnet18().double()
    ref = ref_model.ResNetHumanPoseModel(copy.deepcopy(base), n_chans=16, dilate=dilate,
                                         truncate=truncate, reg='js')
    mine = omodel.ResNetHumanPoseModel(copy.deepcopy(base), n_chans=16, dilate=dilate,
                                       truncate=truncate, reg='js')
    mine.load_state_dict(ref.state_dict())
    ref.double(); mine.double()
    x = torch.randn(2, 3, 224, 224, dtype=torch.float64)
    t = torch.rand(2, 16, 2, dtype=torch.float64) * 2 - 1
    a, b = ref(x), mine(x)
    assert a.shape == (2, 16, 2) and ref.heatmaps.shape == (2, 16, hm, hm)
    assert mine.heatmaps.shape == (2, 16, hm, hm)
    assert (a - b).abs().max() <= 1e-10
    la, lb = ref.forward_loss(a, t, None), mine.forward_loss(b, t, None)
    assert abs(la.item() - lb.item()) <= 1e-10
    la.backward(); lb.backward()
    for (n, p), (_, q) in zip(ref.named_parameters(), mine.named_parameters()):
        assert (p.grad - q.grad).abs().max() <= 1e-9 * max(1.0, p.grad.abs().max().item()), n


@pytest.mark.parametrize('preact', ['softmax', 'thresholded_softmax', 'abs', 'relu', 'sigmoid'])
def test_preact_variants(preact):
    ref_nn, ref_hg, ref_model = REF
    x = torch.randn(3, 16, 8, 8, dtype=torch.float64)
    r = ref_model.HumanPoseModel()._hm_preact(x, preact)
    m = omodel.hm_preact(x, preact)
    assert (r - m).abs().max() <= 1e-14


def test_nn_functions_random():
    ref_nn, _, _ = REF
    g = torch.Generator().manual_seed(3)
    for dtype, tol in ((torch.float64, 1e-12), (torch.float32, 1e-6)):
        hm = torch.softmax(torch.randn(4, 16, 64 * 64, generator=g).to(dtype) * 3, -1)
        hm = hm.view(4, 16, 64, 64).requires_grad_()
        mu = (torch.rand(4, 16, 2, generator=g) * 2 - 1).to(dtype)
        mask = (torch.rand(4, 16, generator=g) < 0.7).to(dtype)
        for name in ('kl_reg_loss', 'js_reg_loss', 'mse_reg_loss', 'variance_reg_loss'):
            a = getattr(ref_nn, name)(hm, mu, 1 / 32, mask)
            b = getattr(onn, name)(hm, mu, 1 / 32, mask)
            assert abs(a.item() - b.item()) <= tol * max(1, abs(a.item())), name
            ga, = torch.autograd.grad(a, hm)
            gb, = torch.autograd.grad(b, hm)
            assert (ga - gb).abs().max() <= tol * max(1, ga.abs().max().item()), name
        assert (ref_nn.dsnt(hm) - onn.dsnt(hm)).abs().max() <= tol
        assert (ref_nn.make_gauss(mu, 64, 64, 1 / 32) - onn.make_gauss(mu, 64, 64, 1 / 32)
                ).abs().max() <= tol
        assert (ref_nn.softmax_2d(hm) - onn.softmax_2d(hm)).abs().max() <= tol
        a = ref_nn.thresholded_softmax(hm.flatten(-2), -0.5)
        b = onn.thresholded_softmax(hm.flatten(-2), -0.5)
        assert (a - b).abs().max() <= tol


def test_inference_flip_restatement_matches_reference_pieces():
    """The reference's `inference.py` imports absent packages (progressbar, tele, torchdata), so its
    `generate_predictions` cannot run here; pin the pieces the oracle restates instead: `reverse_tensor`
    (util.py:207-210) and the flip-average steps (inference.py:33-57) run by hand on a reference model."""
    ref_nn, ref_hg, ref_model = REF
    ref_util = refimport.load_reference_module('dsnt.util')
    from dsnt_oracle import inference as oinf, model as omodel
    from dsnt import synthetic
    t = torch.arange(24.).view(2, 3, 4)
    assert torch.equal(oinf.reverse_tensor(t, -1), ref_util.reverse_tensor(t, -1))
    rm = ref_model.build_mpii_pose_model(base='hg1', output_strat='dsnt', reg='js')
    om = omodel.build_mpii_pose_model(base='hg1', output_strat='dsnt', reg='js')
    synthetic.fill_state_dict(rm, seed=0)
    om.load_state_dict(rm.state_dict())
    rm.eval()
    x, _, _ = synthetic.batch(1, size=64, seed=3, mask_p=1.0)
    data = [{'input': x[0], 'transform_m': torch.eye(2, dtype=torch.float64) * 50,
             'transform_b': torch.ones(1, 2, dtype=torch.float64)}]
    got = oinf.generate_predictions(om, data, use_flipped=True, batch_size=1)
    with torch.no_grad():
        both = torch.cat([x, ref_util.reverse_tensor(x, -1)], 0)
        hm = rm.forward_part1(both)[-1]
        hm1, hm2 = hm.split(1)
        hm2 = ref_util.reverse_tensor(hm2, -1).index_select(-3, oinf.HFLIP_INDICES)
        coords = rm.compute_coords(rm.forward_part2((hm1 + hm2) / 2))
        want = torch.baddbmm(data[0]['transform_b'][None], coords.double(), data[0]['transform_m'][None])
    assert (got - want).abs().max().item() <= 1e-9


def test_heatmap_util_matches_reference():
    """`gauss` strategy helpers (util.py:70-198).  The reference's encode_heatmaps calls Python round() on a
    tensor element, which today's PyTorch rejects (TypeError: an API drift, not a denial), so encoding is
    pinned through the reference's own draw_gaussian at the documented rounded pixel + its known answers."""
    from dsnt_oracle import util as outil
    ref_util = refimport.load_reference_module('dsnt.util')
    g = torch.Generator().manual_seed(3)
    hm = torch.randn(3, 5, 16, 16, generator=g)
    hm[0, 0].zero_()                      # maximum not positive -> (0, 0) pixel
    hm[0, 1, 0, 7] = 9.0                  # border pixel: no neighbour offset
    hm[0, 2, 5, 5] = 9.0; hm[0, 2, 5, 4] = hm[0, 2, 5, 6] = 1.0   # equal neighbours: sign 0
    for nb in (True, False):
        assert torch.equal(ref_util.decode_heatmaps(hm.clone(), use_neighbours=nb),
                           outil.decode_heatmaps(hm, use_neighbours=nb))
    assert torch.equal(ref_util.get_preds(hm.clone()), outil.get_preds(hm))
    for (x, y, sigma, norm, clip) in [(4, 4, 1, False, None), (0, 4, 1, False, 7), (-3, 2, 1.5, False, 7),
                                      (-4, 2, 1, False, 7), (18, 3, 1, False, 7), (19, 3, 1, False, 7),
                                      (7, 15, 2, True, 7), (3.7, 2.2, 1, True, None)]:
        a = torch.zeros(16, 16); b = torch.zeros(16, 16)
        ref_util.draw_gaussian(a, x, y, sigma, normalize=norm, clip_size=clip)
        outil.draw_gaussian(b, x, y, sigma, normalize=norm, clip_size=clip)
        assert torch.equal(a, b), (x, y, sigma, norm, clip)
    coords = torch.rand(4, 6, 2, generator=g) * 2.4 - 1.2          # some joints off the map
    enc = outil.encode_heatmaps(coords, 16, 16, 1.25)
    px = coords.clone().add_(1); px[:, :, 0].mul_(16 / 2); px[:, :, 1].mul_(16 / 2); px.add_(-0.5)
    for i in range(4):
        for j in range(6):
            want = torch.zeros(16, 16)
            ref_util.draw_gaussian(want, round(px[i, j, 0].item()), round(px[i, j, 1].item()), 1.25,
                                   normalize=False, clip_size=7)
            assert torch.equal(want, enc[i, j])


def test_gauss_strategy_model_matches_reference_pieces():
    """HourglassHumanPoseModel with output_strat='gauss' (model.py:247-258, 268-269): the reference path
    itself needs a CUDA device (`.cuda()` at :253) and the drifted round(); check the oracle's loss against
    mse_loss over reference-drawn targets and the decode against the reference's decode_heatmaps."""
    import torch.nn.functional as F
    from dsnt_oracle import util as outil
    ref_util = refimport.load_reference_module('dsnt.util')
    torch.manual_seed(0)
    m = omodel.build_mpii_pose_model(base='hg2')                    # builder default: gauss
    assert m.output_strat == 'gauss'
    m.train()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 64, 64, generator=g)
    target = torch.rand(2, 16, 2, generator=g) * 2 - 1
    out = m(x)
    assert isinstance(out, list) and len(out) == 2 and out[0].shape == (2, 16, 16, 16)
    loss = m.forward_loss(out, target, None)
    tgt = outil.encode_heatmaps(target, 16, 16, 1.0)
    want = sum(F.mse_loss(o, tgt) for o in out)
    assert abs(loss.item() - want.item()) <= 1e-7
    loss.backward()
    assert all(p.grad is not None for p in m.parameters())
    assert torch.equal(m.compute_coords(out), ref_util.decode_heatmaps(out[-1].detach().clone()))
