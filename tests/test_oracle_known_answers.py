"""Pin the oracle to every known-answer vector the reference's tests hold for the path.

Each test restates the expected values of a reference test (cited) and checks the
ORACLE (`oracle/dsnt_oracle`) against them.  The reference's harness runs with
default dtype double and tolerance 1e-5 (`tests/common.py:18,72`); same here.
"""
import pytest
import torch

from dsnt_oracle import nn as onn
from dsnt_oracle.evaluator import PCKhEvaluator

TOL = 1e-5


@pytest.fixture(autouse=True)
def _double_default():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(0)
    yield
    torch.set_default_dtype(old)


SIMPLE_INPUT = [[[[0.0, 0.0, 0.0, 0.0, 0.0],
                  [0.0, 0.0, 0.0, 0.1, 0.0],
                  [0.0, 0.0, 0.1, 0.6, 0.1],
                  [0.0, 0.0, 0.0, 0.1, 0.0],
                  [0.0, 0.0, 0.0, 0.0, 0.0]]]]
# d MSE(dsnt(h), (0.5,0.5)) / dh: 0.48 at the top-left, -0.04 per column, -0.20 per row
SIMPLE_GRAD = [[[[0.48 - 0.04 * c - 0.20 * r for c in range(5)] for r in range(5)]]]


def test_dsnt_forward_backward():  # reference tests/test_nn.py:10-50
    h = torch.tensor(SIMPLE_INPUT, requires_grad=True)
    out = onn.dsnt(h)
    assert (out.detach() - torch.tensor([[[0.4, 0.0]]])).abs().max() <= TOL
    torch.nn.functional.mse_loss(out, torch.tensor([[[0.5, 0.5]]])).backward()
    assert (h.grad - torch.tensor(SIMPLE_GRAD)).abs().max() <= TOL


def test_dsnt_batchless():  # tests/test_nn.py:52-66
    h = torch.tensor(SIMPLE_INPUT[0], requires_grad=True)
    out = onn.dsnt(h)
    assert out.shape == (1, 2)
    assert (out.detach() - torch.tensor([[0.4, 0.0]])).abs().max() <= TOL
    torch.nn.functional.mse_loss(out, torch.tensor([[0.5, 0.5]])).backward()
    assert (h.grad - torch.tensor(SIMPLE_GRAD[0])).abs().max() <= TOL


def test_euclidean_loss():  # tests/test_nn.py:86-109
    a = torch.tensor([[[3.0, 4], [3, 4]], [[3, 4], [3, 4]]], requires_grad=True)
    loss = onn.euclidean_loss(a, torch.zeros(2, 2, 2))
    loss.backward()
    assert abs(loss.item() - 5.0) <= TOL
    assert (a.grad - torch.tensor([0.15, 0.20]).expand(2, 2, 2)).abs().max() <= TOL


def test_euclidean_loss_mask():  # tests/test_nn.py:111-130
    out = torch.tensor([[[0.0, 0], [1, 1], [0, 0]], [[1, 1], [0, 0], [0, 0]]])
    mask = torch.tensor([[1.0, 0, 1], [0, 1, 1]])
    assert abs(onn.euclidean_loss(out, torch.zeros(2, 3, 2), mask).item()) <= TOL


def test_thresholded_softmax_forward():  # tests/test_nn.py:134-149
    got = onn.thresholded_softmax(torch.tensor([2.0, 1, 3]), 1.5)
    assert (got - torch.tensor([0.26894142, 0, 0.73105858])).abs().max() <= TOL
    got = onn.thresholded_softmax(torch.tensor([[2.0, 1, 3], [4, 0, 0]]), 1.5)
    want = torch.tensor([[0.26894142, 0, 0.73105858], [1, 0, 0]])
    assert (got - want).abs().max() <= TOL


@pytest.mark.parametrize('shape', [(20,), (3, 20)])
def test_thresholded_softmax_gradcheck(shape):  # tests/test_nn.py:140-154
    x = torch.randn(*shape, requires_grad=True)
    assert torch.autograd.gradcheck(lambda t: onn.thresholded_softmax(t, 0), (x,))


def test_make_gauss():  # tests/test_nn.py:158-167
    want = torch.tensor([[0.0030, 0.0133, 0.0219, 0.0133, 0.0030],
                         [0.0133, 0.0596, 0.0983, 0.0596, 0.0133],
                         [0.0219, 0.0983, 0.1621, 0.0983, 0.0219],
                         [0.0133, 0.0596, 0.0983, 0.0596, 0.0133],
                         [0.0030, 0.0133, 0.0219, 0.0133, 0.0030]])
    got = onn.make_gauss(torch.tensor([0.0, 0.0]), 5, 5, sigma=0.4)
    assert (got - want).abs().max() <= 1e-4


@pytest.mark.parametrize('loss_fn,shift_mean', [
    (onn.kl_reg_loss, True), (onn.mse_reg_loss, True), (onn.js_reg_loss, True),
    (onn.variance_reg_loss, False)])
def test_reg_loss_minimum(loss_fn, shift_mean):  # tests/test_nn.py:170-239
    mean, std = torch.tensor([0.0, 0.0]), 0.4

    def calc(m, s):
        return loss_fn(onn.make_gauss(m, 5, 5, sigma=s), mean, std, mask=None).item()

    lo = calc(mean, std)
    assert abs(lo) <= 1e-3
    assert calc(mean, std + 0.2) > lo + 1e-3
    assert calc(mean, std - 0.2) > lo + 1e-3
    if shift_mean:
        assert calc(mean + 0.1, std) > lo + 1e-3
        assert calc(mean - 0.1, std) > lo + 1e-3


def test_kl_reg_loss_mask():  # tests/test_nn.py:204-224
    t = torch.zeros(2, 4, 4)
    t[0, 2, 3] = t[0, 3, 2] = 0.1
    t[0, 3, 3] = 0.8
    t[1, 0, 0] = 0.8
    t[1, 0, 1] = t[1, 1, 0] = 0.1
    got = onn.kl_reg_loss(t, torch.tensor([[1.0, 1], [0, 0]]), 1, torch.tensor([1.0, 0]))
    assert abs(got.item() - 1.2228811717796824) <= TOL


def test_pckh_distance():  # tests/test_evaluator.py:8-16
    d = PCKhEvaluator.calculate_pckh_distance(
        torch.tensor([951.84, 580.64]), torch.tensor([804.0, 711]), 117.962)
    assert abs(float(d) - 1.6709) <= 1e-4


def test_pckh_add():  # tests/test_evaluator.py:18-39
    ev = PCKhEvaluator(threshold=0.5)
    pred = torch.tensor([[[951.84, 580.64]], [[317.76, 406.75]], [[float('inf')] * 2]])
    target = torch.tensor([[[804.0, 711]], [[317, 412]], [[float('nan')] * 2]])
    ev.add(pred, target, torch.tensor([[1.0], [1], [0]]),
           torch.tensor([117.962, 44.046, 78.481]))
    assert ev.meters['all'].value()[0] == 0.5


# ---------------------------------------------------------------- heat-map ("gauss") strategy: util.py
from dsnt_oracle import util as outil  # noqa: E402

_G = [0.00034, 0.01111, 0.13534, 0.60653, 1.00000, 0.60653, 0.13534, 0.01111, 0.00034]   # exp(-d^2/2), d=-4..4


def test_draw_gaussian():  # tests/test_util.py:8-24
    expected = torch.tensor([[[_G[i] * _G[j] for j in range(9)] for i in range(9)]], dtype=torch.float32)
    actual = torch.zeros(1, 9, 9, dtype=torch.float32)
    outil.draw_gaussian(actual, 4, 4, 1, normalize=False)
    assert (expected - actual).abs().max().item() <= 1.5e-5     # the table is rounded to 5 decimals


_CLIPPED = [[0.00000, 0.00000, 0.00000, 0.00000, 0.00000],
            [0.01111, 0.00674, 0.00150, 0.00012, 0.00000],
            [0.13534, 0.08208, 0.01832, 0.00150, 0.00000],
            [0.60653, 0.36788, 0.08208, 0.00674, 0.00000],
            [1.00000, 0.60653, 0.13534, 0.01111, 0.00000]]


def test_draw_gaussian_clipped():  # tests/test_util.py:26-38
    actual = torch.zeros(1, 5, 5, dtype=torch.float32)
    outil.draw_gaussian(actual, 0, 4, 1, normalize=False, clip_size=7)
    assert (torch.tensor([_CLIPPED], dtype=torch.float32) - actual).abs().max().item() <= TOL


def test_encode_heatmaps():  # tests/test_util.py:40-53
    coords = torch.tensor([[[-0.8, 0.8]]], dtype=torch.float32)
    actual = outil.encode_heatmaps(coords, 5, 5)
    assert actual.dtype == torch.float32
    assert (torch.tensor([[_CLIPPED]], dtype=torch.float32) - actual).abs().max().item() <= TOL
    assert torch.equal(coords, torch.tensor([[[-0.8, 0.8]]], dtype=torch.float32))   # works on a copy


def test_decode_heatmaps():  # tests/test_util.py:55-64
    heatmaps = torch.tensor([[[[0.0, 0.9], [0.0, 0.1]]]], dtype=torch.float32)
    actual = outil.decode_heatmaps(heatmaps)
    assert (torch.tensor([[[0.5, -0.5]]], dtype=torch.float32) - actual).abs().max().item() <= 1e-7


def test_decode_heatmaps_use_neighbours():  # tests/test_util.py:66-77
    heatmaps = torch.tensor([[[[0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0], [0.0, 0.9, 0.1, 0.0],
                               [0.0, 0.1, 0.0, 0.0]]]], dtype=torch.float32)
    actual = outil.decode_heatmaps(heatmaps, use_neighbours=True)
    assert (torch.tensor([[[-0.125, 0.375]]], dtype=torch.float32) - actual).abs().max().item() <= 1e-7
