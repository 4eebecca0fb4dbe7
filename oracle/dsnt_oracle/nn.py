"""Oracle restatement of the DSNT operator library (TEST INFRASTRUCTURE ONLY).

Follows `/root/reference/src/dsnt/nn.py` function by function (line ranges are
cited per function).  Written for modern PyTorch on CPU; no `Variable`, no
in-place tricks.  Numerics deliberately mirror the reference, including its
quirks: meshgrids are built by `torch.linspace` in the *default* dtype and then
cast (nn.py:35-44, 186-194), eps values 1e-12 / 1e-24, un-guarded sqrt in the
Euclidean loss (nn.py:113-114).
"""

import math

import torch
import torch.nn.functional as F


def _axis(n):
    """Pixel-centre coordinates of an axis of `n` cells spanning (-1, 1).

    nn.py:30-36 — linspace(-(n-1)/n, (n-1)/n, n) in the default dtype.
    """
    edge = (n - 1) / n
    return torch.linspace(-edge, edge, n)


def generate_xy(inp):
    """X and Y meshgrids broadcast to `inp`'s shape (nn.py:25-46)."""
    h, w = inp.shape[-2], inp.shape[-1]
    lead = [1] * (inp.dim() - 2)
    xs = _axis(w).view(*lead, 1, w).expand_as(inp).to(inp.dtype)
    ys = _axis(h).view(*lead, h, 1).expand_as(inp).to(inp.dtype)
    return xs, ys


def expectation_2d(values, probabilities):
    """Sum over the last two dims of values*probabilities (nn.py:49-63)."""
    weighted = values * probabilities
    return weighted.flatten(-2).sum(-1)


def dsnt(heatmaps):
    """Heatmaps [..., H, W] -> coordinates [..., 2] as (x, y) (nn.py:66-78)."""
    xs, ys = generate_xy(heatmaps)
    return torch.stack([expectation_2d(xs, heatmaps), expectation_2d(ys, heatmaps)], -1)


def masked_average(losses, mask=None):
    """sum(l*m)/clamp(sum(m),1), or mean with numel clamp (nn.py:81-94)."""
    if mask is None:
        return losses.sum() / max(losses.numel(), 1)
    return (losses * mask).sum() / mask.sum().clamp(1)


def euclidean_loss(actual, target, mask=None):
    """Masked mean of per-point L2 distances (nn.py:97-116)."""
    dist = (actual - target).pow(2).sum(-1).sqrt()
    return masked_average(dist, mask)


class _ThresholdedSoftmaxFn(torch.autograd.Function):
    """nn.py:119-139 — masked exp / (sum + eps), softmax-style backward."""

    @staticmethod
    def forward(ctx, inp, threshold, eps):
        keep = (inp >= threshold).to(inp.dtype)
        shifted = inp - inp.max(-1, keepdim=True)[0]
        e = shifted.exp() * keep
        out = e / (e.sum(-1, keepdim=True) + eps)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (out,) = ctx.saved_tensors
        dot = (grad_output * out).sum(-1, keepdim=True)
        return out * (grad_output - dot), None, None


def thresholded_softmax(inp, threshold=-math.inf, eps=1e-12):
    """nn.py:142-157."""
    return _ThresholdedSoftmaxFn.apply(inp, threshold, eps)


def softmax_2d(inp):
    """Softmax over the last two dims jointly (nn.py:160-165).

    The reference calls `F.softmax(flat)` with the implicit dim, which for a 2-D
    input is dim 1.
    """
    shape = inp.shape
    return F.softmax(inp.reshape(-1, shape[-1] * shape[-2]), dim=1).view(*shape)


def make_gauss(coords, width, height, sigma):
    """Normalised 2-D Gaussians centred on `coords` (nn.py:168-205)."""
    lead = [1] * (coords.dim() - 1)
    xs = _axis(width).view(*lead, 1, width).expand(*lead, height, width).to(coords.dtype)
    ys = _axis(height).view(*lead, height, 1).expand(*lead, height, width).to(coords.dtype)
    k = -0.5 * (1 / sigma) ** 2
    dx2 = (xs - coords[..., 0:1].unsqueeze(-1)) ** 2
    dy2 = (ys - coords[..., 1:2].unsqueeze(-1)) ** 2
    g = ((dx2 + dy2) * k).exp()
    total = g.sum(-1, keepdim=True).sum(-2, keepdim=True) + 1e-24
    return g / total


def _kl_2d(p, q, eps=1e-24):
    """nn.py:208-211."""
    return (p * ((p + eps).log() - (q + eps).log())).sum(-1).sum(-1)


def _js_2d(p, q, eps=1e-24):
    """nn.py:214-216."""
    m = 0.5 * (p + q)
    return 0.5 * _kl_2d(p, m, eps) + 0.5 * _kl_2d(q, m, eps)


def kl_reg_loss(heatmaps, mu_t, sigma_t, mask=None):
    """nn.py:219-234."""
    g = make_gauss(mu_t, heatmaps.size(-1), heatmaps.size(-2), sigma_t)
    return masked_average(_kl_2d(heatmaps, g), mask)


def js_reg_loss(heatmaps, mu_t, sigma_t, mask=None):
    """nn.py:237-252."""
    g = make_gauss(mu_t, heatmaps.size(-1), heatmaps.size(-2), sigma_t)
    return masked_average(_js_2d(heatmaps, g), mask)


def mse_reg_loss(heatmaps, mu_t, sigma_t, mask=None):
    """nn.py:255-271."""
    g = make_gauss(mu_t, heatmaps.size(-1), heatmaps.size(-2), sigma_t)
    return masked_average(((heatmaps - g) ** 2).sum(-1).sum(-1), mask)


def variance_reg_loss(heatmaps, mu_t, sigma_t, mask=None):
    """nn.py:274-298 — (Var_x - sigma^2)^2 + (Var_y - sigma^2)^2 per heatmap."""
    xs, ys = generate_xy(heatmaps)
    mx = expectation_2d(xs, heatmaps)[..., None, None]
    my = expectation_2d(ys, heatmaps)[..., None, None]
    var = torch.stack([expectation_2d((xs - mx) ** 2, heatmaps),
                       expectation_2d((ys - my) ** 2, heatmaps)], -1)
    return masked_average(((var - sigma_t ** 2) ** 2).sum(-1), mask)
