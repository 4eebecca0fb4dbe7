"""DSNT operator library on MI355X — same surface as the reference's `dsnt.nn`.

Every public name, positional/keyword argument and default of
`/root/reference/src/dsnt/nn.py` is kept (`dsnt`, `euclidean_loss`, `thresholded_softmax`,
`make_gauss`, `kl_reg_loss`, `js_reg_loss`, `mse_reg_loss`, `variance_reg_loss`, ...), so
`train.py` / `infer.py` style callers import it unchanged.  Underneath, each op is a
`torch.autograd.Function` that enqueues hand-written HIP kernels from libdsnt_hip.so on the
current stream (csrc/head.hip).  Tensors must be fp32 and resident on the HIP device: there is
no CPU fallback, by design.
"""
import math

import numpy as np
import torch
from torch.autograd import Function

from . import _lib
from ._lib import ptr, f32, call

PREACT_MODES = {'softmax': 0, 'thresholded_softmax': 1, 'abs': 2, 'relu': 3, 'sigmoid': 4}
REG_KINDS = {'js': 0, 'kl': 1, 'mse': 2, 'var': 3}


def _rows(t, trailing):
    n = 1
    for s in t.shape[:t.dim() - trailing]:
        n *= s
    return n


# ------------------------------------------------------------------ meshgrids (nn.py:25-63)
def generate_xy(inp):
    """X and Y meshgrids expanded to `inp`'s shape (reference nn.py:25-46).

    Compatibility helper only — the kernels evaluate x_w = (2w-(W-1))/W in registers and never
    read a materialised meshgrid.
    """
    h, w = inp.shape[-2], inp.shape[-1]
    lead = [1] * (inp.dim() - 2)
    xs = torch.linspace(-(w - 1) / w, (w - 1) / w, w, device=inp.device).view(*lead, 1, w)
    ys = torch.linspace(-(h - 1) / h, (h - 1) / h, h, device=inp.device).view(*lead, h, 1)
    return xs.expand_as(inp).to(inp.dtype), ys.expand_as(inp).to(inp.dtype)


def expectation_2d(values, probabilities):
    """sum(values * probabilities) over the last two dims (reference nn.py:49-63)."""
    return (values * probabilities).flatten(-2).sum(-1)


# ------------------------------------------------------------------ dsnt (nn.py:66-78)
class _Dsnt(Function):
    @staticmethod
    def forward(ctx, heatmaps):
        hm = f32(heatmaps).contiguous()
        h, w = hm.shape[-2], hm.shape[-1]
        rows = _rows(hm, 2)
        coords = torch.empty(*hm.shape[:-2], 2, device=hm.device, dtype=hm.dtype)
        call('dsnt_expect_fwd', ptr(hm), ptr(coords), rows, h, w)
        ctx.shape = hm.shape
        return coords

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        ghm = torch.empty(ctx.shape, device=g.device, dtype=g.dtype)
        call('dsnt_expect_bwd', ptr(g), ptr(ghm), _rows(ghm, 2), ctx.shape[-2], ctx.shape[-1])
        return ghm


def dsnt(heatmaps):
    """Differentiable spatial to numerical transform: [..., H, W] -> [..., 2] as (x, y)."""
    return _Dsnt.apply(heatmaps)


# ------------------------------------------------------------------ 'fc' output strategy (model.py:222-223, 293-303)
class _Fc2(Function):
    @staticmethod
    def forward(ctx, heatmaps, weight, bias):
        hm = f32(heatmaps).contiguous()
        hw = hm.shape[-2] * hm.shape[-1]
        rows = _rows(hm, 2)
        if tuple(weight.shape) != (2, hw):
            raise RuntimeError('dsnt.nn.fc_coords: weight must be [2, %d], got %s' % (hw, tuple(weight.shape)))
        w = f32(weight).contiguous()
        b = None if bias is None else f32(bias).contiguous()
        out = torch.empty(*hm.shape[:-2], 2, device=hm.device, dtype=hm.dtype)
        call('dsnt_fc2_fwd', ptr(hm), ptr(w), ptr(b), ptr(out), rows, hw)
        ctx.save_for_backward(hm, w)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        hm, w = ctx.saved_tensors
        g = g.contiguous()
        hw = w.shape[1]
        ghm = torch.empty_like(hm) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w)
        gb = torch.empty(2, device=w.device, dtype=w.dtype) if ctx.has_bias else None
        call('dsnt_fc2_bwd', ptr(g), ptr(hm), ptr(w), ptr(ghm), ptr(gw), ptr(gb), _rows(hm, 2), hw)
        return ghm, gw, gb


def fc_coords(heatmaps, weight, bias=None):
    """`out_fc(hm.view(-1, H*W)).view(..., 2)`: the reference's fully-connected output strategy."""
    return _Fc2.apply(heatmaps, weight, bias)


# ------------------------------------------------------------------ masked average (nn.py:81-94)
class _MaskedAverage(Function):
    @staticmethod
    def forward(ctx, losses, mask):
        l = f32(losses).contiguous()
        m = None if mask is None else f32(mask).contiguous()
        if m is not None and m.shape != l.shape:
            m = m.expand_as(l).contiguous()
        out2 = torch.empty(2, device=l.device, dtype=l.dtype)
        call('dsnt_masked_avg_fwd', ptr(l), ptr(m), ptr(out2), l.numel())
        ctx.save_for_backward(out2, m)
        ctx.shape = l.shape
        return out2[0].clone()

    @staticmethod
    def backward(ctx, g):
        out2, m = ctx.saved_tensors
        gl = torch.empty(ctx.shape, device=g.device, dtype=g.dtype)
        call('dsnt_masked_avg_bwd', ptr(g.contiguous().view(1)), ptr(m), ptr(out2), ptr(gl),
             gl.numel())
        return gl, None


def masked_average(losses, mask=None):
    """sum(l*m)/clamp(sum m, 1); plain mean (numel clamped to >= 1) without a mask."""
    return _MaskedAverage.apply(losses, mask)


# ------------------------------------------------------------------ euclidean loss (nn.py:97-116)
class _EuclidDist(Function):
    @staticmethod
    def forward(ctx, actual, target):
        a = f32(actual).contiguous()
        t = f32(target).expand_as(a).contiguous()
        d = a.shape[-1]
        n = a.numel() // d
        dist = torch.empty(a.shape[:-1], device=a.device, dtype=a.dtype)
        call('dsnt_euclid_fwd', ptr(a), ptr(t), ptr(dist), n, d)
        ctx.save_for_backward(a, t, dist)
        return dist

    @staticmethod
    def backward(ctx, g):
        a, t, dist = ctx.saved_tensors
        ga = torch.empty_like(a)
        call('dsnt_euclid_bwd', ptr(a), ptr(t), ptr(dist), ptr(g.contiguous()), ptr(ga),
             dist.numel(), a.shape[-1])
        gt = -ga if ctx.needs_input_grad[1] else None
        return ga, gt


def euclidean_loss(actual, target, mask=None):
    """Average Euclidean distance between predicted and target points.

    actual, target: ([batches x] n x d); mask: ([batches x] n) or None.
    """
    return masked_average(_EuclidDist.apply(actual, target), mask)


# ------------------------------------------------------------------ heat-map normalisation
class _Preact(Function):
    """model.py:24-45 normalisations; mode 1 is nn.py:119-139."""

    @staticmethod
    def forward(ctx, inp, mode, threshold, eps):
        x = f32(inp).contiguous()
        hw = x.shape[-1]
        y = torch.empty_like(x)
        call('dsnt_preact_fwd', ptr(x), ptr(y), x.numel() // hw, hw, mode, threshold, eps)
        ctx.mode, ctx.threshold, ctx.eps = mode, threshold, eps
        ctx.save_for_backward(x if mode >= 2 else None, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        gx = torch.empty_like(y)
        hw = y.shape[-1]
        call('dsnt_preact_bwd', ptr(x), ptr(y), ptr(g.contiguous()), ptr(gx), y.numel() // hw, hw,
             ctx.mode, ctx.threshold, ctx.eps)
        return gx, None, None, None


class ThresholdedSoftmax(Function):
    """Same call convention as the reference's Function (nn.py:119-139)."""

    @staticmethod
    def forward(ctx, inp, threshold=-np.inf, eps=1e-12):
        x = f32(inp).contiguous()
        hw = x.shape[-1]
        y = torch.empty_like(x)
        call('dsnt_preact_fwd', ptr(x), ptr(y), x.numel() // hw, hw, 1, float(threshold), float(eps))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, grad_output):
        (y,) = ctx.saved_tensors
        gx = torch.empty_like(y)
        hw = y.shape[-1]
        call('dsnt_preact_bwd', None, ptr(y), ptr(grad_output.contiguous()), ptr(gx),
             y.numel() // hw, hw, 1, 0.0, 0.0)
        return gx, None, None


def thresholded_softmax(inp, threshold=-np.inf, eps=1e-12):
    """Softmax over the last dim with inputs below `threshold` forced to exactly zero."""
    return ThresholdedSoftmax.apply(inp, threshold, eps)


def softmax_2d(inp):
    """Softmax with the last two tensor dimensions combined (reference nn.py:160-165)."""
    size = inp.size()
    flat = inp.reshape(-1, size[-1] * size[-2])
    return _Preact.apply(flat, 0, 0.0, 0.0).view(*size)


def hm_preact(x, preact):
    """`HumanPoseModel._hm_preact` (model.py:24-45): [..., J, H, W] -> [-1, J, H, W]."""
    if preact not in PREACT_MODES:
        raise Exception('unrecognised heatmap preactivation function: {}'.format(preact))
    n_chans, h, w = x.size(-3), x.size(-2), x.size(-1)
    flat = x.reshape(-1, h * w)
    mode = PREACT_MODES[preact]
    thr = -0.5 if mode == 1 else 0.0
    eps = 0.0 if mode == 0 else 1e-12
    return _Preact.apply(flat, mode, thr, eps).view(-1, n_chans, h, w)


# ------------------------------------------------------------------ make_gauss (nn.py:168-205)
class _MakeGauss(Function):
    @staticmethod
    def forward(ctx, coords, width, height, sigma):
        c = f32(coords).contiguous()
        out = torch.empty(*c.shape[:-1], height, width, device=c.device, dtype=c.dtype)
        call('dsnt_make_gauss', ptr(c), ptr(out), c.numel() // 2, height, width, float(sigma))
        ctx.save_for_backward(c)
        ctx.hw, ctx.sigma = (height, width), float(sigma)
        return out

    @staticmethod
    def backward(ctx, g):
        (c,) = ctx.saved_tensors
        gc = torch.empty_like(c)
        call('dsnt_make_gauss_bwd', ptr(c), ptr(f32(g).contiguous()), ptr(gc), c.numel() // 2, ctx.hw[0], ctx.hw[1],
             ctx.sigma)
        return gc, None, None, None


def make_gauss(coords, width, height, sigma):
    """Normalised 2-D Gaussians with means `coords` (normalised units) and std `sigma`."""
    return _MakeGauss.apply(coords, width, height, sigma)


def _kl_2d(p, q, eps=1e-24):
    """Compatibility helper (nn.py:208-211); the reg-loss functions below use fused kernels."""
    return (p * ((p + eps).log() - (q + eps).log())).sum(-1).sum(-1)


def _js_2d(p, q, eps=1e-24):
    """Compatibility helper (nn.py:214-216)."""
    m = 0.5 * (p + q)
    return 0.5 * _kl_2d(p, m, eps) + 0.5 * _kl_2d(q, m, eps)


# ------------------------------------------------------------------ regularisers (nn.py:219-298)
class _RegRows(Function):
    """Per-heat-map divergence from the target Gaussian (or variance error), fused."""

    @staticmethod
    def forward(ctx, heatmaps, mu_t, sigma_t, kind):
        hm = f32(heatmaps).contiguous()
        h, w = hm.shape[-2], hm.shape[-1]
        rows = _rows(hm, 2)
        mu = None
        if kind != 3:
            mu = f32(mu_t).expand(*hm.shape[:-2], 2).contiguous()
        out = torch.empty(hm.shape[:-2], device=hm.device, dtype=hm.dtype)
        call('dsnt_reg_fwd', ptr(hm), ptr(mu), ptr(out), rows, h, w, float(sigma_t), kind)
        ctx.save_for_backward(hm, mu)
        ctx.sigma, ctx.kind = float(sigma_t), kind
        ctx.mu_shape = tuple(mu_t.shape) if torch.is_tensor(mu_t) else None
        return out

    @staticmethod
    def backward(ctx, g):
        hm, mu = ctx.saved_tensors
        g = f32(g).contiguous()
        rows, h, w = _rows(hm, 2), hm.shape[-2], hm.shape[-1]
        ghm = gmu = None
        if ctx.needs_input_grad[0]:
            ghm = torch.empty_like(hm)
            call('dsnt_reg_bwd', ptr(hm), ptr(mu), ptr(g), ptr(ghm), rows, h, w, ctx.sigma, ctx.kind)
        if ctx.needs_input_grad[1]:
            # the reference's target is make_gauss(mu_t) inside autograd (nn.py:219-271): differentiable in mu_t
            if ctx.kind == 3:                       # the variance regulariser never reads mu_t
                gmu = torch.zeros(ctx.mu_shape, device=hm.device, dtype=hm.dtype)
            else:
                gmu = torch.empty_like(mu)
                call('dsnt_reg_bwd_mu', ptr(hm), ptr(mu), ptr(g), ptr(gmu), rows, h, w, ctx.sigma, ctx.kind)
                gmu = gmu.sum_to_size(ctx.mu_shape)  # mu_t may have been broadcast over leading dimensions
        return ghm, gmu, None, None


def kl_reg_loss(heatmaps, mu_t, sigma_t, mask=None):
    """Average KL(heatmap || target Gaussian)."""
    return masked_average(_RegRows.apply(heatmaps, mu_t, sigma_t, 1), mask)


def js_reg_loss(heatmaps, mu_t, sigma_t, mask=None):
    """Average Jensen-Shannon divergence between heatmaps and target Gaussians."""
    return masked_average(_RegRows.apply(heatmaps, mu_t, sigma_t, 0), mask)


def mse_reg_loss(heatmaps, mu_t, sigma_t, mask=None):
    """Average summed squared error between heatmaps and target Gaussians."""
    return masked_average(_RegRows.apply(heatmaps, mu_t, sigma_t, 2), mask)


def variance_reg_loss(heatmaps, mu_t, sigma_t, mask=None):
    """Mean squared error between heatmap variances and sigma_t^2 (mu_t unused)."""
    return masked_average(_RegRows.apply(heatmaps, mu_t, sigma_t, 3), mask)


# ------------------------------------------------------------------ fused head (model.py:233-307)
class _HeadForward(Function):
    """softmax preact + dsnt in one pass: logits [B,J,H,W] -> (heatmaps, coords [B,J,2])."""

    @staticmethod
    def forward(ctx, logits):
        x = f32(logits).contiguous()
        h, w = x.shape[-2], x.shape[-1]
        rows = _rows(x, 2)
        hm = torch.empty_like(x)
        coords = torch.empty(*x.shape[:-2], 2, device=x.device, dtype=x.dtype)
        call('dsnt_head_fwd', ptr(x), ptr(hm), ptr(coords), rows, h, w)
        ctx.save_for_backward(hm)
        return hm, coords

    @staticmethod
    def backward(ctx, g_hm, g_coords):
        (hm,) = ctx.saved_tensors
        h, w = hm.shape[-2], hm.shape[-1]
        rows = _rows(hm, 2)
        g = None
        if g_coords is not None:
            g = torch.empty_like(hm)
            call('dsnt_expect_bwd', ptr(g_coords.contiguous()), ptr(g), rows, h, w)
        if g_hm is not None:
            g = g_hm.contiguous() if g is None else g + g_hm
        gx = torch.empty_like(hm)
        call('dsnt_preact_bwd', None, ptr(hm), ptr(g), ptr(gx), rows, h * w, 0, 0.0, 0.0)
        return gx


def head_forward(logits):
    return _HeadForward.apply(logits)


class _HeadLoss(Function):
    """euclidean_loss + reg_coeff * reg_loss for one stack, straight from the logits' saved heat-maps.

    Train step (gradients wanted, heat-maps of <= 4096 pixels): ONE kernel reads the heat-maps and leaves both the
    per-row loss terms and d loss / d logits for an upstream gradient of 1 (`dsnt_head_loss_grad`); backward only
    rescales it when the upstream gradient is not 1 — the heat-maps are read once, not twice (SURVEY.md Appendix A:
    4 HBM passes per stack).  Otherwise: loss rows now, `dsnt_head_bwd` in backward."""

    @staticmethod
    def forward(ctx, logits, hm, coords, target, mask, kind, sigma, reg_coeff, denom2):
        h, w = hm.shape[-2], hm.shape[-1]
        rows = _rows(hm, 2)
        t = f32(target).expand_as(coords).contiguous()
        m = None if mask is None else f32(mask).expand(coords.shape[:-1]).contiguous()
        dist = torch.empty(rows, device=hm.device, dtype=hm.dtype)
        reg = torch.empty(rows, device=hm.device, dtype=hm.dtype) if kind >= 0 else None
        if denom2 is None:
            denom2 = mask_denominator(m, rows, hm.device)
        ctx.g0 = None
        if ctx.needs_input_grad[0] and h * w <= 4096:
            g0 = torch.empty_like(hm)
            call('dsnt_head_loss_grad', ptr(hm), ptr(coords), ptr(t), ptr(m), ptr(denom2), ptr(dist), ptr(reg), ptr(g0),
                 rows, h, w, sigma, kind, reg_coeff)
            ctx.g0 = g0
        else:
            call('dsnt_head_loss_rows', ptr(hm), ptr(coords), ptr(t), ptr(dist), ptr(reg), rows, h, w,
                 sigma, kind)
        out = torch.empty(3, device=hm.device, dtype=hm.dtype)      # [loss, e2[0], e2[1]]
        e2 = out[1:]
        call('dsnt_head_loss_reduce', ptr(dist), ptr(reg), ptr(m), ptr(denom2), reg_coeff, ptr(out), ptr(e2), rows)
        ctx.save_for_backward(hm, coords, t, m, dist, e2)
        ctx.kind, ctx.sigma, ctx.reg_coeff = kind, sigma, reg_coeff
        return out[0]

    @staticmethod
    def backward(ctx, g):
        hm, coords, t, m, dist, e2 = ctx.saved_tensors
        h, w = hm.shape[-2], hm.shape[-1]
        rows = _rows(hm, 2)
        g1 = f32(g).contiguous().view(1)
        if ctx.g0 is not None:                 # first backward through the fused kernel's gradient: rescale in place
            g_logits, ctx.g0 = ctx.g0, None
            call('dsnt_scale_by_scalar', ptr(g_logits), ptr(g1), g_logits.numel())
            return g_logits, None, None, None, None, None, None, None, None
        g_dist = torch.empty(rows, device=hm.device, dtype=hm.dtype)
        call('dsnt_masked_avg_bwd', ptr(g1), ptr(m), ptr(e2), ptr(g_dist), rows)
        g_reg = (g_dist * ctx.reg_coeff) if ctx.kind >= 0 else None   # same mask and denominator
        g_logits = torch.empty_like(hm)
        call('dsnt_head_bwd', ptr(hm), ptr(coords), ptr(t), ptr(dist), ptr(g_dist), ptr(g_reg),
             ptr(g_logits), rows, h, w, ctx.sigma, ctx.kind)
        return g_logits, None, None, None, None, None, None, None, None


def mask_denominator(mask, rows, device):
    """{sum(mask), clamp(sum(mask), 1)} as two device floats (`masked_average`'s denominator, nn.py:81-94; mask None:
    `rows`): computed once per `forward_loss` call and shared by every stack's head_loss."""
    m = None if mask is None else f32(mask).contiguous()
    out = torch.empty(2, device=device, dtype=torch.float32)
    call('dsnt_mask_denom', ptr(m), ptr(out), rows)
    return out


def head_loss(logits, hm, coords, target, mask, reg, sigma, reg_coeff, denom2=None):
    kind = REG_KINDS.get(reg, -1)
    return _HeadLoss.apply(logits, hm, coords, target, mask, kind, float(sigma), float(reg_coeff), denom2)
