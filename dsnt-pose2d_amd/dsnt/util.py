"""Heat-map target encoding / decoding of the reference's `dsnt.util` on the HIP device.

`encode_heatmaps`, `decode_heatmaps` and `get_preds` keep the names, arguments and results of
`/root/reference/src/dsnt/util.py:129-198`; `heatmap_mse_loss` is the loss the `gauss` output strategy builds
from them (`/root/reference/src/dsnt/model.py:147-156, 247-258`) with the target bump evaluated inside the
kernel instead of being rendered on the CPU and copied over every step.  Tensors must be fp32 and resident on
the HIP device (no CPU fallback); results stay on the device — `compute_coords` moves them to the CPU as the
reference does.  Unlike the reference, `encode_heatmaps` does not shift and scale the caller's `coords` in
place (`util.py:133-136`).
"""
import torch
from torch.autograd import Function

from ._lib import ptr, f32, call


def encode_heatmaps(coords, width, height, sigma=1):
    '''Convert normalised coordinates [B, J, 2] into heatmaps [B, J, height, width] (util.py:129-147).'''
    c = f32(coords).detach().contiguous()
    out = torch.empty(*c.shape[:-1], height, width, device=c.device, dtype=torch.float32)
    call('dsnt_encode_heatmaps', ptr(c), ptr(out), c.numel() // 2, height, width, float(sigma))
    return out


def decode_heatmaps(heatmaps, use_neighbours=True):
    '''Convert heatmaps [B, J, H, W] into normalised coordinates [B, J, 2] (util.py:172-198).'''
    hm = f32(heatmaps).detach().contiguous()
    height, width = hm.size(-2), hm.size(-1)
    out = torch.empty(*hm.shape[:-2], 2, device=hm.device, dtype=torch.float32)
    rows = hm.numel() // (height * width)
    call('dsnt_decode_heatmaps', ptr(hm), ptr(out), rows, height, width, 1 if use_neighbours else 0)
    return out


def get_preds(heatmaps):
    '''Arg-max pixel coordinates (x, y) as floats, (0, 0) where the maximum is not positive (util.py:150-169).'''
    height, width = heatmaps.size(-2), heatmaps.size(-1)
    c = decode_heatmaps(heatmaps, use_neighbours=False)
    px = torch.round((c[..., 0] + 1) * (width / 2) - 0.5)
    py = torch.round((c[..., 1] + 1) * (height / 2) - 0.5)
    return torch.stack([px, py], -1)


class _HeatmapMSE(Function):
    """mean((hm - encode_heatmaps(target))^2) over every element, target never materialised."""

    @staticmethod
    def forward(ctx, hm, target, sigma):
        x = f32(hm).contiguous()
        t = f32(target).detach().contiguous()
        height, width = x.size(-2), x.size(-1)
        rows = x.numel() // (height * width)
        if t.numel() != rows * 2:
            raise RuntimeError('heatmap_mse_loss: target %s does not match heat-maps %s' % (tuple(t.shape), tuple(x.shape)))
        per_row = torch.empty(rows, device=x.device, dtype=torch.float32)
        call('dsnt_heatmap_mse_fwd', ptr(x), ptr(t), ptr(per_row), rows, height, width, float(sigma))
        ctx.save_for_backward(x, t)
        ctx.sigma = float(sigma)
        return per_row.sum() / x.numel()

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        height, width = x.size(-2), x.size(-1)
        rows = x.numel() // (height * width)
        gs = g.to(torch.float32).reshape(1).contiguous()
        dx = torch.empty_like(x)
        call('dsnt_heatmap_mse_bwd', ptr(x), ptr(t), ptr(gs), ptr(dx), rows, height, width, ctx.sigma)
        return dx, None, None


def heatmap_mse_loss(heatmaps, target_coords, sigma=1):
    """`nn.functional.mse_loss(heatmaps, encode_heatmaps(target_coords, W, H, sigma))` (model.py:150-156)."""
    return _HeatmapMSE.apply(heatmaps, target_coords, sigma)
