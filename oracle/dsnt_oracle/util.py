"""Oracle restatement of the heat-map ("gauss") target encoding and decoding (TEST INFRASTRUCTURE ONLY).

Follows `/root/reference/src/dsnt/util.py`: `draw_gaussian` :70-126, `encode_heatmaps` :129-147,
`get_preds` :150-169, `decode_heatmaps` :172-198 — written against today's PyTorch on the CPU, pinned by the
reference's own known answers (`/root/reference/tests/test_util.py:8-77`, see
tests/test_oracle_known_answers.py).  Differences that are deliberate: `encode_heatmaps` works on a copy (the
reference shifts and scales the caller's `coords` tensor in place, `util.py:133-136`); indices are
integer-divided explicitly (`idx / height` on a LongTensor floor-divided in PyTorch 0.3, `util.py:161`).
"""
import math

import torch


def draw_gaussian(img_tensor, x, y, sigma, normalize=False, clip_size=None):
    """Draw an (unnormalised unless `normalize`) Gaussian bump centred on pixel (int(x), int(y)) into a
    [H, W] or [1, H, W] tensor, in place (util.py:70-126)."""
    x = int(x)
    y = int(y)
    if img_tensor.dim() == 2:
        height, width = img_tensor.shape
    elif img_tensor.dim() == 3:
        n_chans, height, width = img_tensor.shape
        assert n_chans == 1, 'expected img_tensor to have one channel'
        img_tensor = img_tensor[0]
    else:
        raise Exception('expected img_tensor to have 2 or 3 dimensions')
    radius = max(width, height)
    if clip_size is not None:
        radius = clip_size / 2
    if radius < 0.5 or x <= -radius or y <= -radius or \
            x >= (width - 1) + radius or y >= (height - 1) + radius:
        return
    start_x = max(0, math.ceil(x - radius))
    end_x = min(width, int(x + radius + 1))
    start_y = max(0, math.ceil(y - radius))
    end_y = min(height, int(y + radius + 1))
    w = end_x - start_x
    h = end_y - start_y
    subimg = img_tensor[start_y:end_y, start_x:end_x]
    xs = torch.arange(start_x, end_x).to(img_tensor.dtype).view(1, w).expand_as(subimg)
    ys = torch.arange(start_y, end_y).to(img_tensor.dtype).view(h, 1).expand_as(subimg)
    k = -0.5 * (1 / sigma) ** 2
    subimg.copy_((xs - x) ** 2)
    subimg.add_((ys - y) ** 2)
    subimg.mul_(k)
    subimg.exp_()
    if normalize:
        val_sum = subimg.sum()
        if val_sum > 0:
            subimg.div_(val_sum)


def encode_heatmaps(coords, width, height, sigma=1):
    """Normalised coordinates [B, J, 2] -> float32 heat-maps [B, J, H, W]: an unnormalised 7x7-clipped
    Gaussian at the rounded pixel position (util.py:129-147; `round` is Python's round-half-even on the
    float32 pixel coordinate)."""
    c = coords.detach().to('cpu', torch.float32).clone()
    c.add_(1)
    c[:, :, 0].mul_(width / 2)
    c[:, :, 1].mul_(height / 2)
    c.add_(-0.5)
    batch_size, n_chans = c.size(0), c.size(1)
    target = torch.zeros(batch_size, n_chans, height, width, dtype=torch.float32)
    for i in range(batch_size):
        for j in range(n_chans):
            x = round(c[i, j, 0].item())
            y = round(c[i, j, 1].item())
            draw_gaussian(target[i, j], x, y, sigma, normalize=False, clip_size=7)
    return target


def get_preds(heatmaps):
    """Arg-max pixel (x, y) of every [H, W] map as floats; (0, 0) where the maximum is not positive
    (util.py:150-169; note y = idx // HEIGHT as in the reference: equal to idx // width for square maps)."""
    batch_size, n_chans, height, width = heatmaps.shape
    maxval, idx = torch.max(heatmaps.reshape(batch_size, n_chans, -1), 2)
    maxval = maxval.view(batch_size, n_chans, 1)
    idx = idx.view(batch_size, n_chans, 1)
    coords = idx.repeat(1, 1, 2)
    coords[:, :, 0] = coords[:, :, 0] % width
    coords[:, :, 1] = torch.div(coords[:, :, 1], height, rounding_mode='floor')
    coords = coords.float()
    pred_mask = maxval.gt(0).repeat(1, 1, 2).float()
    return coords * pred_mask


def decode_heatmaps(heatmaps, use_neighbours=True):
    """Heat-maps [B, J, H, W] -> normalised coordinates [B, J, 2] (util.py:172-198): arg-max pixel, moved a
    quarter pixel towards the larger neighbour on each axis when the pixel is interior."""
    heatmaps = heatmaps.detach().to('cpu')
    coords = get_preds(heatmaps)
    _, _, height, width = heatmaps.shape
    if use_neighbours:
        for i in range(coords.size(0)):
            for j in range(coords.size(1)):
                x = int(coords[i, j, 0])
                y = int(coords[i, j, 1])
                if 0 < x < width - 1 and 0 < y < height - 1:
                    hm = heatmaps[i, j]
                    coords[i, j, 0] += 0.25 * float(torch.sign(hm[y, x + 1] - hm[y, x - 1]))
                    coords[i, j, 1] += 0.25 * float(torch.sign(hm[y + 1, x] - hm[y - 1, x]))
    coords.add_(0.5)
    coords[:, :, 0].mul_(2 / width)
    coords[:, :, 1].mul_(2 / height)
    coords.add_(-1)
    return coords
