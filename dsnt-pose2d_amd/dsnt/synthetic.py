"""Deterministic synthetic weights and inputs (no torch RNG, no files).

There is no dataset or checkpoint on the GPU box, so parity tests, the golden
generator and `bench.py` all draw weights and batches from here.  Values come
from numpy's PCG64 keyed by (seed, crc32(name)), which is stable across numpy
versions and platforms, so the CPU oracle, the reference (in the build
container) and the HIP path see bit-identical fp32 inputs.

Image statistics: the reference subtracts the MPII per-channel mean for
hourglass models (`/root/reference/src/dsnt/model.py:227`,
`src/dsnt/data.py:47-55`); the constants live in the absent `torchdata`
package, so a fixed stand-in mean is declared here.
"""
import zlib

import numpy as np
import torch

# Stand-in for torchdata.mpii.MPII_Image_Mean (absent dependency); declared, not measured.
IMAGE_MEAN = (0.44, 0.44, 0.40)


def _rng(seed, name):
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def fill_state_dict(model, seed=0):
    """Overwrite every parameter (not the running stats) of `model` in place.

    Conv/linear weights ~ N(0, 2/fan_in) (He), biases ~ U(-0.1, 0.1), BN gamma ~
    U(0.5, 1.5), BN beta ~ U(-0.2, 0.2).  Keys are the reference's state_dict
    names, so the same call initialises reference, oracle and product models
    identically.
    """
    sd = model.state_dict()
    new = {}
    for key, t in sd.items():
        if key.endswith('running_mean') or key.endswith('running_var') or \
                key.endswith('num_batches_tracked'):
            continue
        r = _rng(seed, key)
        shape = tuple(t.shape)
        if t.dim() >= 2:
            fan_in = int(np.prod(shape[1:]))
            v = r.standard_normal(shape, dtype=np.float64) * np.sqrt(2.0 / fan_in)
        elif key.endswith('weight'):      # BN gamma
            v = r.uniform(0.5, 1.5, shape)
        elif '.bn' in key or key.startswith('bn') or _is_bn_bias(sd, key):
            v = r.uniform(-0.2, 0.2, shape)
        else:                             # conv / linear bias
            v = r.uniform(-0.1, 0.1, shape)
        new[key] = torch.from_numpy(np.asarray(v, dtype=np.float32)).to(t.dtype)
    model.load_state_dict(new, strict=False)
    return model


def _is_bn_bias(sd, key):
    return key.endswith('bias') and (key[:-4] + 'running_mean') in sd


def batch(batch_size, size=256, n_joints=16, seed=1, mask_p=1.0, dtype=torch.float32):
    """Synthetic crops U(0,1) minus IMAGE_MEAN, targets U(-1,1), Bernoulli(mask_p) mask."""
    r = _rng(seed, 'batch')
    img = r.random((batch_size, 3, size, size), dtype=np.float32)
    img -= np.asarray(IMAGE_MEAN, dtype=np.float32).reshape(1, 3, 1, 1)
    target = r.uniform(-1, 1, (batch_size, n_joints, 2)).astype(np.float32)
    mask = (r.random((batch_size, n_joints)) < mask_p).astype(np.float32)
    return (torch.from_numpy(img).to(dtype), torch.from_numpy(target).to(dtype),
            torch.from_numpy(mask).to(dtype))


def tensor(name, shape, seed=0, scale=1.0, dtype=torch.float32, kind='normal'):
    """A named deterministic tensor (N(0, scale^2) or U(-scale, scale))."""
    r = _rng(seed, name)
    if kind == 'normal':
        v = r.standard_normal(shape, dtype=np.float64) * scale
    else:
        v = r.uniform(-scale, scale, shape)
    return torch.from_numpy(np.asarray(v, dtype=np.float32)).to(dtype)


def pckh_inputs(batch_size, n_joints=16, seed=2):
    """Synthetic head lengths U(40,120) px and an affine back-projection (m, b)."""
    r = _rng(seed, 'pckh')
    head = torch.from_numpy(r.uniform(40, 120, (batch_size,)))
    scale = r.uniform(80, 160, (batch_size,))
    m = torch.zeros(batch_size, 2, 2, dtype=torch.float64)
    m[:, 0, 0] = torch.from_numpy(scale)
    m[:, 1, 1] = torch.from_numpy(scale)
    b = torch.from_numpy(r.uniform(200, 600, (batch_size, 1, 2)))
    return head, m, b
