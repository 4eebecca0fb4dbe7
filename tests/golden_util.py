"""Helpers to compare tensors with the summaries stored in tests/golden/*.npz."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + '.npz'))


def check_summary(g, prefix, t, atol, rtol=0.0):
    """Sampled entries, sum and L2 norm of `t` against the golden summary `prefix`."""
    a = t.detach().double().cpu().reshape(-1).numpy()
    idx, val = g[prefix + '.idx'], g[prefix + '.val']
    scale = max(1.0, float(np.abs(val).max()))
    err = np.abs(a[idx] - val).max()
    assert err <= atol * scale + rtol * scale, (prefix, 'samples', err)
    l2 = float(g[prefix + '.l2'])
    got = float(np.sqrt((a * a).sum()))
    assert abs(got - l2) <= (atol * 50 + rtol) * max(1.0, l2), (prefix, 'l2', got, l2)
    return err
