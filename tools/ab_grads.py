"""Per-parameter gradient distance between two switch settings of one full-size train step (the comparison of
tests/test_fallback_gpu.py, printed as a distribution instead of asserted).

usage: python tools/ab_grads.py <base> <batch> <smooth 0|1> <variant> [<variant> ...]
       variant = default | off_r3 | off_r4 | copies | OFF:<DSNT_OFF value> | X:<DSNT_X value>
The first variant is the reference of the comparison."""
import os
import sys
import tempfile
import pathlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'dsnt-pose2d_amd')]
import torch  # noqa: E402
import test_fallback_gpu as tf  # noqa: E402

NAMED = {'default': (None, None), 'fold3': ('fold3', None), 'bwd_r4': ('bwd1+stem4w', None), 'fwd_r4': ('fwd1+stem4', None), 'off_r3': ('conv3s+gemm1+wgrad3+wgrad1', None), 'off_r4': ('bwd1+fwd1+stem4+stem4w', None),
         'copies': (None, 'share_grads=0,defer_res=0')}


def main():
    base, batch, smooth = sys.argv[1], int(sys.argv[2]), sys.argv[3] == '1'
    runs = []
    d = pathlib.Path(tempfile.mkdtemp())
    for i, v in enumerate(sys.argv[4:]):
        if v.startswith('OFF:'):
            off, x = v[4:], None
        elif v.startswith('X:'):
            off, x = None, v[2:]
        else:
            off, x = NAMED[v]
        runs.append((v, tf._run(d, 'r%d' % i, off, base, batch, 256, x=x, smooth=smooth)))
    ref_name, ref = runs[0]
    floor = 1e-3 * max(v.double().norm().item() for v in ref['grads'].values())
    for name, r in runs[1:]:
        errs = sorted(((r['grads'][n].double() - v.double()).norm().item() / max(v.double().norm().item(), floor), n)
                      for n, v in ref['grads'].items())
        fn = torch.cat([v.reshape(-1) for v in r['grads'].values()]).double()
        fo = torch.cat([v.reshape(-1) for v in ref['grads'].values()]).double()
        cos = (fn @ fo / (fn.norm() * fo.norm())).item()
        print('%s vs %s (%s b%d %s): loss %.9g / %.9g, coords %.2e, 1-cos %.2e' % (
            name, ref_name, base, batch, 'smooth' if smooth else 'relu', r['loss'], ref['loss'],
            (r['coords'] - ref['coords']).abs().max().item(), 1 - cos))
        q = [errs[int(f * (len(errs) - 1))][0] for f in (0.5, 0.9, 0.99, 1.0)]
        print('  rel-L2 per parameter: median %.2e  p90 %.2e  p99 %.2e  max %.2e' % tuple(q))
        for e, n in errs[-8:]:
            print('    %.2e  %s' % (e, n))


if __name__ == '__main__':
    main()
