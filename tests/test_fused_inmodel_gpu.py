"""The round-4 one-pass kernels (csrc/bwd1.hip, csrc/fwd1.hip) under the oracle INSIDE a model.

In production `dsnt_conv1x1_bwd_f16x3` takes a convolution from 16384 output rows and `dsnt_conv1x1_fwd_f16x3` from 4096,
which the oracle-sized models (batch 2-4, 128 / 256 px) reach at the stem only — so the engine's wiring of those launches
(which BatchNorm's coefficients, which shared / continued gradient, which slab feeds which parameter) would be checked by
kernel-level tests with the TEST's own wiring alone.  Here every threshold is forced to zero in a child process
(tests/fused_child.py: the library reads them once per process) and the golden vectors of the reference
(/root/reference/src/dsnt/hourglass.py:30-50 through tests/golden/make_golden.py) plus the oracle's every-gradient checks
are run again, calling the very functions of tests/test_model_gpu.py; the child reports the launch census of the tapes,
which is asserted here: >= 13 one-pass backwards carried a folded BatchNorm apply, and every row of `b1_cfgs` ran."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
B1_ROWS = [(128, 256), (256, 128), (128, 128), (64, 64), (128, 64), (256, 256)]     # csrc/bwd1.hip b1_cfgs


def _child(tmp_path, case):
    env = dict(os.environ)
    for k in ('DSNT_OFF', 'DSNT_X', 'DSNT_DEBUG_NO_RELU'):
        env.pop(k, None)
    env.update(DSNT_MFMA='bf16x6', DSNT_SPLIT='f16x3', DSNT_BF16X6_MIN_ROWS='0', DSNT_X_BWD1_MIN_ROWS='0',
               DSNT_X_FWD1_MIN_ROWS='0')
    out = str(tmp_path / (case.replace(':', '_') + '.json'))
    r = subprocess.run([sys.executable, os.path.join(HERE, 'fused_child.py'), case, out], env=env, timeout=900,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    with open(out) as f:
        return json.load(f)


def _pairs(d):
    return {tuple(int(v) for v in k.split(',')[:2]) for k in d}


def _check_census(rep, stacks, rows_main):
    n_bwd1 = sum(rep['bwd1'].values())
    n_folded = sum(rep['bwd1_folded'].values())
    # every 1x1 convolution with a row count the kernel takes (a multiple of 32) is on it: 2-3 per Bottleneck down to 4x4 ...
    assert n_bwd1 >= 6 + stacks * 20, rep['bwd1']
    # ... and bn2's backward rides in conv1's launch wherever conv1 runs on it (13 per hg2 step in production)
    assert n_folded >= 13, rep['bwd1_folded']
    assert _pairs(rep['bwd1']) == set(B1_ROWS), rep['bwd1']
    assert _pairs(rep['bwd1_folded']) >= {(128, 256), (128, 128), (64, 64)}, rep['bwd1_folded']
    assert _pairs(rep['bwd1_raw']) >= {(256, 128), (128, 64), (256, 256)}, rep['bwd1_raw']     # projections, `fc`
    # the production shape classes (64^2 / 32^2 level of the stacks: rows_main and a quarter of it) are among them
    for cout, cin in ((128, 256), (256, 128)):
        assert '%d,%d,%d' % (cout, cin, rows_main) in rep['bwd1'], rep['bwd1']
        assert '%d,%d,%d' % (cout, cin, rows_main // 4) in rep['bwd1'], rep['bwd1']
    assert '128,256,%d' % rows_main in rep['bwd1_folded']
    assert sum(rep['fwd1'].values()) >= 6 + stacks * 20, rep['fwd1']


@pytest.mark.parametrize('tag,stacks,rows_main', [('hg2_128', 2, 2 * 32 * 32), ('hg2_256', 2, 2 * 64 * 64),
                                                   ('hg8_128', 8, 2 * 32 * 32)])
def test_goldens_with_every_1x1_convolution_on_the_one_pass_kernels(tmp_path, tag, stacks, rows_main):
    rep = _child(tmp_path, 'golden:' + tag)
    _check_census(rep, stacks, rows_main)


@pytest.mark.parametrize('kind', ['smooth', 'relu'])
def test_hg2_every_gradient_vs_oracle_on_the_one_pass_kernels(tmp_path, kind):
    """All 396 gradients of hg2 (batch 4, 128 px) against the oracle — smooth network: 1e-3 per parameter, cosine 1 - 1e-7."""
    rep = _child(tmp_path, 'hg2_grads:' + kind)
    _check_census(rep, 2, 4 * 32 * 32)
