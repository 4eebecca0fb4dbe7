"""The round-4 one-pass kernels (csrc/bwd1.hip, csrc/fwd1.hip) under the oracle INSIDE a model.

In production `dsnt_conv1x1_bwd_f16x3` takes a convolution from 16384 output rows and `dsnt_conv1x1_fwd_f16x3` from 4096,
which the oracle-sized models (batch 2-4, 128 / 256 px) reach at the stem only — so the engine's wiring of those launches
(which BatchNorm's coefficients, which shared / continued gradient, which slab feeds which parameter) would be checked by
kernel-level tests with the TEST's own wiring alone.  Here every threshold is forced to zero in a child process
(tests/fused_child.py: the library reads them once per process) and the golden vectors of the reference
(/root/reference/src/dsnt/hourglass.py:30-50 through tests/golden/make_golden.py) plus the oracle's every-gradient checks
are run again, calling the very functions of tests/test_model_gpu.py; the child reports the launch census of the tapes,
which is asserted here: >= 13 one-pass backwards carried a folded BatchNorm apply, and every row of `b1_cfgs` ran.

Round 6: the same for the 3x3 kernel of csrc/conv3s.hip (/root/reference/src/dsnt/hourglass.py:22-23,36-40 — conv2 of the
Bottleneck and the BatchNorm behind it).  Which FORM of it a launch takes is decided by row thresholds too: in production
(batch 32, 131072 rows at 64 x 64) the forward runs on the 16x16x32 form with the LDS-DMA weight ring (`MF`), the data gradient
with bn3's backward folded into its operand load (MODE 4, `fold3`, DSNT_X_FOLD3_ROWS = 16384) on the 32x32x16 form, and its
dL/dy by-product feeds the halo weight-gradient kernel directly; an oracle-sized model (<= 8192 rows) takes the column-split
form (`SP`), MODE 3 + a separate apply, and the grouped weight gradient instead.  VARIANTS below force each production form onto
the small models; the census (the library itself reports the form: dsnt_conv_fwd_stream_form) is asserted per variant."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
B1_ROWS = [(128, 256), (256, 128), (128, 128), (64, 64), (128, 64), (256, 256)]     # csrc/bwd1.hip b1_cfgs


# conv3s forms (bit 0 column split, 1 the 16x16x32 form, 2 8 x 16 patches)
PLAIN, SP, MF, P16 = 0, 1, 2, 4
# what each variant forces (on top of the one-pass 1x1 thresholds) and which (list, MODE, form) cells its census must hold
VARIANTS = {
    # the round-5 set: every 1x1 convolution on bwd1 / fwd1; the 3x3 launches of these sizes take the column split by themselves
    'onepass': (dict(), [('f', 0, SP), ('b', 3, SP)]),
    # production's forms: forward on MF, every eligible data gradient MODE 4 on the 32x32x16 form (and on 8 x 16 patches at
    # 16 x 16), its dL/dy read by a DIRECT weight-gradient launch on the weight-gradient lane (no grouped launches)
    'prod': (dict(DSNT_X_FOLD3_ROWS='0', DSNT_X_C3_SPLIT_TILES='0', DSNT_X_GROUP_ROWS='0', DSNT_X_WGRAD_LANE_ROWS='1'),
             [('f', 0, MF), ('b', 4, PLAIN), ('b', 4, P16)]),
    # ... and production's MODE 3 (a separate apply in front of it), no column split
    'prod3': (dict(DSNT_X_C3_SPLIT_TILES='0', DSNT_X_GROUP_ROWS='0', DSNT_X_WGRAD_LANE_ROWS='1', DSNT_OFF='fold3'),
              [('f', 0, MF), ('b', 3, PLAIN), ('b', 3, P16)]),
    # every mode on the 16x16x32 form
    'mf': (dict(DSNT_X_FOLD3_ROWS='0', DSNT_X_C3_SPLIT_TILES='0', DSNT_X_C3_MF16='7'), [('f', 0, MF), ('b', 4, MF)]),
    'mf3': (dict(DSNT_X_C3_SPLIT_TILES='0', DSNT_X_C3_MF16='7', DSNT_OFF='fold3'), [('f', 0, MF), ('b', 3, MF)]),
    # the folded data gradient beside the column split of the other launches
    'fold': (dict(DSNT_X_FOLD3_ROWS='0'), [('f', 0, SP), ('b', 4, PLAIN)]),
}


def _child(tmp_path, case, variant='onepass'):
    env = dict(os.environ)
    for k in ('DSNT_OFF', 'DSNT_X', 'DSNT_DEBUG_NO_RELU'):
        env.pop(k, None)
    for k in [k for k in env if k.startswith('DSNT_X_')]:
        env.pop(k)
    env.update(DSNT_MFMA='bf16x6', DSNT_SPLIT='f16x3', DSNT_BF16X6_MIN_ROWS='0', DSNT_X_BWD1_MIN_ROWS='0',
               DSNT_X_FWD1_MIN_ROWS='0')
    env.update(VARIANTS[variant][0])
    out = str(tmp_path / (case.replace(':', '_') + '_' + variant + '.json'))
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


def _c3_cells(rep, cout=128, cin=128):
    """{(list, MODE, form): launches} of the Cout x Cin launches of csrc/conv3s.hip in the child's tapes."""
    cells = {}
    for k, n in rep['c3'].items():
        which, co, ci, rows, mode, form = k.split(',')
        if int(co) == cout and int(ci) == cin:
            key = (which, 0 if mode == '1' else int(mode), int(form))
            cells[key] = cells.get(key, 0) + n
    return cells


def _check_c3(rep, variant, stacks):
    """The 128 -> 128 3x3 convolutions of the Bottlenecks (hourglass.py:22-23) ran on the forms the variant is there for."""
    cells = _c3_cells(rep)
    for cell in VARIANTS[variant][1]:
        if cell[2] == P16 and not any(k[2] & P16 for k in cells):
            continue                        # (this model has no 16-pixel-wide level the kernel takes)
        assert cells.get(cell, 0) >= 1, (variant, cell, cells)
    env = VARIANTS[variant][0]
    n4 = sum(n for (which, mode, form), n in cells.items() if mode == 4)
    n3 = sum(n for (which, mode, form), n in cells.items() if mode == 3)
    if env.get('DSNT_X_FOLD3_ROWS') == '0':
        # conv2 of every Bottleneck the kernel takes: 2 at the first level + 3 at the second of each stack (+ the stem's)
        assert n4 >= 5 * stacks and n3 == 0, cells
        assert rep['names_bwd'].get('dsnt_conv_dgrad_f16x3_stream_apply', 0) >= n4
        if env.get('DSNT_X_GROUP_ROWS') == '0':
            # ... each one's dL/dy by-product is the dY operand of a direct weight-gradient launch behind it
            assert rep['wgrad_after_fold3'] >= n4, (rep['wgrad_after_fold3'], n4)
    elif 'fold3' in env.get('DSNT_OFF', ''):
        assert n3 >= 5 * stacks and n4 == 0, cells
    if env.get('DSNT_X_C3_SPLIT_TILES') == '0':
        assert not any(form & SP for (which, mode, form) in cells), cells
    if env.get('DSNT_X_C3_MF16') == '7':
        # everything 32 pixels wide is on the 16x16x32 form
        assert not any(form == PLAIN for (which, mode, form) in cells), cells


GOLDENS = {'hg2_128': (2, 2 * 32 * 32), 'hg2_256': (2, 2 * 64 * 64), 'hg8_128': (8, 2 * 32 * 32)}


@pytest.mark.parametrize('tag,variant', [('hg2_128', 'onepass'), ('hg2_256', 'onepass'), ('hg8_128', 'onepass'),
                                         ('hg2_128', 'prod'), ('hg2_256', 'prod'), ('hg8_128', 'prod'),
                                         ('hg2_256', 'prod3'), ('hg2_256', 'mf'), ('hg8_128', 'mf'), ('hg2_256', 'mf3'),
                                         ('hg2_256', 'fold')])
def test_goldens_with_the_production_kernels_forced_onto_the_small_models(tmp_path, tag, variant):
    """The reference's golden vectors (coords of every stack 1e-4, loss, heat-maps, running statistics, gradient norms) with every
    1x1 convolution on the one-pass kernels and the 3x3 convolutions on the forms `variant` names."""
    stacks, rows_main = GOLDENS[tag]
    rep = _child(tmp_path, 'golden:' + tag, variant)
    _check_census(rep, stacks, rows_main)
    _check_c3(rep, variant, stacks)


@pytest.mark.parametrize('kind,variant', [('smooth', 'onepass'), ('relu', 'onepass'), ('smooth', 'prod'), ('relu', 'prod'),
                                          ('smooth', 'prod3'), ('smooth', 'mf'), ('smooth', 'mf3')])
def test_hg2_every_gradient_vs_oracle_on_the_production_kernels(tmp_path, kind, variant):
    """All 396 gradients of hg2 (batch 4, 128 px) against the oracle — smooth network: 1e-3 per parameter, cosine 1 - 1e-7.
    A wrong coefficient vector, a stale dL/dy handed to the weight gradient or a mis-wired BatchNorm in a folded launch moves
    one layer's gradients by O(1) and fails this by orders of magnitude."""
    rep = _child(tmp_path, 'hg2_grads:' + kind, variant)
    _check_census(rep, 2, 4 * 32 * 32)
    _check_c3(rep, variant, 2)


@pytest.mark.parametrize('variant', ['prod'])        # ('mf' passes too; 25 s of GPU time per variant)
def test_hg8_every_gradient_vs_oracle_on_the_production_kernels(tmp_path, variant):
    """hg8's 1464 gradients (batch 2, 128 px, smooth network) inside the oracle's own fp32-vs-fp64 envelope
    (tests/test_model_gpu.py::test_hg8_every_gradient_vs_oracle_on_the_smooth_network) with the production forms forced."""
    rep = _child(tmp_path, 'hg8_grads:smooth', variant)
    _check_census(rep, 8, 2 * 32 * 32)
    _check_c3(rep, variant, 8)
