"""Sensitivity of the in-model tests of the folded 3x3 data gradient (tests/test_fused_inmodel_gpu.py, variant `prod`): the same run with
the folded BatchNorm's vectors mis-wired must FAIL.  python3 tools/mutate_fold3.py none|swap|scale under the variant's environment."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'dsnt-pose2d_amd'), os.path.join(ROOT, 'oracle')]
import dsnt.engine as E
real = E.BnBwdApply
which = sys.argv[1]
class Mutated(real):
    """The struct of the folded BatchNorm backward with two of its vectors exchanged (which = swap / scale), or as it is (none)."""
    def __init__(self, y, scale, mean, invstd, coef):
        if which == 'swap':          # mean <-> invstd of the folded BatchNorm
            super().__init__(y, scale, invstd, mean, coef)
        elif which == 'scale':       # the wrong scale vector (here: the BatchNorm's mean)
            super().__init__(y, mean, mean, invstd, coef)
        else:
            super().__init__(y, scale, mean, invstd, coef)
E.BnBwdApply = Mutated
import fused_child
try:
    fused_child.main('hg2_grads:smooth', '/tmp/mut_%s.json' % which)
    print('MUTATION', which, 'NOT DETECTED')
except AssertionError as e:
    print('MUTATION', which, 'detected:', str(e)[:200].replace('\n', ' '))
