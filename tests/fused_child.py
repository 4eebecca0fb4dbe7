"""Child process of tests/test_fused_inmodel_gpu.py (not collected by pytest: no test_ prefix).

The library reads its row thresholds ONCE per process (csrc/bwd1.hip `dsnt_bwd1_plan`, csrc/fwd1.hip), so forcing the
round-4 one-pass kernels onto the small convolutions of the oracle-sized models needs a process of its own.  The parent
sets DSNT_X_BWD1_MIN_ROWS=0, DSNT_X_FWD1_MIN_ROWS=0, DSNT_BF16X6_MIN_ROWS=0 and the fp16x3 path; this script then runs
the SAME golden / oracle checks as tests/test_model_gpu.py (calling its functions) and looks at the tapes of the models
those functions built: which launches ran, with which geometry, and whether a BatchNorm backward was folded in.

Reference for what is being checked: /root/reference/src/dsnt/hourglass.py:30-50 (the Bottleneck whose 1x1 convolutions
these kernels run), golden vectors from tests/golden/make_golden.py.

usage: python tests/fused_child.py <case> <result.json>
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [HERE, os.path.join(ROOT, 'dsnt-pose2d_amd'), os.path.join(ROOT, 'oracle')]

# (cout, cin) of every row of b1_cfgs (csrc/bwd1.hip)
B1_ROWS = [(128, 256), (256, 128), (128, 128), (64, 64), (128, 64), (256, 256)]


def tape_report(models):
    """Launch census over the train-mode programs of `models`: bwd1 launches by (Cout, Cin, rows), how many carried a
    folded BatchNorm backward (dsnt_bn_bwd_apply != NULL), fwd1 launches by shape; `c3`: the launches of csrc/conv3s.hip by
    (list f / b, Cout, Cin, rows, MODE, form) — MODE 0 / 1 plain, 3 BatchNorm-backward epilogue, 4 folded BatchNorm backward
    (`fold3`); form bit 0 column split, 1 the 16x16x32 form, 2 8 x 16 patches; `wgrad_after_fold3`: folded launches whose
    dy_out a later weight-gradient launch names."""
    rep = {'bwd1': {}, 'bwd1_folded': {}, 'bwd1_raw': {}, 'fwd1': {}, 'names_bwd': {}, 'c3': {}, 'wgrad_after_fold3': 0}

    def c3(lib, e, g, mode, which):
        # the library itself says which form of csrc/conv3s.hip this launch takes in this process (dsnt_conv_fwd_stream_form)
        form = lib.dsnt_conv_fwd_stream_form(g, mode)
        assert form >= 0, (e[2], mode)
        go = g._obj
        k = '%s,%d,%d,%d,%d,%d' % (which, go.Cout, go.Cin, go.N * go.Ho * go.Wo, mode, form)
        rep['c3'][k] = rep['c3'].get(k, 0) + 1
    for m in models:
        root = m.hg if hasattr(m, 'hg') else m
        for prog in root._runner().programs.values():
            if not prog.record:
                continue
            tape = prog.tape
            lib = tape.lib
            folded = {id(e) for e, u in tape.f16_uses if u.get('kind') == 'bwd1' and 'g_apply' in u}
            dy_out = {}
            for e in tape.bwd:
                if e[0] is None:
                    continue
                rep['names_bwd'][e[2]] = rep['names_bwd'].get(e[2], 0) + 1
                if e[2] == 'dsnt_conv_fwd_f16x3_stream':          # a 3x3 data gradient: ..., res1 [10], res2, part, g, bnb, tail
                    c3(lib, e, e[1][-3], 3 if e[1][-2] is not None else (1 if e[1][10] is not None else 0), 'b')
                elif e[2] == 'dsnt_conv_dgrad_f16x3_stream_apply':   # dz, ap, dy_out [2], ..., g, bnb, tail
                    c3(lib, e, e[1][-3], 4, 'b')
                    dy_out[e[1][2]] = False          # (device pointers are plain ints on the tape)
                elif dy_out:
                    # the dL/dy a folded launch wrote has to be what a weight gradient reads next (hourglass.py:36-40: conv2's)
                    for v in e[1]:
                        if isinstance(v, int) and v in dy_out and 'wgrad' in e[2] and not dy_out[v]:
                            dy_out[v] = True
                            rep['wgrad_after_fold3'] += 1
                if e[2] == 'dsnt_conv1x1_bwd_f16x3':
                    g = e[1][-1]._obj
                    k = '%d,%d,%d' % (g.Cout, g.Cin, g.N * g.H * g.W)
                    rep['bwd1'][k] = rep['bwd1'].get(k, 0) + 1
                    if id(e) in folded:
                        rep['bwd1_folded'][k] = rep['bwd1_folded'].get(k, 0) + 1
                    if not e[1][0]._obj.scale:
                        rep['bwd1_raw'][k] = rep['bwd1_raw'].get(k, 0) + 1
            for e in tape.fwd:
                if e[0] is not None and e[2] == 'dsnt_conv_fwd_f16x3_stream':
                    c3(lib, e, e[1][-3], 1 if e[1][10] is not None else 0, 'f')
                if e[0] is not None and e[2] == 'dsnt_conv1x1_fwd_f16x3':
                    g = e[1][-2]._obj
                    k = '%d,%d,%d' % (g.Cout, g.Cin, g.N * g.H * g.W)
                    rep['fwd1'][k] = rep['fwd1'].get(k, 0) + 1
    return rep


def main(case, out_path):
    import dsnt.model as dmodel
    built = []
    real = dmodel.build_mpii_pose_model

    def recording(*a, **k):
        m = real(*a, **k)
        built.append(m)
        return m
    dmodel.build_mpii_pose_model = recording
    import test_model_gpu as tm

    if case.startswith('golden:'):
        tag = case.split(':')[1]
        base, size, reg = {'hg2_128': ('hg2', 128, 'js'), 'hg2_256': ('hg2', 256, 'js'), 'hg8_128': ('hg8', 128, 'js'),
                           'hg1_128': ('hg1', 128, 'none')}[tag]
        tm.test_end_to_end_vs_golden(base, size, reg, tag, 'f16x3')
    elif case == 'hg2_grads:smooth':
        tm.test_hg2_every_gradient_vs_oracle(True)
    elif case == 'hg2_grads:relu':
        tm.test_hg2_every_gradient_vs_oracle(False)
    elif case == 'hg8_grads:smooth':
        tm.test_hg8_every_gradient_vs_oracle_on_the_smooth_network()
    else:
        raise SystemExit('unknown case ' + case)
    rep = tape_report(built)
    rep['case'] = case
    with open(out_path, 'w') as f:
        json.dump(rep, f)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
