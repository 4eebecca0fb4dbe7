"""The launch lists of the full-size hg2 train step, traced on the CPU (no kernel is launched: only the engine's list building
runs, with the device check of `_lib.ptr` bypassed): the SCHEDULE the measurements in DESIGN.md §3 "round 2" rely on —
which lane carries what, where lanes wait for each other, which launches were fused away."""
import ctypes as C
import collections

import pytest
import torch


@pytest.fixture(scope='module')
def tape(monkeypatch_module):
    from dsnt import _lib
    import dsnt.engine as E
    # (every module that binds `ptr` by name is imported BEFORE the patch, or it would keep the permissive one for good)
    from dsnt.model import build_mpii_pose_model
    from dsnt.hourglass import Arena, Program
    monkeypatch_module.setattr(_lib, 'ptr', lambda t: C.c_void_p(t.data_ptr()) if t is not None else None)
    monkeypatch_module.setattr(E._lib, 'ptr', _lib.ptr)
    m = build_mpii_pose_model(base='hg2', output_strat='dsnt', reg='js')
    m.train()
    ar = Arena(m.hg, torch.device('cpu'))
    return Program(m.hg, ar, (32, 3, 256, 256), True, False).tape


@pytest.fixture(scope='module')
def monkeypatch_module():
    mp = pytest.MonkeyPatch()
    yield mp
    mp.undo()


def _launches(lst):
    return [(name, lane, args) for fn, args, name, lane in lst if fn is not None]


def test_forward_list_preparation_runs_beside_the_stem(tape):
    fwd = tape.fwd
    names = [(e[2], e[3]) for e in fwd]
    assert names[0] == ('dsnt_fill_zero', 0)                      # the forward's bound slots, before any producer
    prep = [n for n, lane in names if lane == 1 and n in ('dsnt_split_bf16x3', 'dsnt_f16_prep_weights',
                                                            'dsnt_f16_prep_bn_bounds', 'dsnt_conv_pack_dgrad_all')]
    assert prep == ['dsnt_split_bf16x3', 'dsnt_f16_prep_weights', 'dsnt_f16_prep_bn_bounds', 'dsnt_conv_pack_dgrad_all',
                    'dsnt_f16_prep_weights']
    # the main lane runs the stem (its own one-launch weight preparation, the space-to-depth image, the convolution, the
    # BatchNorm + ReLU with statistics) BEFORE it waits for the side lane's preparation
    # (round 5: the main lane waits for the FORWARD half of the preparation only — relayed through the idle lane 3, which took the
    # dependency right behind that half; the backward-only half, the re-packed / split data-gradient weights, stays on lane 1)
    relay = next(i for i, e in enumerate(fwd) if e[2] == 'sync' and e[1][0] == 1 and e[1][1] == 3)
    side_before = [e[2] for e in fwd[:relay] if e[3] == 1 and e[0] is not None]
    side_after = [e[2] for e in fwd[relay:] if e[3] == 1 and e[0] is not None][:2]
    assert side_before == ['dsnt_split_bf16x3', 'dsnt_f16_prep_weights', 'dsnt_f16_prep_bn_bounds']
    assert side_after == ['dsnt_conv_pack_dgrad_all', 'dsnt_f16_prep_weights']
    first_sync = next(i for i, e in enumerate(fwd) if e[2] == 'sync' and e[1][0] == 3 and e[1][1] == 0)
    assert relay < first_sync
    before = [e[2] for e in fwd[:first_sync] if e[3] == 0 and e[0] is not None]
    # (the stem convolution itself: the halo kernel of csrc/stem4.hip, one statistics row per workgroup)
    assert before[:5] == ['dsnt_fill_zero', 'dsnt_s2d_input', 'dsnt_s2d_weights_prep', 'dsnt_stem4_fwd_f16x3', 'dsnt_bn_finalize']
    assert 'dsnt_bn_act_fwd_stats' in before and before.count('dsnt_stem4_fwd_f16x3') == 1 and 'dsnt_conv_fwd_f16x3_ex' not in before
    # ... and that wait comes before the first launch that reads ANY prepared plane or bound, the stream-layout 3x3 kernel included
    readers = ('dsnt_conv_fwd_f16x3_ex', 'dsnt_conv_fwd_f16x3_stream', 'dsnt_conv_fwd_bf16x6_ex', 'dsnt_conv_fwd_bf16x6')
    first_reader = next(i for i, e in enumerate(fwd) if e[2] in readers and id(e) not in tape._prep_exempt)
    assert first_sync < first_reader
    first_stream = next(i for i, e in enumerate(fwd) if e[2] == 'dsnt_conv_fwd_f16x3_stream')
    assert first_sync < first_stream and 'dsnt_conv_fwd_f16x3_stream' in tape._PREP_CONSUMERS
    assert 'dsnt_nchw_to_nhwc' not in [e[2] for e in fwd]         # the NHWC copy of the image is gone
    assert {e[3] for e in fwd if e[0] is not None} == {0, 1, 3}     # two side lanes for the skip branches


def test_backward_list_lanes_and_fusions(tape):
    bwd = _launches(tape.bwd)
    by_lane = collections.Counter(lane for _, lane, _ in bwd)
    assert set(by_lane) == {0, 1, 2, 3}
    wg = [(n, lane, a) for n, lane, a in bwd if n in ('dsnt_conv_wgrad_f16x3', 'dsnt_conv_wgrad_bf16x6')]
    on_lane2 = [x for x in wg if x[1] == 2]
    # DSNT_WGRAD_SHARE_CHIP in `accumulate` — except on the network's first convolution: no data gradient follows it, its
    # weight gradient is the LAST launch of backward and has the chip to itself
    # (+ DSNT_WGRAD_NARROW, bit 2, on the 3x3 ones: half as many slabs again)
    assert len(on_lane2) >= 14 and all(x[2][8] in (2, 6) for x in on_lane2[:-1]) and on_lane2[-1][2][8] == 0
    assert sum(1 for x in on_lane2 if x[2][8] == 6) >= 12
    assert wg[-1] is on_lane2[-1]
    assert all(x[2][8] == 0 for x in wg if x[1] != 2)
    # slab reductions, grouped small weight gradients and the gradient-bucket markers live on the weight-gradient lane
    assert all(lane == 2 for n, lane, _ in bwd if n in ('dsnt_wgrad_reduce_all', 'dsnt_conv_wgrad_group'))
    marks = [(e[1], e[3]) for e in tape.bwd if e[0] is None and e[2] == 'bucket']
    assert marks == [(2, 2), (1, 2), (0, 2)]
    # every skip branch's gradient is joined inside the pool's backward: accumulate = 1 there.  The stack input that also feeds the
    # intermediate-supervision sum (hourglass.py:175) holds that sum's gradient already when its branch's arrives in a buffer of its
    # own: the pool's backward adds it in the same pass (dsnt_maxpool2_bwd_add), and fc_ READS dL/dy in place — no axpy launch left
    pools = [(n, a) for n, _, a in bwd if n.startswith('dsnt_maxpool2_bwd')]
    assert len(pools) == 9 and sum(1 for _, a in pools if a[3] == 1) == 8
    assert sum(1 for n, _ in pools if n == 'dsnt_maxpool2_bwd_add') == 1
    assert sum(1 for n, _, _ in bwd if n.startswith('dsnt_axpy')) == 0
    # the weight gradients of the low-resolution convolutions with a residual input (conv3 of the Bottlenecks below 32 x 32) wait for
    # their bucket's grouped launch as well: dL/dy is the BASE the skip gradient continues out of place (dsnt_bn_act_bwd_apply_base),
    # not a donated buffer — one weight-gradient launch is left on the dependency chain's lanes (fc_: its dL/dy is shared)
    assert sum(1 for x in wg if x[1] in (0, 1, 3)) <= 1
    assert sum(1 for n, _, _ in bwd if n in ('dsnt_bn_act_bwd_apply_base', 'dsnt_bn_act_bwd_apply_pro_base')) >= 14
    # nothing of the per-step weight preparation is left at the head of the backward list
    assert [n for n, _, _ in bwd[:3]][0] == 'dsnt_fill_zero' and 'dsnt_conv_pack_dgrad_all' not in [n for n, _, _ in bwd]


def test_one_pass_backward_of_the_1x1_convolutions(tape):
    """conv1 / conv3 of the 64 x 64 and 32 x 32 Bottlenecks (hourglass.py:20,25) run their whole backward as ONE launch
    (csrc/bwd1.hip): no separate weight gradient, no separate data gradient; for conv1 the BatchNorm backward of bn2 is folded
    in — its finalise launch leaves the bound of dx (dsnt_bn_bwd_finalize_bound) and no apply launch writes dx; launches on the
    skip-branch lanes carry DSNT_CONV_SHARE_CHIP (a workgroup holds most of a CU's LDS for the whole launch)."""
    bwd = _launches(tape.bwd)
    fused = [(lane, a) for n, lane, a in bwd if n == 'dsnt_conv1x1_bwd_f16x3']
    folded = [a for _, a in fused if a[2] is not None]
    assert len(fused) == 31 and len(folded) == 13
    # ... four of them without a BatchNorm in front (two projection shortcuts, two `fc`): dL/dx written as it is, no statistics
    assert sum(1 for _, a in fused if a[9] is None) == 4
    # round 5: bn3's backward rides in the operand load of conv2's data gradient at the same levels (csrc/conv3s.hip MODE 4): one
    # launch per 3x3 convolution of >= 16384 rows, each with its own finalise-with-bound launch, its own materialised dL/dy
    # (argument 2: what the weight gradient reads) and the weight gradient AFTER it in list order
    fold3 = [(i, lane, a) for i, (n, lane, a) in enumerate(bwd) if n == 'dsnt_conv_dgrad_f16x3_stream_apply']
    assert len(fold3) == 13 and len({a[2].value for _, _, a in fold3}) == 13
    assert all(((a[9] & 2) == 2) == (lane != 0) for _, lane, a in fold3)
    for i, _, a in fold3:
        readers = [j for j, (n, _, b) in enumerate(bwd) if n == 'dsnt_conv_wgrad_f16x3' and b[4].value == a[2].value]
        assert len(readers) == 1 and readers[0] > i
    assert sum(1 for n, _, _ in bwd if n == 'dsnt_bn_bwd_finalize_bound') == len(folded) + len(fold3)
    assert all(((a[12] & 2) == 2) == (lane != 0) for lane, a in fused)
    assert {lane for lane, _ in fused} == {0, 1, 3}
    # the dz a folded launch reads is private (it outlives the launches of the op that wrote it) and its bound slot differs per layer
    assert len({a[1].value for a in folded}) == len(folded) and len({a[7].value for a in folded}) == len(folded)
    # what is left of the apply pass: 97 launches before the folds of round 4, 84 after them, 71 with round 5's
    applies = sum(1 for n, _, _ in bwd if n.startswith('dsnt_bn_act_bwd_apply'))
    assert applies <= 71, applies
    # every fused launch's slab is reduced by its bucket's one reduction launch
    assert sum(1 for n, _, _ in bwd if n == 'dsnt_wgrad_reduce_all') == 3


def test_persistent_kernels_share_the_chip_on_side_lanes(tape):
    """DSNT_CONV_SHARE_CHIP (bit 1 of in_relu, argument 9 of dsnt_conv_fwd_f16x3_ex / _stream): set on every 1x1 and stream-kernel
    launch of the skip-branch lanes (1, 3), forward and backward, never on the dependency chain's lane — the persistent 3x3 and
    1x1 kernels hold most of a CU's LDS for a whole launch (csrc/conv3s.hip, csrc/gemm1.hip)."""
    seen = collections.Counter()
    for lst in (tape.fwd, tape.bwd):
        for name, lane, args in _launches(lst):
            if name not in ('dsnt_conv_fwd_f16x3_ex', 'dsnt_conv_fwd_f16x3_stream', 'dsnt_conv1x1_fwd_f16x3'):
                continue
            g = args[12] if name == 'dsnt_conv1x1_fwd_f16x3' else args[13]
            g = getattr(g, '_obj', g)
            flag = int(args[9]) & 2
            if lane == 0:
                assert flag == 0, (name, lane)
            elif name.endswith('_stream') or g.R == 1:
                assert flag == 2, (name, lane)
                seen[name] += 1
    # (the large 1x1 convolutions run forward on the LDS-staged streaming kernel, csrc/fwd1.hip; the tiled kernel keeps the rest)
    assert seen['dsnt_conv_fwd_f16x3_stream'] >= 8 and seen['dsnt_conv1x1_fwd_f16x3'] >= 8, seen
    # the statistics of a fwd1 launch are one row per workgroup: the finalise launch behind it is handed that count
    fwd = _launches(tape.fwd)
    for i, (name, lane, args) in enumerate(fwd):
        if name == 'dsnt_conv1x1_fwd_f16x3' and args[11] is not None:
            fin = next((a for n, _, a in fwd[i + 1:] if n == 'dsnt_bn_finalize' and a[0].value == args[11].value), None)
            assert fin is None or 0 < fin[1] <= 512       # (none: the output goes through an up-sampling that leaves its own statistics)


def test_launch_counts_stay_bounded(tape):
    nf, nb = len(_launches(tape.fwd)), len(_launches(tape.bwd))
    assert nf <= 235 and nb <= 340, (nf, nb)


def _trace(monkeypatch_module, base, training, shape, **kw):
    from dsnt import _lib
    import dsnt.engine as E
    # (every module that binds `ptr` by name is imported BEFORE the patch, or it would keep the permissive one for good)
    from dsnt.model import build_mpii_pose_model
    from dsnt.hourglass import Arena, Program
    monkeypatch_module.setattr(_lib, 'ptr', lambda t: C.c_void_p(t.data_ptr()) if t is not None else None)
    monkeypatch_module.setattr(E._lib, 'ptr', _lib.ptr)
    m = build_mpii_pose_model(base=base, output_strat='dsnt', **kw)
    m.train(training)
    root = m.hg if hasattr(m, 'hg') else m._runner().root
    return Program(root, Arena(root, torch.device('cpu')), shape, training, False).tape


def test_eval_mode_schedule_of_hg2(monkeypatch_module):
    """Inference (inference.py:33-48): no backward list, every BatchNorm's vectors from ONE table-driven launch, no
    finalise launches, no statistics passes, the side lane still carries the skip branches; the large convolutions run
    fp16x3 with bounds their producers leave (dsnt_out_bounds.amax_bn), so the forward list starts by zeroing them."""
    tape = _trace(monkeypatch_module, 'hg2', False, (32, 3, 256, 256), reg='none')
    fwd = _launches(tape.fwd)
    names = [n for n, _, _ in fwd]
    assert not tape.bwd
    assert names.count('dsnt_bn_eval_prep') == 1 and 'dsnt_bn_finalize' not in names and 'dsnt_bn_stats' not in names
    assert names[0] == 'dsnt_fill_zero'
    assert names.count('dsnt_conv_fwd_f16x3_ex') + names.count('dsnt_conv_fwd_f16x3_stream') + names.count('dsnt_conv1x1_fwd_f16x3') >= 40
    assert names.count('dsnt_conv_fwd_f16x3_stream') == 19       # the 3x3 convolutions of the 128 / 64 / 32 / 16 pixel levels
    assert {lane for _, lane, _ in fwd} == {0, 1}            # the forward-only trace forks every skip branch onto lane 1
    assert len(fwd) <= 135, len(fwd)


def test_eval_mode_forward_with_a_backward_list(monkeypatch_module):
    """A backward through an eval-mode forward (autograd on model.eval(): hourglass.Runner traces `record=True` beside
    `training=False`): BatchNorm vectors from the running statistics, no statistics passes forward; backward, every BatchNorm
    finalise carries DSNT_BN_FROZEN (zero coefficients: dx = gamma invstd dz) and none is folded into a consumer's prologue; none of
    the fp16x3 kernels (their backward bounds assume batch statistics)."""
    from dsnt.model import build_mpii_pose_model
    from dsnt.hourglass import Arena, Program
    from dsnt import _lib
    import dsnt.engine as E
    monkeypatch_module.setattr(_lib, 'ptr', lambda t: C.c_void_p(t.data_ptr()) if t is not None else None)
    monkeypatch_module.setattr(E._lib, 'ptr', _lib.ptr)
    m = build_mpii_pose_model(base='hg2', output_strat='dsnt', reg='js')
    m.eval()
    prog = Program(m.hg, Arena(m.hg, torch.device('cpu')), (4, 3, 128, 128), False, False, record=True)
    tape = prog.tape
    assert prog.record and not prog.training and not tape.use_f16x3
    fwd, bwd = _launches(tape.fwd), _launches(tape.bwd)
    fn, bn = [n for n, _, _ in fwd], [n for n, _, _ in bwd]
    assert fn.count('dsnt_bn_eval_prep') == 1 and 'dsnt_bn_finalize' not in fn and not any('stats' in n for n in fn)
    fin = [a for n, _, a in bwd if n == 'dsnt_bn_bwd_finalize']
    assert len(fin) >= 90 and all(int(a[6]) & 2 for a in fin)
    assert 'dsnt_bn_act_bwd_apply_pro' not in bn and 'dsnt_bn_bwd_finalize_bound' not in bn
    assert not any('_f16x3' in n or n.startswith('dsnt_f16_') for n in fn + bn)
    assert bn.count('dsnt_wgrad_reduce_all') == 3


def test_resnet34_train_schedule(monkeypatch_module):
    """BASELINE config 1's model (ResNet-34 + DSNT, batch 8, 8x8 heat-maps): one lane chain (no skip branches to fork),
    post-activation blocks as conv -> [BN+ReLU in the next conv's load] -> conv -> one bn_add_act launch, strided
    convolutions' data gradients on the native phase kernel (no zero-stuffed copies), weight gradients reduced once per
    bucket."""
    tape = _trace(monkeypatch_module, 'resnet34', True, (8, 3, 256, 256))
    fwd, bwd = _launches(tape.fwd), _launches(tape.bwd)
    fn, bn = [n for n, _, _ in fwd], [n for n, _, _ in bwd]
    assert fn.count('dsnt_bn_add_act_fwd') == 16                     # 3 + 4 + 6 + 3 BasicBlocks
    assert fn.count('dsnt_maxpool3s2_fwd') == 1 and bn.count('dsnt_maxpool3s2_bwd') == 1
    assert bn.count('dsnt_conv_dgrad_strided') == 6                  # three stage transitions x (conv1 + downsample)
    assert bn.count('dsnt_zero_insert') == 0
    assert bn.count('dsnt_wgrad_reduce_all') >= 1 and bn.count('dsnt_wgrad_reduce_all') <= 6
    assert not any(n.startswith('dsnt_axpy') for n in fn)
    assert len(fwd) <= 180 and len(bwd) <= 330, (len(fwd), len(bwd))


def test_persistent_stage_runs_of_the_recorded_lists(tape):
    """Round 6 (csrc/stage.h): the 8 x 8 / 4 x 4 levels of every hourglass (hourglass.py:78-90) are runs of small dependent launches
    of one lane that dsnt_list_fuse turns into persistent stage launches.  Recording a list needs no GPU (launches are captured, not
    enqueued); dsnt_list_fuse_plan runs the same run finder as dsnt_list_fuse."""
    import ctypes as C
    from dsnt import _lib
    lib = _lib.load()
    for lst, lo, hi in ((tape.fwd, 2 * 24, 2 * 32), (tape.bwd, 2 * 40, 2 * 56)):
        h, marks = tape._compile(lst)
        try:
            size = lib.dsnt_list_size(h)
            n = C.c_int(0)
            s = lib.dsnt_list_fuse_plan(h, 3, 512, C.byref(n))
            # per hourglass: the 4 x 4 chain, the two 8 x 8 runs around it and the 8 x 8 skip branch on its side lane
            assert 2 * 3 <= s <= 2 * 5, (s, n.value)
            assert lo <= n.value <= hi, (s, n.value)
            assert lib.dsnt_list_fuse_bytes(h, 3, 512) == 512 * n.value + 64 * s
            # nothing to fuse if runs had to be longer than any run is / launches narrower than any of these are
            assert lib.dsnt_list_fuse_plan(h, 64, 512, None) == 0 and lib.dsnt_list_fuse_plan(h, 3, 1, None) == 0
            assert lib.dsnt_list_size(h) == size        # planning changes nothing
            # argument checks of the fusing call come before anything touches the device: more than 256 stage workgroups cannot be
            # co-resident (the barrier would wait for ever — bounded, but wrong); the workspace is the caller's and must be big enough
            assert lib.dsnt_list_fuse(h, None, 0, 3, 512, 300) != 0 and b'grid_cap' in lib.dsnt_last_error()
            assert lib.dsnt_list_fuse(h, None, 0, 3, 512, 64) != 0 and b'workspace' in lib.dsnt_last_error()
            assert lib.dsnt_list_fuse(h, C.c_void_p(64), 64, 3, 512, 64) != 0 and b'workspace' in lib.dsnt_last_error()
            assert lib.dsnt_list_size(h) == size and lib.dsnt_list_stages(h, None) == 0
        finally:
            lib.dsnt_list_destroy(h)
