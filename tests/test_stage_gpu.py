"""The persistent low-resolution stage (csrc/stage.h, round 6) against the separate launches it replaces.

On the 8 x 8 and 4 x 4 levels of an hourglass (/root/reference/src/dsnt/hourglass.py:78-90: the inner `_hour_glass_forward`
recursions) a train step is runs of small dependent launches of one lane; `dsnt_list_fuse` replays each run as ONE persistent
kernel that walks the recorded launches' workgroup indices through the SAME device functions, with a chip-wide barrier where a
kernel boundary was.  Same instructions, same summation order: every output of a step — loss, coordinates, heat-maps, running
statistics, every parameter gradient — must be BIT-identical with the stage on (DSNT_STAGE=1) and off (the default), the stages must have
replaced the launches they were built for (census), and no barrier may have given up.  (This is HIP-vs-HIP by construction; the
oracle statement comes from the goldens and every-gradient tests of tests/test_model_gpu.py on the default path.)

The stage is OFF by default: the same-box A/B is negative (profiles/r06_stage_ab.txt — hg2 batch 32 +0.9 ms, hg8 batch 16 +1.6 ms with
every run fused, +-0 with only the 4 x 4 level's launches inside).  The tests stay because the kernels' device functions are shared
with the stand-alone launches (csrc/ew_bodies.h, conv_ksplit_body) and this is the check that both callers see the same bits."""
import pytest
import torch

from dsnt import synthetic

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _step(base, batch, size, steps=1):
    from dsnt.model import build_mpii_pose_model
    from dsnt import optim
    m = build_mpii_pose_model(base=base, output_strat='dsnt', reg='js')
    synthetic.fill_state_dict(m, seed=0)
    m.cuda().train()
    x, t, k = synthetic.batch(batch, size=size, seed=1, mask_p=0.9)
    x, t, k = x.to(DEV), t.to(DEV), k.to(DEV)
    m.hg._runner().ensure(torch.device(DEV))
    opt = optim.RMSprop(m, lr=2.5e-4)
    for _ in range(steps):
        out = m(x)
        loss = m.forward_loss(out, t, k)
        opt.zero_grad()
        loss.backward()
        if steps > 1:
            opt.step()
    torch.cuda.synchronize()
    runner = m.hg._runner()
    prog = [p for p in runner.programs.values() if p.training][0]
    res = {'loss': loss.detach().clone(), 'coords': [o.detach().clone() for o in out],
           'heatmaps': m.heatmaps.detach().clone(),
           'grads': {n: p.grad.detach().clone() for n, p in m.named_parameters()},
           'buffers': {n: b.detach().clone() for n, b in m.named_buffers()},
           'params': {n: p.detach().clone() for n, p in m.named_parameters()}}
    return res, prog.tape


def _assert_identical(a, b):
    assert torch.equal(a['loss'], b['loss']), (a['loss'].item(), b['loss'].item())
    for u, v in zip(a['coords'], b['coords']):
        assert torch.equal(u, v)
    assert torch.equal(a['heatmaps'], b['heatmaps'])
    for group in ('grads', 'buffers', 'params'):
        bad = [n for n in a[group] if not torch.equal(a[group][n], b[group][n])]
        assert not bad, (group, len(bad), bad[:5])


@pytest.mark.parametrize('base,batch,size,steps', [('hg2', 2, 256, 1), ('hg8', 2, 128, 1), ('hg2', 4, 128, 3)])
def test_stage_is_bit_identical_to_the_launches_it_replaces(monkeypatch, base, batch, size, steps):
    monkeypatch.setenv('DSNT_STAGE', '1')
    on, tape_on = _step(base, batch, size, steps)
    stacks = int(base[2:])
    n_stage = sum(s for s, _ in tape_on.stage_census)
    n_inside = sum(n for _, n in tape_on.stage_census)
    # per hourglass and direction: the 4 x 4 chain, the 8 x 8 runs around it, the 8 x 8 skip branch (more on these small inputs,
    # whose 16 x 16 / 32 x 32 levels are K-split sized too)
    assert n_stage >= 2 * 3 * stacks and n_inside >= 2 * 25 * stacks, tape_on.stage_census
    assert tape_on.stage_errors() == 0
    monkeypatch.delenv('DSNT_STAGE')
    off, tape_off = _step(base, batch, size, steps)
    assert tape_off.stage_census == [] and tape_off.stage_errors() == 0
    _assert_identical(on, off)


def test_stage_full_size_steps_are_identical_and_reproducible(monkeypatch):
    """hg2 at batch 32 / 256 px (BASELINE config 3): the production run structure (8 stages per list: profiles/r06_*), three
    optimiser steps with the stage on twice and off once — same bits everywhere."""
    monkeypatch.setenv('DSNT_STAGE', '1')
    a, tape = _step('hg2', 32, 256, 3)
    assert sum(s for s, _ in tape.stage_census) == 16 and tape.stage_errors() == 0, tape.stage_census
    b, _ = _step('hg2', 32, 256, 3)
    _assert_identical(a, b)
    monkeypatch.delenv('DSNT_STAGE')
    c, _ = _step('hg2', 32, 256, 3)
    _assert_identical(a, c)


def test_stage_first_step_of_a_cold_process_hg8(tmp_path, monkeypatch):
    """The case that exposed the one real bug of the stage: the FIRST step of hg8 at batch 16 in a fresh process.  `__syncthreads()`
    fences LDS only, so a wave could wait in the stage's barrier with global stores still in flight while lane 0 announced the launch
    as done — a workgroup on another XCD then read stale bytes (3 % error in ONE layer's gradients, found by running the whole suite
    under DSNT_STAGE=1).  Every wave now releases at agent scope in front of the barrier (csrc/conv.hip `stage_barrier`); this run
    compares every gradient of that first step, stage on against off, bit for bit, each in a child process of its own."""
    import test_fallback_gpu as tf
    monkeypatch.setenv('DSNT_STAGE', '1')
    on = tf._run(tmp_path, 'stage_on', None, 'hg8', 16, 256)
    monkeypatch.delenv('DSNT_STAGE')
    off = tf._run(tmp_path, 'stage_off', None, 'hg8', 16, 256)
    assert on['loss'] == off['loss'] and torch.equal(on['coords'], off['coords'])
    bad = [n for n, v in off['grads'].items() if not torch.equal(on['grads'][n], v)]
    assert not bad, (len(bad), bad[:5])
    for n, v in off['running'].items():
        assert torch.equal(on['running'][n], v), n
