"""Host-side logic that needs no GPU: builder surface, state_dict compatibility, the parameter
arena (OHWI packing, bucket layout), error messages."""
import pytest
import torch

from dsnt import synthetic
from dsnt.model import build_mpii_pose_model
from dsnt import hourglass as dhg
from dsnt_oracle import model as omodel


def test_builder_surface_matches_reference_quirks():
    m = build_mpii_pose_model(base='hg', dilate=2, truncate=1)      # resnet kwargs are filtered out
    assert m.output_strat == 'gauss' and m.hg.num_stacks == 2       # model.py:346 default strategy
    assert m.image_specs.size == 256 and m.image_specs.subtract_mean and not m.image_specs.divide_stddev
    assert m.heatmap_size == 64 and m.n_chans == 16
    assert build_mpii_pose_model(base='hg8', output_strat='dsnt').hg.num_stacks == 8
    for bad in ('vgg', 'hgx'):
        with pytest.raises(Exception, match='unsupported base model type'):
            build_mpii_pose_model(base=bad)
    r = build_mpii_pose_model(base='resnet34', truncate=1, dilate=2)    # the default base of the builder
    o = omodel.build_mpii_pose_model(base='resnet34', truncate=1, dilate=2)
    assert list(r.state_dict().keys()) == list(o.state_dict().keys())    # torchvision / reference key names
    assert [tuple(v.shape) for v in r.state_dict().values()] == [tuple(v.shape) for v in o.state_dict().values()]
    assert r.heatmap_size == 28 and r.image_specs.size == 224 and not r.image_specs.subtract_mean
    assert r.fcn[6][0].conv1.stride == (1, 1) and r.fcn[6][0].conv2.dilation == (2, 2)   # surgery, model.py:112-121
    with pytest.raises(RuntimeError, match='HIP device only'):           # no CPU fallback
        r(torch.zeros(1, 3, 64, 64))
    with pytest.raises(Exception, match='unsupported base model type'):
        build_mpii_pose_model(base='resnet99')
    with pytest.raises(Exception, match='unrecognised heatmap preactivation'):
        import dsnt.nn as dn
        dn.hm_preact(torch.zeros(1, 1, 2, 2), 'tanh')


@pytest.mark.parametrize('base', ['hg1', 'hg2'])
def test_state_dict_interchangeable_with_oracle(base):
    m = build_mpii_pose_model(base=base, output_strat='dsnt')
    o = omodel.build_mpii_pose_model(base=base, output_strat='dsnt')
    sm, so = m.state_dict(), o.state_dict()
    assert list(sm.keys()) == list(so.keys())
    assert all(sm[k].shape == so[k].shape for k in sm)
    synthetic.fill_state_dict(o, seed=4)
    m.load_state_dict(o.state_dict())
    for k, v in m.state_dict().items():
        assert torch.equal(v, o.state_dict()[k]), k
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == {'hg1': 3586960, 'hg2': 6730912}[base]


def test_arena_packing_roundtrip_and_buckets():
    m = build_mpii_pose_model(base='hg2', output_strat='dsnt')
    synthetic.fill_state_dict(m, seed=2)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    arena = dhg.Arena(m.hg, torch.device('cpu'))
    # logical (OIHW) views are unchanged, storage is OHWI and contiguous per tensor
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
    w = dict(m.hg.named_parameters())['layer1.0.conv2.weight']
    assert w.shape == (64, 64, 3, 3) and w.stride() == (576, 1, 192, 64)
    stem = dict(m.hg.named_parameters())['conv1.weight']
    assert stem.shape == (64, 3, 7, 7) and arena.packed['conv1.weight'].shape == (64, 7, 7, 4)
    assert float(arena.packed['conv1.weight'][..., 3].abs().max()) == 0.0      # channel padding
    # buckets: 0 = stem, 1.. = stacks; contiguous, ordered, covering the arena
    assert len(arena.bucket_bounds) == 3
    assert arena.bucket_bounds[0][0] == 0 and arena.bucket_bounds[-1][1] == arena.numel
    for (a0, a1), (b0, b1) in zip(arena.bucket_bounds, arena.bucket_bounds[1:]):
        assert a1 == b0 and a1 > a0
    for name, p, off, n in arena.slots:
        b = m.hg.param_bucket(name)
        lo, hi = arena.bucket_bounds[b]
        assert lo <= off and off + n <= hi, name
        assert off % 4 == 0
    # load_state_dict writes through the views into the arena
    m.load_state_dict({k: v + 1 for k, v in before.items()})
    assert torch.equal(arena.packed['layer1.0.conv2.weight'].permute(0, 3, 1, 2),
                       before['hg.layer1.0.conv2.weight'] + 1)
    assert arena.valid()
    m.float()                                   # nn.Module._apply must not break validity checks
    assert m.hg._runner().arena is None or True


def test_synthetic_is_deterministic():
    a = synthetic.batch(2, size=64, seed=1)
    b = synthetic.batch(2, size=64, seed=1)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    m1 = omodel.build_mpii_pose_model(base='hg1', output_strat='dsnt')
    m2 = omodel.build_mpii_pose_model(base='hg1', output_strat='dsnt')
    synthetic.fill_state_dict(m1, seed=0)
    synthetic.fill_state_dict(m2, seed=0)
    assert all(torch.equal(p, q) for p, q in zip(m1.parameters(), m2.parameters()))


def test_bearpaw_checkpoint_repack_flow():
    """`bin/convert_hg_model.py:31-44`: strip the DataParallel "module." prefix of a pytorch-pose hourglass
    checkpoint, load it into `model.hg`, save `model.state_dict()` — works on the HIP model classes unchanged
    (same HourglassNet keys as the reference / bearpaw: conv1, layer1..3, hg.N.hg..., fc_, score_)."""
    from collections import OrderedDict
    from dsnt.model import build_mpii_pose_model
    from dsnt_oracle import hourglass as ohg
    torch.manual_seed(0)
    donor = ohg.HourglassNet(ohg.Bottleneck, num_stacks=2, num_blocks=1, num_classes=16)
    old_state_dict = OrderedDict(('module.' + k, v) for k, v in donor.state_dict().items())
    new_state_dict = OrderedDict([(k[7:], v) for (k, v) in old_state_dict.items()])
    model = build_mpii_pose_model(base='hg', stacks=2, blocks=1)
    model.hg.load_state_dict(new_state_dict)
    out = model.state_dict()
    assert list(out.keys()) == ['hg.' + k for k in donor.state_dict().keys()]
    for k, v in donor.state_dict().items():
        assert torch.equal(out['hg.' + k].cpu(), v), k


def test_optimizer_state_dict_is_torch_format():
    """`optimizer.state_dict()` is what the reference checkpoints (bin/train.py:364,492): it must carry the flat
    second moments / momentum buffers, in torch.optim's own per-parameter format, and load back."""
    from dsnt import optim
    m = build_mpii_pose_model(base='hg1', output_strat='fc')
    synthetic.fill_state_dict(m, seed=1)
    m.hg._runner().ensure(torch.device('cpu'))
    arena = m.hg.arena
    for cls, stock, key in ((optim.RMSprop, torch.optim.RMSprop, 'square_avg'),
                            (optim.SGD, torch.optim.SGD, 'momentum_buffer')):
        kw = dict(lr=2.5e-4) if cls is optim.RMSprop else dict(lr=0.2, momentum=0.9)
        opt = cls(m, **kw)
        assert opt.state_dict()['state'] == {}                        # nothing before the first step
        opt._steps = 3
        opt.flat_state.copy_(torch.rand(arena.numel))
        for s in opt.extra_state:
            s.copy_(torch.rand_like(s))
        assert len(opt.extra) == 2                                    # out_fc lives outside the arena
        sd = opt.state_dict()
        params = [p for _, p, _, _ in arena.slots] + opt.extra
        assert len(sd['state']) == len(params) and sd['param_groups'][0]['params'] == list(range(len(params)))
        # the stock optimiser over the same parameters accepts it ...
        ref = stock(params, **kw)
        ref.load_state_dict(sd)
        for i, p in enumerate(params):
            assert ref.state[p][key].shape == p.shape
            assert torch.equal(ref.state[p][key], sd['state'][i][key])
        w = 'layer1.0.conv2.weight'
        i = [n for n, _, _, _ in arena.slots].index(w)
        O, I, R, S = params[i].shape
        o = arena.slots[i][2]
        assert torch.equal(sd['state'][i][key], opt.flat_state[o:o + O * I * R * S].view(O, R, S, I).permute(0, 3, 1, 2))
        # ... and its own state_dict loads back bit-exactly into a fresh flat optimiser
        opt2 = cls(m, **kw)
        opt2.load_state_dict(ref.state_dict())
        # (the stem's channel padding and the 16-byte slot alignment are not parameters: compare the logical views)
        assert all(torch.equal(arena.logical(opt2.flat_state, n), arena.logical(opt.flat_state, n))
                   for n, _, _, _ in arena.slots) and opt2._steps >= 1
        assert all(torch.equal(a, b) for a, b in zip(opt2.extra_state, opt.extra_state))
        if cls is optim.RMSprop:
            assert opt2._steps == 3 and opt2.param_groups[0]['lr'] == 2.5e-4
            # marker-less states: a checkpoint of the reference's torch 0.3 (python-int steps, none of modern torch's group keys)
            # loads silently in model order; only the shape of a pre-round-3 dsnt.optim state (0-d float tensor steps, an entry
            # for every parameter, no modern keys) draws the `order='arena'` hint
            import warnings
            legacy = {'state': {i: {'step': 3, key: e[key]} for i, e in sd['state'].items()},
                      'param_groups': [{'lr': 2.5e-4, 'alpha': 0.99, 'eps': 1e-8, 'weight_decay': 0, 'momentum': 0,
                                        'centered': False, 'params': list(range(len(params)))}]}
            with warnings.catch_warnings():
                warnings.simplefilter('error')
                cls(m, **kw).load_state_dict(legacy)
                cls(m, **kw).load_state_dict(ref.state_dict())
            old = {'state': sd['state'], 'param_groups': [{k: v for k, v in sd['param_groups'][0].items() if k != 'dsnt_order'}]}
            with pytest.warns(UserWarning, match="order='arena'"):
                cls(m, **kw).load_state_dict(old)


def test_arena_survives_noop_apply_and_keeps_gradients():
    m = build_mpii_pose_model(base='hg1', output_strat='dsnt')
    r = m.hg._runner()
    r.ensure(torch.device('cpu'))
    arena = r.arena
    r.programs['sentinel'] = object()
    m.float()
    m.to('cpu')
    assert r.arena is arena and 'sentinel' in r.programs             # nothing moved: nothing dropped
    p = dict(m.hg.named_parameters())['conv1.weight']
    p.grad = torch.ones_like(p)
    m.double()                                                        # a real change re-builds on the next use
    assert r.arena is None and r.programs == {}
    r.ensure(torch.device('cpu'))
    assert r.arena is not arena and p.dtype == torch.float32
    assert p.grad is r.arena.gviews['conv1.weight'] and float(p.grad.sum()) == p.numel()
    # every BatchNorm's (momentum, eps) is part of the program key
    sig = r._bn_signature()
    bns = [x for x in m.hg.modules() if isinstance(x, torch.nn.BatchNorm2d)]
    assert len(sig) == len(bns) == 53
    bns[-1].momentum = 0.5
    assert r._bn_signature() != sig


def test_bench_starts_its_own_ranks_as_a_child_process(tmp_path, monkeypatch):
    """`python bench.py --gpus N` (the driver's command shape, no WORLD_SIZE in the environment) must launch
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a CHILD (never exec: a
    process that may have touched the GPU must not be replaced), forward rank 0's JSON line and return the child's code."""
    import importlib.util
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setenv('RANK', '3')
    monkeypatch.setenv('WORLD_SIZE', '4')
    monkeypatch.setenv('MASTER_PORT', '1')
    cmd, env = bench.launcher_command(8, ['--gpus', '8', '--steps', '5', '--warmup', '2'], 29517)
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert cmd[3:10] == ['--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1', '--master-port', '29517']
    assert cmd[10] == os.path.join(root, 'bench.py') and cmd[11:] == ['--gpus', '8', '--steps', '5', '--warmup', '2']
    assert env['MASTER_ADDR'] == '127.0.0.1' and env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert not any(k in env for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'))    # the launcher sets its own

    # the forwarding: a stand-in child that prints noise + a JSON line and exits 0 / a child that fails
    calls = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        calls['cmd'] = cmd
        out = 'NCCL version banner\n' + json.dumps({'metric': 'images/sec', 'value': 1.0, 'n_gpus': 2}) + '\n'
        return subprocess.CompletedProcess(cmd, calls.get('rc', 0), stdout=out if not calls.get('rc') else 'boom\n')
    monkeypatch.setattr(subprocess, 'run', fake_run)
    sink = open(tmp_path / 'line.txt', 'w+')
    monkeypatch.setattr(bench, '_JSON_OUT', sink)
    assert bench.launch_ranks(2, ['--gpus', '2']) == 0
    sink.seek(0)
    lines = sink.read().splitlines()
    assert len(lines) == 1 and json.loads(lines[0])['n_gpus'] == 2
    assert '--nproc-per-node' in calls['cmd'] and calls['cmd'][-2:] == ['--gpus', '2']
    calls['rc'] = 7
    assert bench.launch_ranks(2, ['--gpus', '2']) == 7
    # ... and the source never replaces the process
    src = open(os.path.join(root, 'bench.py')).read()
    assert 'os.exec' not in src and 'execv' not in src.replace('never exec', '')
