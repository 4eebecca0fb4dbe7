"""World-size-2 tests of the data-parallel host logic on the gloo backend (CPU tensors):
bucketed asynchronous all-reduce over a flat gradient arena, weight broadcast, batch sharding,
and 'mean of per-shard gradients' semantics."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from dsnt import parallel
        # flat "gradient arena" with 3 buckets; rank r holds (r+1) * arange
        n = 1000
        flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
        bounds = [(0, 100), (100, 640), (640, 1000)]
        red = parallel.GradientAllReducer(flat, bounds)
        for k in (2, 1, 0):                    # backward completes the last bucket first
            red.bucket_ready(k)
        red.wait()
        want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        ok_sum = torch.equal(flat, want)
        # the non-finite guard's flag is exchanged (MAX) in wait(), behind the last bucket: rank 1 saw a NaN loss, rank 0 did
        # not -> both ranks end with the loss bit up and skip the same update (bin/train.py:360-371 stops its one process);
        # a flag raised AFTER the first bucket went out (guard.check enqueued after backward started) is still covered
        flag = torch.tensor([1 if rank == 1 else 0, 0], dtype=torch.int32)
        red3 = parallel.GradientAllReducer(torch.ones(8) * (rank + 1), [(0, 4), (4, 8)], flag=flag)
        flag0 = torch.zeros(2, dtype=torch.int32)
        red3 = parallel.GradientAllReducer(torch.ones(8) * (rank + 1), [(0, 4), (4, 8)], flag=flag0)
        red3.bucket_ready(1)
        flag0.copy_(flag)                      # raised between the first bucket and the wait
        red3.bucket_ready(0)
        n_pending = len(red3.pending)          # two buckets; the flag goes out in wait()
        red3.wait()
        ok_sum = ok_sum and flag0.tolist() == [1, 0] and n_pending == 2 and red3.bucket_bytes() == [16, 16]
        red3.flag.zero_()                      # the next backward exchanges it again
        red3.reduce_all()
        ok_sum = ok_sum and len(red3.pending) == 2
        red3.wait()
        ok_sum = ok_sum and flag0.tolist() == [0, 0] and red3.exposed_comm_ms() is None
        # a check enqueued AFTER backward() returned: wait() has already exchanged the flag for this step, so the
        # optimiser's pre_update hook (reducer.sync_flag) exchanges it once more — every rank skips, none diverges; with no late
        # check there is no second collective
        class _Guard:                          # the two fields of dsnt.guard.NanGuard the reducer looks at
            def __init__(self, flag):
                self.flag, self.checks_enqueued = flag, 0
        flag4 = torch.zeros(2, dtype=torch.int32)
        g4 = _Guard(flag4)
        red4 = parallel.GradientAllReducer(torch.ones(4) * (rank + 1), [(0, 4)], flag=flag4)
        red4.guard = g4
        g4.checks_enqueued += 1                # the usual check, before backward: clean on both ranks
        red4.reduce_all()
        red4.wait()
        seen = red4.flag_checks_seen
        red4.sync_flag()                       # nothing new: no collective, nothing changes
        ok_sum = ok_sum and seen == 1 and flag4.tolist() == [0, 0]
        g4.checks_enqueued += 1                # a LATE check: rank 1's loss was not finite
        if rank == 1:
            flag4[0] = 1
        red4.sync_flag()
        ok_sum = ok_sum and flag4.tolist() == [1, 0] and red4.flag_checks_seen == 2
        # broadcast rank 0's weights
        w = torch.full((17,), float(rank + 5))
        parallel.broadcast_flat(w, 0)
        ok_bcast = bool((w == 5).all())

        # 'mean of per-shard gradients' == gradient of the mean loss over equal shards
        torch.manual_seed(0)
        lin = torch.nn.Linear(8, 3)
        x = torch.randn(4 * world, 8)
        y = torch.randn(4 * world, 3)
        per = x.shape[0] // world
        xs, ys = x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]
        loss = ((lin(xs) - ys) ** 2).mean()
        loss.backward()
        g = torch.cat([p.grad.reshape(-1) for p in lin.parameters()])
        red2 = parallel.GradientAllReducer(g, [(0, g.numel())])
        red2.reduce_all()
        red2.wait()
        g = g / world
        lin2 = torch.nn.Linear(8, 3)
        lin2.load_state_dict(lin.state_dict())
        ((lin2(x) - y) ** 2).mean().backward()
        g_full = torch.cat([p.grad.reshape(-1) for p in lin2.parameters()])
        ok_mean = bool((g - g_full).abs().max() < 1e-6)
        q.put((rank, ok_sum, ok_bcast, ok_mean))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])       # (4: the 4-GPU point of the scaling curve has the same host logic as 2 and 8)
def test_bucketed_allreduce_world2(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    assert all(r[1] and r[2] and r[3] for r in results), results


def _dp_worker(rank, world, port, q):
    """DataParallel attached to real (CPU-resident) model arenas: markers that never fire, parameters outside
    the arena, weight broadcast, shard()."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from dsnt import parallel, synthetic
        from dsnt.model import build_mpii_pose_model
        res = {}
        for base, strat in (('hg2', 'fc'), ('resnet18', 'dsnt')):
            m = build_mpii_pose_model(base=base, output_strat=strat)
            synthetic.fill_state_dict(m, seed=rank)                 # ranks start from DIFFERENT weights
            runner = (m.hg if hasattr(m, 'hg') else m)._runner()
            runner.ensure(torch.device('cpu'))
            arena = runner.arena
            dp = parallel.DataParallel(m)
            ref = build_mpii_pose_model(base=base, output_strat=strat)
            synthetic.fill_state_dict(ref, seed=0)
            same = all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), ref.state_dict().values()))
            # a backward whose list announces only the LAST bucket (or none at all, as ResNet models did):
            arena.fresh.copy_(torch.arange(arena.numel, dtype=torch.float32) % 97 * (rank + 1))
            nb = len(arena.bucket_bounds)
            if nb > 1:
                runner.bucket_hook(nb - 1)
            runner.before_publish()
            want = torch.arange(arena.numel, dtype=torch.float32) % 97 * sum(r + 1 for r in range(world))
            late = list(dp.reducer.last_late)
            ok_sum = torch.equal(arena.fresh, want) and arena.publish_scale == 1.0 / world
            # parameters outside the arena get the mean gradient from the hook
            ok_extra = True
            if dp.extra:
                inp = torch.ones(3, m.out_fc.in_features)
                (m.out_fc(inp).sum() * (rank + 1)).backward()
                mean = sum(r + 1 for r in range(world)) / world
                ok_extra = bool(torch.allclose(m.out_fc.weight.grad, torch.full_like(m.out_fc.weight, 3.0 * mean)) and
                                torch.allclose(m.out_fc.bias.grad, torch.full_like(m.out_fc.bias, 3.0 * mean)))
            res[base] = (same, ok_sum, late, len(dp.extra), ok_extra, nb)
        x = torch.arange(8.0).view(8, 1)
        shard_ok = torch.equal(dp.shard(x), x[rank * 4:(rank + 1) * 4])
        try:
            dp.shard(torch.zeros(7, 1))
            shard_ok = False
        except ValueError:
            pass
        q.put((rank, res, shard_ok))
    finally:
        dist.destroy_process_group()


def test_dataparallel_host_logic_world2():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res, shard_ok in results:
        assert shard_ok
        same, ok_sum, late, n_extra, ok_extra, nb = res['hg2']
        assert same and ok_sum and ok_extra and n_extra == 2 and nb == 3 and late == [0, 1], res
        same, ok_sum, late, n_extra, ok_extra, nb = res['resnet18']
        assert same and ok_sum and n_extra == 0 and nb == 1 and late == [0], res


def test_single_process_is_a_noop():
    from dsnt import parallel
    flat = torch.ones(10)
    red = parallel.GradientAllReducer(flat, [(0, 10)])
    red.reduce_all()
    red.wait()
    assert torch.equal(flat, torch.ones(10))
