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


def test_bucketed_allreduce_world2():
    world = 2
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
    assert sorted(r[0] for r in results) == [0, 1]
    assert all(r[1] and r[2] and r[3] for r in results), results


def test_single_process_is_a_noop():
    from dsnt import parallel
    flat = torch.ones(10)
    red = parallel.GradientAllReducer(flat, [(0, 10)])
    red.reduce_all()
    red.wait()
    assert torch.equal(flat, torch.ones(10))
