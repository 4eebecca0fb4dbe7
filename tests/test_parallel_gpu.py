"""Data parallelism on the GPU with TWO ranks (north_star: gradient all-reduce overlapped with backward).
On a box with >= 2 devices the process group is `nccl` (= RCCL), one device per rank — the production transport, whose
collectives are ordered after the CURRENT stream at enqueue (the bucket hook runs with the producing lane's stream
current) and whose Work.wait() holds the publishing stream, not the host.  On the one-MI355X test box both ranks share
the device, so the group is `gloo` over device tensors (RCCL refuses two ranks on one device); everything else is the
production path: the traced backward list with its three lanes, the
bucket markers fired on the lane that finishes a bucket's slab reduction, the asynchronous all-reduce per bucket, the
1/world scale in the publish kernel.  Checked: every rank ends with the SAME gradient, equal to the mean of the two
shards' single-process gradients, for an hourglass and a ResNet model (the latter has one bucket and an out-of-arena
`out_fc` whose gradient is averaged by a post-accumulate hook)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _backend():
    """('nccl', one device per rank) when the box has two devices, else ('gloo', both ranks on cuda:0)."""
    return 'nccl' if torch.cuda.device_count() >= 2 else 'gloo'


def _init(rank, world, port, backend):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = torch.device('cuda:%d' % (rank if backend == 'nccl' else 0))
    torch.cuda.set_device(dev)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    assert dist.get_backend() == backend
    return dev


def _worker(rank, world, port, base, strat, size, q, backend):
    import torch.distributed as dist
    from dsnt.model import build_mpii_pose_model
    from dsnt import parallel, synthetic
    dev = _init(rank, world, port, backend)
    try:
        m = build_mpii_pose_model(base=base, output_strat=strat, reg='js')
        synthetic.fill_state_dict(m, seed=0)
        m.to(dev).train()
        x, t, k = synthetic.batch(2 * world, size=size, seed=3, mask_p=0.9)
        sl = slice(2 * rank, 2 * rank + 2)
        x, t, k = x[sl].to(dev), t[sl].to(dev), k[sl].to(dev)

        def grads():
            for p in m.parameters():
                p.grad = None
            m.forward_loss(m(x), t, k).backward()
            torch.cuda.synchronize()
            return torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()

        local = grads()                                   # this shard alone
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        mean = sum(g.double() for g in gathered) / world
        dp = parallel.DataParallel(m)
        got = grads()
        late = list(dp.reducer.last_late)
        both = [torch.empty_like(got) for _ in range(world)]
        dist.all_gather(both, got)
        scale = float(mean.abs().max())
        q.put((rank, float((got.double() - mean).abs().max()) / scale, bool(torch.equal(both[0], both[1])), late,
               float((local.double() - mean).abs().max()) / scale))
    except Exception as e:      # noqa: BLE001 — reported to the parent
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('base,strat,size', [('hg2', 'dsnt', 128), ('resnet18', 'fc', 224)])
def test_two_ranks_average_their_gradients(base, strat, size):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, base, strat, size, q, _backend())) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for r in res:
        if len(r) == 2 and 'gloo' in r[1].lower() and ('cuda' in r[1].lower() or 'hip' in r[1].lower()):
            pytest.skip('this torch build has no gloo collectives on device tensors: %s' % r[1])
        assert len(r) == 5, r
        rank, err, same, late, spread = r
        assert same, 'ranks ended with different gradients'
        assert err <= 2e-6, (rank, err)                  # the mean of the shards' gradients (fp32 sum, then * 1/world)
        assert spread > 1e-3, spread                      # ... and the shards really differed
        assert late == [], late                           # every bucket was announced by its marker, none at the wait


def _guard_worker(rank, world, port, q, backend, check_late):
    import torch.distributed as dist
    from dsnt.model import build_mpii_pose_model
    from dsnt import parallel, synthetic, optim
    from dsnt.guard import NanGuard
    dev = _init(rank, world, port, backend)
    try:
        m = build_mpii_pose_model(base='hg1', output_strat='dsnt', reg='js')
        synthetic.fill_state_dict(m, seed=0)
        m.to(dev).train()
        x, t, k = synthetic.batch(2 * world, size=128, seed=3, mask_p=0.9)
        sl = slice(2 * rank, 2 * rank + 2)
        x, t, k = x[sl].to(dev), t[sl].to(dev), k[sl].to(dev)
        m.hg._runner().ensure(dev)
        guard = NanGuard(dev)
        opt = optim.RMSprop(m, lr=2.5e-4, guard=guard)
        parallel.DataParallel(m, opt)

        def step(poison=None):
            loss = m.forward_loss(m(x), t, k)
            if poison is not None:
                loss = loss * poison
            if not check_late:
                guard.check(loss)                # train.py:360: right after forward_loss
            opt.zero_grad()
            if check_late:
                # ... or enqueued while backward is under way (here: from a hook on the loss, i.e. after the first launches
                # of backward): the flag goes out in the reducer's wait(), behind the last bucket, so it is still exchanged
                loss.register_hook(lambda g: (guard.check(loss), g)[1])
            loss.backward()
            opt.step()
            torch.cuda.synchronize()

        def gathered():
            mine = m.hg.arena.params.clone()
            both = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            return mine, bool(torch.equal(both[0], both[1]))

        start = m.hg.arena.params.clone()
        step()
        after1, same1 = gathered()
        step(poison=float('nan') if rank == 1 else None)       # ONE rank's loss is NaN
        after2, same2 = gathered()
        q.put((rank, same1, not torch.equal(start, after1), same2, bool(torch.equal(after1, after2)), guard.flag.tolist()))
    except Exception as e:      # noqa: BLE001 — reported to the parent
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('check_late', [False, True], ids=['check_before_backward', 'check_during_backward'])
def test_a_nonfinite_loss_on_one_rank_stops_the_update_on_every_rank(check_late):
    """train.py:360-371 under data parallelism: the guard's device flag is exchanged (MAX) behind the last gradient bucket,
    so a NaN loss on rank 1 makes BOTH ranks skip the whole optimiser update — their parameters stay bit-identical (a
    rank-local flag would let rank 0 apply the finite elements of the all-reduced gradient and the replicas diverge)."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_guard_worker, args=(r, 2, port, q, _backend(), check_late)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for r in res:
        if len(r) == 2 and 'gloo' in r[1].lower() and ('cuda' in r[1].lower() or 'hip' in r[1].lower()):
            pytest.skip('this torch build has no gloo collectives on device tensors: %s' % r[1])
        assert len(r) == 6, r
        rank, same1, moved1, same2, frozen2, flag = r
        assert same1 and moved1                 # a finite step: both ranks move, identically
        assert same2 and frozen2, (rank, same2, frozen2)      # the poisoned step: nobody moves
        assert flag[0] == 1, flag               # the loss bit is up on BOTH ranks
