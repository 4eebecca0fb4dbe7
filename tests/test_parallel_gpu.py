"""Data parallelism on the GPU with TWO ranks (north_star: gradient all-reduce overlapped with backward).
Both ranks share the one MI355X of the test box, so the process group is `gloo` over device tensors (RCCL refuses two
ranks on one device); everything else is the production path: the traced backward list with its three lanes, the
bucket markers fired on the lane that finishes a bucket's slab reduction, the asynchronous all-reduce per bucket, the
1/world scale in the publish kernel.  Checked: every rank ends with the SAME gradient, equal to the mean of the two
shards' single-process gradients, for an hourglass and a ResNet model (the latter has one bucket and an out-of-arena
`out_fc` whose gradient is averaged by a post-accumulate hook)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, base, strat, size, q):
    import torch.distributed as dist
    from dsnt.model import build_mpii_pose_model
    from dsnt import parallel, synthetic
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dev = torch.device('cuda:0')
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
    procs = [ctx.Process(target=_worker, args=(r, 2, port, base, strat, size, q)) for r in range(2)]
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
