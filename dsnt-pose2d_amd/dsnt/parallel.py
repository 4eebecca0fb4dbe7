"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference has no parallelism at all (`bin/train.py:220` is a single `model.cuda()`); this is
the new capability BASELINE.json asks for.  Semantics: the minibatch is sharded
on dim 0, weights are replicated, BatchNorm uses per-replica batch statistics, gradients are
summed with an all-reduce over the flat gradient arena and divided by the world size before
the (identical) optimiser step on every rank.

The arena is laid out bucket-by-bucket in backward-completion order (stem last), and the traced
backward list carries a marker after the last launch that writes into each bucket: the
all-reduce of bucket k is enqueued (async, on RCCL's own stream) while the backward kernels of
bucket k-1 are still being launched — communication overlaps the rest of backward.  xGMI is
point-to-point (7 links x ~153 GB/s); hg2's 27 MB of gradients are ~0.3 ms of ring time, so a
handful of large buckets is the right granularity.  A bucket whose marker never fired (a
backbone traced without markers) is reduced before the gradients are published, so no model can
silently step on its local gradient.

Parameters outside the arena (the `fc` output strategy's `out_fc`) are broadcast at attach time
and their gradients averaged by a post-accumulate hook.

BatchNorm running statistics: every rank keeps the running statistics of
its own shard; they are buffers, not gradients, and are never exchanged during training.  A
checkpoint takes RANK 0's (what `model.state_dict()` returns there — the reference's
`train.py:488-496` saves from its single process), or, after `sync_running_stats()`, the mean
over ranks (mean of running means / running variances; `num_batches_tracked` is identical).

Works with any `torch.distributed` backend: "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests
(which exercise this host logic on flat CPU tensors).
"""
import torch
import torch.distributed as dist


class GradientAllReducer:
    """Bucketed asynchronous sum-all-reduce over contiguous ranges of one flat tensor."""

    def __init__(self, flat, bounds, group=None, flag=None):
        self.flat = flat
        self.bounds = list(bounds)          # [(start, end)] element ranges, one per bucket
        self.group = group
        # the non-finite guard's device flag (dsnt.guard.NanGuard.flag, int32): exchanged with MAX in `wait()` — after the last
        # bucket, before the gradients are published — so that EVERY rank takes the same skip decision in the optimiser kernels:
        # a NaN loss on one rank stops the whole job's update, as the reference's single process stops itself
        # (bin/train.py:360-371).  In wait(), not beside the first bucket: `guard.check(loss)` may be enqueued any time UP TO THE END
        # of `loss.backward()` (before it, as bin/train.py does, or from a hook during it) and is covered — the collective is ordered
        # after everything on the publishing stream at that point; it queues right behind the last bucket's all-reduce, one small
        # message.  A check enqueued AFTER backward() has returned stays rank-local for that step (one rank would skip, the others
        # update): DataParallel's optimiser hook exchanges the flag once more in front of the update for that case (`sync_flag`)
        self.flag = flag
        self.guard = None                   # (DataParallel: the NanGuard that owns `flag`, for `sync_flag`)
        self.flag_checks_seen = None        # guard.checks_enqueued at the last exchange
        self.exposed_ms = []                # diagnostics (bench.py): host time spent waiting for the collectives, per backward
        self.time_waits = False
        self.pending = []
        self.fired = set()                  # buckets enqueued since the last wait()
        self.last_late = []                 # buckets of the last backward that no marker announced (diagnostic)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def bucket_ready(self, k, late=False):
        if k in self.fired:
            return
        self.fired.add(k)
        if late:
            self.last_late.append(k)
        if self.world == 1:
            return
        s, e = self.bounds[k]
        if e > s:
            self.pending.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM,
                                                group=self.group, async_op=True))

    def reduce_all(self, late=False):
        for k in range(len(self.bounds)):
            self.bucket_ready(k, late)

    def wait(self):
        """Every bucket is reduced when this returns: the ones no marker announced are enqueued here."""
        self.last_late = []
        self.reduce_all(late=True)
        if self.world > 1 and self.flag is not None:
            self.pending.append(dist.all_reduce(self.flag, op=dist.ReduceOp.MAX, group=self.group, async_op=True))
            self.flag_checks_seen = self.guard.checks_enqueued if self.guard is not None else None
        if self.time_waits and self.pending and self.flat.is_cuda and self.stream_ordered():
            # exposed communication: what the publishing stream still has to wait for once backward has been enqueued
            # (device time between two events around the waits; read by `exposed_comm_ms` after a synchronisation)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for w in self.pending:
                w.wait()
            e1.record()
            self.exposed_ms.append((e0, e1))
        else:
            for w in self.pending:
                w.wait()
        self.pending = []
        self.fired = set()

    def sync_flag(self):
        """The optimiser's `pre_update` hook: if `guard.check` was called since the flag's exchange in wait() — i.e. AFTER backward()
        returned — exchange it once more, so that no rank updates while another skips.  The call counts are host-side and the same
        on every rank (one program), so every rank takes this branch together."""
        if self.world > 1 and self.flag is not None and self.guard is not None and \
                self.guard.checks_enqueued != self.flag_checks_seen:
            dist.all_reduce(self.flag, op=dist.ReduceOp.MAX, group=self.group)
            self.flag_checks_seen = self.guard.checks_enqueued

    def stream_ordered(self):
        """The backend's Work.wait() blocks the current STREAM (nccl = RCCL), not the host (gloo): only then do two events
        around the waits measure what the publishing stream was held up by."""
        return dist.is_initialized() and dist.get_backend(self.group) == 'nccl'

    def bucket_bytes(self):
        """Bytes each bucket's all-reduce moves per rank (fp32 gradient elements), in bucket order."""
        return [4 * (e - s) for s, e in self.bounds]

    def exposed_comm_ms(self):
        """Mean device time per backward that the publishing stream waited for the collectives (time_waits = True);
        None when nothing was timed — in particular under a host-blocking backend (gloo), where the figure would read ~0."""
        if not self.exposed_ms:
            return None
        torch.cuda.synchronize()
        v = [a.elapsed_time(b) for a, b in self.exposed_ms]
        self.exposed_ms = []
        return sum(v) / len(v)


def broadcast_flat(flat, src=0, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)


class DataParallel:
    """Attach to a built, on-device dsnt model: replicates rank 0's weights and arranges the
    overlapped gradient all-reduce; the 1/world averaging is folded into the kernel that
    publishes the gradients, so stock and dsnt.optim optimisers both see the mean gradient."""

    def __init__(self, model, optimizer=None, group=None):
        from .optim import _find_arena
        self.model = model
        self.runner = _find_arena(model)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        arena = self.runner.arena
        broadcast_flat(arena.params, 0, group)
        for b in model.buffers():
            if b.dtype.is_floating_point:
                broadcast_flat(b.data, 0, group)
        # (an optimiser built with guard= hands its flag over: the skip decision is then the same on every rank.  The flag is
        # exchanged once per backward, in the reducer's wait() right before the gradients are published: a `guard.check(loss)`
        # enqueued on the current stream at any point between the forward and the end of `loss.backward()` is covered; one
        # enqueued AFTER backward returns belongs to the next step's exchange)
        guard = getattr(optimizer, 'guard', None)
        self.reducer = GradientAllReducer(arena.fresh, arena.bucket_bounds, group,
                                          flag=guard.flag if guard is not None else None)
        if guard is not None:
            self.reducer.guard = guard
            guard.pre_update = self.reducer.sync_flag
        self.runner.bucket_hook = self.reducer.bucket_ready
        self.runner.before_publish = self.reducer.wait
        # gradients are published as the MEAN over ranks (works with any optimiser)
        arena.publish_scale = 1.0 / self.world
        # parameters the arena does not hold: same start on every rank, mean gradient after every backward
        in_arena = {id(p) for _, p, _, _ in arena.slots}
        self.extra = [p for p in model.parameters() if id(p) not in in_arena]
        self._hooks = []
        for p in self.extra:
            broadcast_flat(p.data, 0, group)
            if p.requires_grad:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._average_extra))

    def _average_extra(self, p):
        # earlier (already averaged, rank-identical) contributions in p.grad are a fixed point of the mean
        if self.world > 1:
            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group)
            p.grad.div_(self.world)

    def detach(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self.runner.bucket_hook = None
        self.runner.before_publish = None
        if self.reducer.guard is not None:
            self.reducer.guard.pre_update = None
        self.runner.arena.publish_scale = 1.0

    def sync_running_stats(self):
        """Replace every rank's BatchNorm running statistics by their mean over ranks (checkpoint time)."""
        if self.world == 1:
            return
        for name, b in self.model.named_buffers():
            if b.dtype.is_floating_point:
                dist.all_reduce(b.data, op=dist.ReduceOp.SUM, group=self.group)
                b.data.div_(self.world)

    def shard(self, *tensors):
        """This rank's contiguous slice of a global batch (dim 0)."""
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        out = []
        for t in tensors:
            if t.shape[0] % self.world != 0:
                raise ValueError('dsnt.parallel: global batch %d is not divisible by the world size %d '
                                 '(the tail would never be trained on)' % (t.shape[0], self.world))
            per = t.shape[0] // self.world
            out.append(t[rank * per:(rank + 1) * per])
        return out if len(out) > 1 else out[0]
