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
handful of large buckets is the right granularity.

Works with any `torch.distributed` backend: "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests
(which exercise this host logic on flat CPU tensors).
"""
import torch
import torch.distributed as dist


class GradientAllReducer:
    """Bucketed asynchronous sum-all-reduce over contiguous ranges of one flat tensor."""

    def __init__(self, flat, bounds, group=None):
        self.flat = flat
        self.bounds = list(bounds)          # [(start, end)] element ranges, one per bucket
        self.group = group
        self.pending = []
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def bucket_ready(self, k):
        if self.world == 1:
            return
        s, e = self.bounds[k]
        if e > s:
            self.pending.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM,
                                                group=self.group, async_op=True))

    def reduce_all(self):
        for k in range(len(self.bounds)):
            self.bucket_ready(k)

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []


def broadcast_flat(flat, src=0, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)


class DataParallel:
    """Attach to a built, on-device dsnt model: replicates rank 0's weights and arranges the
    overlapped gradient all-reduce; the 1/world averaging is folded into the kernel that
    publishes the gradients, so stock and dsnt.optim optimisers both see the mean gradient."""

    def __init__(self, model, optimizer=None, group=None):
        from .optim import _find_arena
        self.runner = _find_arena(model)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        arena = self.runner.arena
        broadcast_flat(arena.params, 0, group)
        for b in model.buffers():
            if b.dtype.is_floating_point:
                broadcast_flat(b.data, 0, group)
        self.reducer = GradientAllReducer(arena.fresh, arena.bucket_bounds, group)
        self.runner.bucket_hook = self.reducer.bucket_ready
        self.runner.before_publish = self.reducer.wait
        # gradients are published as the MEAN over ranks (works with any optimiser)
        arena.publish_scale = 1.0 / self.world

    def shard(self, *tensors):
        """This rank's contiguous slice of a global batch (dim 0)."""
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        out = []
        for t in tensors:
            per = t.shape[0] // self.world
            out.append(t[rank * per:(rank + 1) * per])
        return out if len(out) > 1 else out[0]
