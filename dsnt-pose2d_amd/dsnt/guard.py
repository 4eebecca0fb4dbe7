"""Non-finite guard of the train step, without a device-to-host synchronisation.

The reference checks `np.isnan(loss.data[0])` right after `forward_loss` — a blocking D2H copy every step — and
on NaN dumps weights, optimiser state and inputs to `model_dump.pth` and raises
(`/root/reference/src/dsnt/bin/train.py:360-371`).  Here the check is a device-side flag:

* `check(loss)` enqueues a one-launch test of the loss scalar (any tensor works) that raises bit DSNT_FLAG_LOSS;
* an optimiser built with `guard=` (dsnt.optim) skips the WHOLE update (arena and out-of-arena parameters) while the
  loss flag is up: after a non-finite loss the weights and optimiser state are those before the step — what the
  reference's dump captures by raising before `optimizer.step()`;
* a non-finite GRADIENT behind a finite loss (e.g. an fp16x3 operand bound that was too small would surface as inf in
  one weight gradient) is handled element-wise: the update kernels skip exactly the non-finite elements and raise
  DSNT_FLAG_GRAD; every finite element IS updated from that backward pass.  (The reference has no such check at all:
  torch.optim would write the NaN into the weights.)  So after this error the weights are finite but they are NOT the
  pre-step weights; from the next step on the raised flag blocks every update until `reset()`;
* `poll()` reads the flag asynchronously: a non-blocking copy into pinned host memory whose event is examined on
  the NEXT poll, so a bad step is reported one step late and a good step never waits.  `sync()` is the blocking
  form (end of an epoch, tests).
"""
import torch

from ._lib import ptr, call

FLAG_LOSS, FLAG_GRAD = 1, 2


class NonFiniteError(FloatingPointError):
    pass


class NanGuard:
    def __init__(self, device):
        self.flag = torch.zeros(2, dtype=torch.int32, device=device)
        self.host = torch.zeros(2, dtype=torch.int32).pin_memory()
        self.event = None
        self.steps_checked = 0
        self.checks_enqueued = 0        # host count of check() calls (the same sequence on every rank of a data-parallel job)
        self.pre_update = None          # dsnt.parallel.DataParallel: exchanges the flag again if a check came after backward

    def check(self, loss):
        """Enqueue the non-finite test of `loss` (scalar or any fp32 tensor) on the current stream.
        Under dsnt.parallel.DataParallel call it before `loss.backward()` returns (the reference checks right after
        forward_loss: train.py:360): the flag is exchanged between the ranks once per backward, in the reducer's wait()
        just before the gradients are published; a check enqueued after that is exchanged once more by the optimiser's
        `pre_update` hook (a second small collective in front of the update — the price of checking late)."""
        self.checks_enqueued += 1
        x = loss.detach()
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        call('dsnt_nonfinite_flag', ptr(x), x.numel(), ptr(self.flag), FLAG_LOSS)
        return loss

    def _raise(self, v):
        what = []
        if v[0] & FLAG_LOSS:
            what.append('non-finite loss')
        if (v[0] | v[1]) & FLAG_GRAD:
            what.append('non-finite gradient')
        how = ('the optimiser skipped that whole update: weights and optimiser state are the pre-step ones '
               '(train.py:360-371 dumps them here)' if v[0] & FLAG_LOSS else
               'the optimiser skipped exactly the non-finite elements (every other element was updated) and updates '
               'nothing further until reset()')
        raise NonFiniteError('dsnt: %s detected on the device; %s' % (' and '.join(what), how))

    def poll(self):
        """Non-blocking: raise if an EARLIER poll's copy of the flag has landed and is non-zero."""
        if self.event is not None and self.event.query():
            v = self.host.tolist()
            self.event = None
            if v[0] or v[1]:
                self._raise(v)
        if self.event is None:
            self.host.copy_(self.flag, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        self.steps_checked += 1

    def sync(self):
        v = self.flag.tolist()          # blocking
        self.event = None
        if v[0] or v[1]:
            self._raise(v)

    def reset(self):
        self.flag.zero_()
        self.event = None
