"""Flat multi-tensor optimisers over the parameter arena (reference `bin/train.py:314-326`).

`RMSprop(lr=2.5e-4)` (torch defaults alpha=0.99, eps=1e-8) and `SGD(momentum=0.9)` as ONE
kernel launch over the whole arena instead of one launch chain per parameter tensor (hg2 has 396
tensors).  Both subclass `torch.optim.Optimizer`, so `param_groups` / lr schedulers /
`zero_grad()` / `state_dict()` behave as callers expect.  If the gradients are not the arena's
own views (someone replaced p.grad) they are gathered into the arena first.
"""
import torch

from . import _lib
from ._lib import ptr, call


def _find_arena(model):
    for m in model.modules():
        r = m.__dict__.get('_tape_runner')
        if r is not None and r.arena is not None:
            return r
    raise RuntimeError('dsnt.optim: no parameter arena yet — move the model to the GPU and run one '
                       'forward pass (or call model.hg._runner().ensure(device); ResNet models: model._runner()) before building '
                       'the optimiser')


class _FlatOptimizer(torch.optim.Optimizer):
    def __init__(self, model, defaults):
        self.runner = _find_arena(model)
        arena = self.runner.arena
        arena_params = {id(p) for _, p, _, _ in arena.slots}
        extra = [p for p in model.parameters() if id(p) not in arena_params]
        if extra:
            raise RuntimeError('dsnt.optim: %d parameters live outside the arena' % len(extra))
        super().__init__([p for _, p, _, _ in arena.slots], defaults)
        self.grad_scale = 1.0
        self._steps = 0

    def _gather_grads(self):
        arena = self.runner.arena
        for name, p, o, n in arena.slots:
            gv = arena.gviews[name]
            if p.grad is None:
                gv.zero_()
            elif p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=set_to_none)


class RMSprop(_FlatOptimizer):
    def __init__(self, model, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0):
        super().__init__(model, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay))
        self.square_avg = torch.zeros_like(self.runner.arena.params)

    @torch.no_grad()
    def step(self, closure=None):
        arena = self.runner.arena
        self._gather_grads()
        g = self.param_groups[0]
        call('dsnt_rmsprop_step', ptr(arena.params), ptr(arena.grads), ptr(self.square_avg),
             arena.numel, float(g['lr']), float(g['alpha']), float(g['eps']),
             float(g['weight_decay']), float(self.grad_scale))
        self._steps += 1


class SGD(_FlatOptimizer):
    def __init__(self, model, lr=1e-3, momentum=0.0, weight_decay=0.0):
        super().__init__(model, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.momentum_buf = torch.zeros_like(self.runner.arena.params)

    @torch.no_grad()
    def step(self, closure=None):
        arena = self.runner.arena
        self._gather_grads()
        g = self.param_groups[0]
        call('dsnt_sgd_step', ptr(arena.params), ptr(arena.grads), ptr(self.momentum_buf),
             arena.numel, float(g['lr']), float(g['momentum']), float(g['weight_decay']),
             float(self.grad_scale), 1 if self._steps == 0 else 0)
        self._steps += 1
