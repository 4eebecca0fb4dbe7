"""Flat multi-tensor optimisers over the parameter arena (reference `bin/train.py:314-326`).

`RMSprop(lr=2.5e-4)` (torch defaults alpha=0.99, eps=1e-8) and `SGD(momentum=0.9)` as ONE
kernel launch over the whole arena instead of one launch chain per parameter tensor (hg2 has 396
tensors).  Both subclass `torch.optim.Optimizer`, so `param_groups` / lr schedulers /
`zero_grad()` behave as callers expect, and `state_dict()` / `load_state_dict()` speak
`torch.optim.RMSprop` / `torch.optim.SGD`'s own format (per-parameter `square_avg` / `momentum_buffer`
in the logical OIHW shapes + `step`): the reference checkpoints `optimizer.state_dict()`
(`bin/train.py:364,492`), and such a checkpoint loads into the stock optimiser and back.
If the gradients are not the arena's own views (someone replaced p.grad) they are gathered into the
arena first.  Parameters the arena does not hold (the `fc` strategy's `out_fc`) are updated tensor by
tensor with the same arithmetic.
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
    STATE_KEY = None        # name of the per-parameter state tensor in torch's format

    def __init__(self, model, defaults, guard=None):
        self.guard = guard          # dsnt.guard.NanGuard: the update is skipped while its device flag is up
        self.runner = _find_arena(model)
        arena = self.runner.arena
        arena_names = {id(p): name for name, p, _, _ in arena.slots}
        self.extra = [p for p in model.parameters() if id(p) not in arena_names]
        # Parameter indices of param_groups / state_dict follow model.parameters() — the order a stock
        # torch.optim.RMSprop(model.parameters()) (bin/train.py:314-326) numbers them in — NOT the arena's slot order
        # (bucket by bucket: stem, stack 0, stack 1, ...), so that a checkpointed optimiser state lands on the same
        # parameters on either side.  _where[i] = ('arena', slot name) or ('extra', index into self.extra).
        extra_idx = {id(p): j for j, p in enumerate(self.extra)}
        ordered = list(model.parameters())
        self._where = [('arena', arena_names[id(p)]) if id(p) in arena_names else ('extra', extra_idx[id(p)])
                       for p in ordered]
        assert len({id(p) for p in ordered}) == len(arena.slots) + len(self.extra), 'arena holds a parameter the model does not'
        super().__init__(ordered, defaults)
        self.grad_scale = 1.0
        self._steps = 0
        self.flat_state = torch.zeros_like(arena.params)
        self.extra_state = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in self.extra]

    def _extra_keep(self, gr):
        """Element mask of an out-of-arena update under the guard (device-side, no host read): all False while the loss
        flag is up, False where the gradient is not finite — those also raise FLAG_GRAD, as the arena kernels do."""
        finite = torch.isfinite(gr)
        self.guard.flag[1:2].bitwise_or_((~finite).any().to(torch.int32) * 2)
        return finite & (self.guard.flag[0] == 0)

    def _state_tensor(self, i):
        kind, key = self._where[i]
        return self.runner.arena.logical(self.flat_state, key) if kind == 'arena' else self.extra_state[key]

    def _gather_grads(self):
        if self.guard is not None and self.guard.pre_update is not None:
            self.guard.pre_update()           # data-parallel: a check enqueued after backward() is exchanged before the update
        arena = self.runner.arena
        if arena.grads_published():           # the common case: every p.grad is the arena's own view
            return
        for name, p, o, n in arena.slots:
            gv = arena.gviews[name]
            if p.grad is None:
                gv.zero_()
            elif p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)

    # ------------------------------------------------------------------ torch-format state
    def state_dict(self):
        step = torch.tensor(float(self._steps))
        state = {}
        if self._has_state():
            for i in range(len(self._where)):
                state[i] = self._entry(self._state_tensor(i).clone(), step)
        groups = []
        for g in self.param_groups:
            pg = {k: v for k, v in g.items() if k != 'params'}
            pg['params'] = list(range(len(g['params'])))
            # which order the indices follow: 'model' = model.parameters() (torch.optim's own numbering).  Revisions before
            # round 3 numbered them in the arena's slot order (bucket by bucket) and wrote no marker: load_state_dict cannot
            # detect that permutation of equal-shaped parameters (hg stack 0 / stack 1) — it warns when a marker-less state
            # does not look like torch.optim's and takes `order='arena'` for such a checkpoint
            pg['dsnt_order'] = 'model'
            groups.append(pg)
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, state_dict, order=None):
        """order: 'model' / 'arena' overrides the state's own `dsnt_order` marker (see below); None = use the marker."""
        groups = state_dict['param_groups']
        if len(groups) != 1 or len(groups[0]['params']) != len(self._where):
            raise ValueError('dsnt.optim: loaded state dict has a different number of parameters')
        # Index order.  'model' (what this class writes) and NO marker (what torch.optim writes: bin/train.py:364,492 checkpoints
        # optimizer.state_dict()) both mean model.parameters() order.  A checkpoint of a dsnt.optim revision before round 3
        # numbered the parameters in ARENA order and carries no marker either — it cannot be told from torch's, so it has to be
        # re-tagged by hand (param_groups[0]['dsnt_order'] = 'arena', or order='arena') and is then re-numbered here; a marker-less
        # state that LOOKS like one (below) loads in model order with a warning; an unknown marker is refused.
        if order is None:
            order = groups[0].get('dsnt_order')
            if order is None:
                order = 'model'
                # What told the two apart before — keys of a modern torch.optim ('foreach', 'maximize', ...) — is absent from the
                # checkpoints of the reference's own torch 0.3 as well, and those are the ones most likely to arrive here; a warning
                # nudging THEM towards order='arena' would permute equal-shaped state.  What a pre-round-3 dsnt.optim state always
                # carried and torch.optim of any age never did: per-entry 'step' stored as a 0-d float TENSOR together with an
                # entry for EVERY parameter from step 0 on (torch 0.3: python int steps; torch >= 1.12: entries only for
                # parameters that had a gradient, plus 'foreach' & co. in the group)
                st0 = state_dict.get('state', {})
                looks_old_dsnt = bool(st0) and len(st0) == len(self._where) and all(
                    torch.is_tensor(e.get('step')) and e['step'].dim() == 0 and e['step'].dtype == torch.float32
                    for e in st0.values()) and not any(
                    k in groups[0] for k in ('foreach', 'maximize', 'capturable', 'differentiable'))
                if looks_old_dsnt:
                    import warnings
                    warnings.warn("dsnt.optim: the loaded state has no 'dsnt_order' marker and has the shape of a dsnt.optim state "
                                  "written before round 3 (arena order); taking its indices as model.parameters() order.  If it "
                                  "was written by such a revision, load it with load_state_dict(state, order='arena')",
                                  stacklevel=2)
        if order == 'arena':
            pos = {name: i for i, (name, _, _, _) in enumerate(self.runner.arena.slots)}
            st, remap = state_dict['state'], {}
            n_arena = len(pos)
            for i, (kind, key) in enumerate(self._where):
                j = pos[key] if kind == 'arena' else n_arena + key      # (arena order listed the out-of-arena parameters last)
                e = st.get(j, st.get(str(j)))
                if e is not None:
                    remap[i] = e
            state_dict = {'state': remap, 'param_groups': groups}
        elif order != 'model':
            raise ValueError("dsnt.optim: unknown parameter order %r in the loaded state (expected 'model' or 'arena')" % (order,))
        for k, v in groups[0].items():
            if k not in ('params', 'dsnt_order'):
                self.param_groups[0][k] = v
        st = state_dict['state']
        steps = 0
        for i in range(len(self._where)):
            e = st.get(i, st.get(str(i)))
            dst = self._state_tensor(i)
            if e is None or e.get(self.STATE_KEY) is None:
                dst.zero_()
                continue
            src = e[self.STATE_KEY]
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError('dsnt.optim: state of parameter %d has shape %s, the parameter %s' %
                                 (i, tuple(src.shape), tuple(dst.shape)))
            dst.copy_(src.to(dst.device, dst.dtype))
            steps = max(steps, int(float(e.get('step', 1))))
        self._steps = steps


class RMSprop(_FlatOptimizer):
    STATE_KEY = 'square_avg'

    def __init__(self, model, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0, guard=None):
        super().__init__(model, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay, momentum=0,
                                     centered=False), guard)

    @property
    def square_avg(self):
        return self.flat_state

    def _has_state(self):
        return self._steps > 0

    @staticmethod
    def _entry(t, step):
        return {'step': step.clone(), 'square_avg': t}

    @torch.no_grad()
    def step(self, closure=None):
        arena = self.runner.arena
        self._gather_grads()
        g = self.param_groups[0]
        args = (ptr(arena.params), ptr(arena.grads), ptr(self.flat_state), arena.numel, float(g['lr']),
                float(g['alpha']), float(g['eps']), float(g['weight_decay']), float(self.grad_scale))
        if self.guard is not None:
            call('dsnt_rmsprop_step_guarded', *args, ptr(self.guard.flag))
        else:
            call('dsnt_rmsprop_step', *args)
        for p, sq in zip(self.extra, self.extra_state):      # torch.optim.RMSprop's arithmetic, tensor by tensor
            if p.grad is None:
                continue
            gr = p.grad * self.grad_scale
            if g['weight_decay'] != 0:
                gr = gr.add(p, alpha=g['weight_decay'])
            if self.guard is not None:                       # as the guarded arena kernel: nothing while the loss flag is
                keep = self._extra_keep(gr)                  # up, non-finite gradient elements skipped (and flagged)
                sq.copy_(torch.where(keep, g['alpha'] * sq + (1 - g['alpha']) * gr * gr, sq))
                p.sub_(torch.where(keep, g['lr'] * gr / (sq.sqrt() + g['eps']), torch.zeros_like(gr)))
                continue
            sq.mul_(g['alpha']).addcmul_(gr, gr, value=1 - g['alpha'])
            p.addcdiv_(gr, sq.sqrt().add_(g['eps']), value=-g['lr'])
        self._steps += 1


class SGD(_FlatOptimizer):
    STATE_KEY = 'momentum_buffer'

    def __init__(self, model, lr=1e-3, momentum=0.0, weight_decay=0.0, guard=None):
        super().__init__(model, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=0,
                                     nesterov=False), guard)

    @property
    def momentum_buf(self):
        return self.flat_state

    def _has_state(self):
        return self._steps > 0 and self.param_groups[0]['momentum'] != 0

    @staticmethod
    def _entry(t, step):
        return {'momentum_buffer': t}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # torch's SGD keeps no step count: a loaded momentum buffer means "not the first step"
        self._steps = 1 if any(e.get('momentum_buffer') is not None for e in state_dict['state'].values()) else 0

    @torch.no_grad()
    def step(self, closure=None):
        arena = self.runner.arena
        self._gather_grads()
        g = self.param_groups[0]
        first = 1 if self._steps == 0 else 0
        args = (ptr(arena.params), ptr(arena.grads), ptr(self.flat_state), arena.numel, float(g['lr']),
                float(g['momentum']), float(g['weight_decay']), float(self.grad_scale), first)
        if self.guard is not None:
            call('dsnt_sgd_step_guarded', *args, ptr(self.guard.flag))
        else:
            call('dsnt_sgd_step', *args)
        for p, buf in zip(self.extra, self.extra_state):
            if p.grad is None:
                continue
            gr = p.grad * self.grad_scale
            if g['weight_decay'] != 0:
                gr = gr.add(p, alpha=g['weight_decay'])
            keep = self._extra_keep(gr) if self.guard is not None else None
            if g['momentum'] != 0:
                nb = gr if first else g['momentum'] * buf + gr
                buf.copy_(nb if keep is None else torch.where(keep, nb, buf))
                gr = buf
            p.sub_(g['lr'] * gr if keep is None else torch.where(keep, g['lr'] * gr, torch.zeros_like(gr)))
        self._steps += 1
