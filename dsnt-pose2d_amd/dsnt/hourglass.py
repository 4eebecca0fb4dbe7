"""Stacked-Hourglass backbone on the MI355X launch-list engine.

Same constructor signatures, attribute names and `state_dict()` keys as
`/root/reference/src/dsnt/hourglass.py` (`Bottleneck` :14-50, `Hourglass` :53-93,
`HourglassNet` :96-177), so checkpoints and callers (`build_mpii_pose_model`,
`convert_hg_model.py`) are interchangeable.  The `torch.nn` sub-modules here are parameter
holders only (they give the reference's initialisation, shapes and key names); `forward` never
calls them.  Instead the module tree is traced into forward/backward kernel launch lists
(`engine.Tape`) over NHWC buffers, parameters live OHWI-packed in one flat arena (the
`nn.Parameter`s become strided views of it with their logical OIHW shapes), and gradients land
in a matching flat arena that the fused optimiser and the RCCL all-reduce consume directly.
"""
import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from .engine import Tape, ConvParams, BnParams

EXPANSION = 2


def _ceil4(n):
    return (n + 3) // 4 * 4


# ====================================================================== parameter arena
class Arena:
    """Flat fp32 storage for every parameter of a module tree (+ a same-shaped gradient arena).

    Conv weights are stored OHWI with Cin padded to a multiple of 4 (only the 3-channel stem
    needs padding); each `nn.Parameter.data` is re-pointed to a strided view with the logical
    OIHW shape, so `state_dict()`, `load_state_dict()` and stock optimisers keep working.
    """

    def __init__(self, module, device):
        self.device = device
        self.publish_scale = 1.0    # 1/world_size under data parallelism
        slots, off = [], 0
        # parameters the traced launch lists never touch (e.g. the 'fc' strategy's out_fc of a ResNet pose
        # model, whose root module is the pose model itself) stay ordinary torch parameters
        skip = tuple(getattr(module, 'tape_exclude', ()))
        named = [(n, p) for n, p in module.named_parameters() if not (skip and n.startswith(skip))]
        bucket_of = getattr(module, 'param_bucket', lambda name: 0)
        order = sorted(range(len(named)), key=lambda i: (bucket_of(named[i][0]), i))
        nb = 1 + max([bucket_of(n) for n, _ in named] + [0])
        starts = [None] * nb
        for i in order:
            name, p = named[i]
            if p.dim() == 4:
                O, I, R, S = p.shape
                n = O * R * S * _ceil4(I)
            else:
                n = p.numel()
            b = bucket_of(name)
            if starts[b] is None:
                starts[b] = off
            slots.append((name, p, off, n))
            off += _ceil4(n)
        self.numel = off
        # contiguous element range of every gradient bucket (for the overlapped all-reduce)
        self.bucket_bounds = []
        for b in range(nb):
            s_ = starts[b] if starts[b] is not None else off
            nxt = [x for x in starts[b + 1:] if x is not None]
            self.bucket_bounds.append((s_, nxt[0] if nxt else off))
        self.params = torch.zeros(off, device=device, dtype=torch.float32)
        self.grads = torch.zeros(off, device=device, dtype=torch.float32)   # what p.grad views
        self.fresh = torch.zeros(off, device=device, dtype=torch.float32)   # what backward writes
        # bf16x6 path: the arena split into three bf16 planes, refreshed by ONE launch per step
        self.planes = torch.zeros(3 * off, device=device, dtype=torch.bfloat16)
        self.packed_planes = {}
        # fp16x3 path: two fp16 planes of every conv weight (scaled by a power of two per conv) + the weight bounds
        self.planes16 = torch.zeros(2 * off, device=device, dtype=torch.float16)
        self.wbounds = torch.zeros(64 * len(slots), device=device, dtype=torch.float32)   # 64 slots per bound
        self.packed_planes16, self.packed_wbound = {}, {}
        self.slots = slots
        self.pviews, self.gviews, self.packed, self.packed_fresh = {}, {}, {}, {}
        with torch.no_grad():
            for name, p, o, n in slots:
                src = p.detach().to(device=device, dtype=torch.float32)
                if p.dim() == 4:
                    O, I, R, S = p.shape
                    Ip = _ceil4(I)
                    store = self.params[o:o + n].view(O, R, S, Ip)
                    store[..., :I].copy_(src.permute(0, 2, 3, 1))
                    self.packed[name] = store
                    self.packed_planes[name] = self.planes[o:o + n]
                    self.packed_planes16[name] = self.planes16[o:o + n]
                    k = len(self.packed_wbound)
                    self.packed_wbound[name] = self.wbounds[64 * k:64 * k + 64]
                    self.packed_fresh[name] = self.fresh[o:o + n].view(O, R, S, Ip)
                    pv = store.permute(0, 3, 1, 2)[:, :I]
                    gv = self.grads[o:o + n].view(O, R, S, Ip).permute(0, 3, 1, 2)[:, :I]
                else:
                    store = self.params[o:o + n].view(p.shape)
                    store.copy_(src)
                    self.packed[name] = store
                    self.packed_fresh[name] = self.fresh[o:o + n].view(p.shape)
                    pv, gv = store, self.grads[o:o + n].view(p.shape)
                p.data = pv
                self.pviews[name], self.gviews[name] = pv, gv
                if p.grad is not None:          # an arena rebuilt under live gradients keeps them (as its own views)
                    gv.copy_(p.grad.to(device=device, dtype=torch.float32))
                    p.grad = gv
        # running statistics: plain contiguous float buffers on the device
        for name, b in module.named_buffers():
            if b.dtype.is_floating_point and (b.device != device or b.dtype != torch.float32
                                              or not b.is_contiguous()):
                b.data = b.detach().to(device=device, dtype=torch.float32).contiguous()
        self.ptr_all = [(p, p.data_ptr()) for _, p, _, _ in slots]
        self.ptr_check = self.ptr_all[:1] + self.ptr_all[-1:]
        self._pg = [(p, self.gviews[name]) for name, p, _, _ in slots]
        self._by_name = {name: (p, o, n) for name, p, o, n in slots}

    def valid(self, full=False):
        """The parameters still live in this arena (first and last slot; `full`: every slot)."""
        return all(p.data_ptr() == ptr and p.device == self.device
                   for p, ptr in (self.ptr_all if full else self.ptr_check))

    def logical(self, flat, name):
        """View of parameter `name` inside any flat buffer laid out like the arena, in the parameter's logical
        (OIHW) shape — optimiser state, gradients."""
        p, o, n = self._by_name[name]
        if p.dim() == 4:
            O, I, R, S = p.shape
            return flat[o:o + n].view(O, R, S, _ceil4(I)).permute(0, 3, 1, 2)[:, :I]
        return flat[o:o + n].view(p.shape)

    def grads_published(self):
        """Every p.grad is this arena's own gradient view (what publish_grads leaves behind)."""
        return all(p.grad is gv for p, gv in self._pg)

    def publish_grads(self, params):
        """Make p.grad reflect this backward (torch semantics: accumulate unless p.grad is None)."""
        state = [p.grad is None for p in params]
        a = float(self.publish_scale)
        if all(state):
            _lib.call('dsnt_axpy', _lib.ptr(self.fresh), _lib.ptr(self.grads), a, 0, self.numel)
            for (name, p, _, _) in self.slots:
                p.grad = self.gviews[name]
        elif not any(state) and all(p.grad.data_ptr() == self.gviews[name].data_ptr()
                                    for (name, p, _, _) in self.slots):
            _lib.call('dsnt_axpy', _lib.ptr(self.fresh), _lib.ptr(self.grads), a, 1, self.numel)
        else:
            for (name, p, o, n) in self.slots:
                fresh = self.fresh[o:o + n] * a
                if p.dim() == 4:
                    O, I, R, S = p.shape
                    fresh = fresh.view(O, R, S, _ceil4(I)).permute(0, 3, 1, 2)[:, :I]
                else:
                    fresh = fresh.view(p.shape)
                if p.grad is None:
                    self.gviews[name].copy_(fresh)
                    p.grad = self.gviews[name]
                else:
                    p.grad.add_(fresh)


# ====================================================================== traced programs
class Program:
    """Forward/backward launch lists for one (input shape, BatchNorm mode, backward wanted or not)."""

    def __init__(self, root, arena, in_shape, training, input_grad, record=None):
        self.training = training
        # a backward list is built in train mode, and — on request — for an eval-mode forward under autograd (frozen BatchNorm
        # statistics; the reference's autograd allows it: model.py:273-307 has no mode check)
        self.record = record = training if record is None else bool(record)
        self.token = 0
        self.in_flight = False      # a forward of this program is waiting for its backward (hourglass._Run)
        self.owner = None           # ... and this is that forward's token
        dev = arena.device
        tape = Tape(dev, training, record)
        tape.want_input_grad = bool(input_grad and record)     # the stem then keeps its data gradient (no space-to-depth form)
        tape.param_arena = arena.params
        self.tape = tape
        names = {id(m): n for n, m in root.named_modules()}

        def key(m, leaf):
            prefix = names[id(m)]
            return (prefix + '.' if prefix else '') + leaf

        class P:   # parameter lookup handed to the trace functions
            @staticmethod
            def conv(m):
                w = arena.packed[key(m, 'weight')]
                gw = arena.packed_fresh[key(m, 'weight')]
                b = arena.packed[key(m, 'bias')] if m.bias is not None else None
                gb = arena.packed_fresh[key(m, 'bias')] if m.bias is not None else None
                cp = ConvParams(w, b, gw, gb, m.stride[0], m.padding[0], m.dilation[0])
                cp.wq, cp.wq_stride = arena.packed_planes[key(m, 'weight')], arena.numel
                cp.wq16, cp.wb = arena.packed_planes16[key(m, 'weight')], arena.packed_wbound[key(m, 'weight')]
                return cp

            _bn = {}

            @staticmethod
            def bn(m):
                if id(m) not in P._bn:
                    P._bn[id(m)] = BnParams(arena.packed[key(m, 'weight')], arena.packed[key(m, 'bias')],
                                            arena.packed_fresh[key(m, 'weight')],
                                            arena.packed_fresh[key(m, 'bias')],
                                            m.running_mean, m.running_var,
                                            0.1 if m.momentum is None else m.momentum, m.eps)
                return P._bn[id(m)]
        P._bn = {}

        N, Cc, H, W = in_shape
        prep_pos = len(tape.fwd)
        if tape.use_bf16x6:
            tape.f('dsnt_split_bf16x3', arena.params, arena.planes, arena.numel)
        prep_head = tuple(tape.fwd[prep_pos:])
        self.in_nchw = tape.empty(N, Cc, H, W)
        x = tape.from_planar(self.in_nchw, _ceil4(Cc), 'input')
        self.in_act = x
        outs = root.trace(tape, x, P)
        single = not isinstance(outs, (list, tuple))
        self.single = single
        outs = [outs] if single else list(outs)
        self.out_channels = root.out_channels
        self.outs, self.gins = [], []
        # several outputs of one shape (the stacks' heat-maps) live in ONE slab [S, N, C, H, W], and so do their incoming
        # gradients: the model surface hands out one clone + views, and backward takes one gradient tensor — the heads and the
        # loss of all stacks are then one launch each instead of one per stack (eight stacks: ~60 small launches and as many
        # host-bound autograd nodes between the two launch lists, GPU idle; tools/head_section.py)
        self.out_slab = self.gin_slab = None
        if not single and len(outs) > 1 and len({(o.N, o.H, o.W) for o in outs}) == 1 and os.environ.get('DSNT_STACKED_OUTPUTS', '1') != '0':
            o = outs[0]
            self.out_slab = tape.empty(len(outs), o.N, self.out_channels, o.H, o.W)
            self.gin_slab = tape.empty(len(outs), o.N, self.out_channels, o.H, o.W) if record else None
        for i, o in enumerate(outs):
            t, g = tape.to_planar(o, self.out_channels, out=None if self.out_slab is None else self.out_slab[i],
                                  gin=None if self.gin_slab is None else self.gin_slab[i])
            self.outs.append(t)
            self.gins.append(g)
        self.input_grad = input_grad and record
        if record:
            tape.finish()
        # weight planes, operand bounds, data-gradient weights: before the first convolution that reads them
        tape.emit_f16_prep(prep_pos, prep_head)
        if record:
            if self.input_grad:
                self.gx = tape.empty(N, Cc, H, W)
                if x.grad is None:
                    raise RuntimeError('input gradient requested but nothing produced it')
                tape.b('dsnt_nhwc_to_nchw', x.grad, self.gx, N, Cc, H * W, x.C)
        self.n_fwd = sum(1 for e in tape.fwd if e[0] is not None)
        self.n_bwd = sum(1 for e in tape.bwd if e[0] is not None)


class StackedOutputs(list):
    """The outputs of a model with several same-shape outputs: a list of per-output tensors (views of `stacked`, what the
    reference's `model(x)` returns — model.py:299-307 appends one tensor per stack), plus the slab [S, N, C, H, W] itself for
    consumers that work on all of them at once (dsnt.model's fused heads)."""

    def __init__(self, stacked):
        super().__init__(stacked.unbind(0))
        self.stacked = stacked


class _Run(Function):
    @staticmethod
    def forward(ctx, runner, prog, x, *params):
        import weakref
        prog.in_nchw.copy_(x)
        prog.token += 1
        prog.tape.run(prog.tape.fwd, probe=runner.probe)
        ctx.runner, ctx.prog, ctx.token, ctx.nparams = runner, prog, prog.token, len(params)
        # this program's saved activations belong to this forward until its backward has run — or until autograd drops the
        # graph (the caller let go of the outputs): a second forward of the same shape meanwhile takes another program
        prog.in_flight = True
        token = prog.owner = prog.token

        def release(p=prog, t=token):
            if p.owner == t:            # (not p.token: a forward that does not record bumps the token without taking ownership)
                p.in_flight, p.owner = False, None
        weakref.finalize(ctx, release)
        if prog.out_slab is not None:
            return (prog.out_slab.clone(),)
        return tuple(o.clone() for o in prog.outs)

    @staticmethod
    def backward(ctx, *gouts):
        prog, runner = ctx.prog, ctx.runner
        if not prog.record:
            raise RuntimeError('dsnt: this forward was traced without a backward list')
        if prog.token != ctx.token:
            if prog.owner == ctx.token:        # this forward's activations are gone: nothing is waiting for them any more
                prog.in_flight, prog.owner = False, None
            raise RuntimeError('dsnt: the saved activations of this forward were overwritten by a '
                               'later forward of the same shape; run backward before forwarding again')
        if prog.gin_slab is not None:
            if gouts[0] is None:
                prog.gin_slab.zero_()
            else:
                prog.gin_slab.copy_(gouts[0])
        else:
            for g, gin in zip(gouts, prog.gins):
                if g is None:
                    gin.zero_()
                else:
                    gin.copy_(g)
        prog.tape.run(prog.tape.bwd, runner.bucket_hook, probe=runner.probe)
        if runner.before_publish is not None:
            runner.before_publish()
        runner.arena.publish_grads(runner.params)
        prog.in_flight = False                 # (a second backward through the same graph re-reads the same activations: allowed)
        prog.owner = None
        gx = prog.gx.clone() if prog.input_grad else None
        return (None, None, gx) + (None,) * ctx.nparams


class Runner:
    """Owns the arena and the traced programs of one root module."""
    MAX_IN_FLIGHT = 4       # forwards of one shape that may wait for their backward at the same time

    def __init__(self, root):
        self.root = root
        self.arena = None
        self.programs = {}
        self.params = []
        self._bns = None
        self.probe = None             # diagnostics: called with the list entry before every launch
        self.bucket_hook = None       # called with k when gradient bucket k is complete (DP)
        self.before_publish = None    # called after the backward list (DP: wait for all-reduce)

    def ensure(self, device):
        if self.arena is None or not self.arena.valid() or self.arena.device != device:
            self.arena = Arena(self.root, device)
            self.programs = {}
            self.params = [p for _, p, _, _ in self.arena.slots]

    def _bn_signature(self):
        # momentum / eps of EVERY BatchNorm are baked into the launch lists; re-trace if a caller changes one
        if self._bns is None:
            self._bns = [m for m in self.root.modules() if isinstance(m, nn.BatchNorm2d)]
        return tuple((m.momentum, m.eps) for m in self._bns)

    def __call__(self, x):
        if not x.is_cuda:
            raise RuntimeError('dsnt: input is on %s; the backbone runs on the HIP device only '
                               '(no CPU fallback) — move the model and input with .cuda()' % x.device)
        if x.dtype != torch.float32:
            raise RuntimeError('dsnt: expected a float32 input, got %s' % x.dtype)
        self.ensure(x.device)
        training = self.root.training
        grad_mode = torch.is_grad_enabled()
        # an eval-mode forward under autograd gets a program with a backward list of its own (BatchNorm on its running statistics,
        # their backward with frozen statistics); eval mode under no_grad — inference.py:33-48 — keeps the forward-only program
        record = grad_mode and (training or x.requires_grad or any(p.requires_grad for p in self.params))
        key = (tuple(x.shape), training, bool(x.requires_grad), self._bn_signature()) + (('eval+backward',) if (record and not training) else ())
        prog = self.programs.get(key)
        if prog is not None and prog.in_flight and (record or training):
            # the reference's autograd lets a caller run several forwards before the first backward (model.py:273-307 has no
            # restriction): each such forward needs its own set of saved activations, i.e. a further traced program of the same
            # shape (static buffers: ~190 MB per image for hg2) — created on demand, kept, at most MAX_IN_FLIGHT of them.  A forward
            # that records nothing (train mode under no_grad) takes another program as well instead of overwriting the
            # activations a pending backward still needs
            k = 1
            while (key, k) in self.programs and self.programs[(key, k)].in_flight:
                k += 1
            if k >= self.MAX_IN_FLIGHT:
                raise RuntimeError('dsnt: %d forward passes of this shape are waiting for their backward; run backward (or drop the '
                                   'outputs) before forwarding again' % k)
            key = (key, k)
            prog = self.programs.get(key)
        if prog is None:
            if x.requires_grad and not self.root.supports_input_grad:
                raise NotImplementedError('dsnt: gradient with respect to the input image is not '
                                          'on the hot path (stride-2 stem data-gradient)')
            prog = Program(self.root, self.arena, tuple(x.shape), training, bool(x.requires_grad), record=record or training)
            self.programs[key] = prog
        if record:
            outs = _Run.apply(self, prog, x, *self.params)
        else:
            with torch.no_grad():
                prog.in_nchw.copy_(x)
                prog.token += 1
                prog.tape.run(prog.tape.fwd, probe=self.probe)
                outs = (prog.out_slab.clone(),) if prog.out_slab is not None else tuple(o.clone() for o in prog.outs)
        if prog.out_slab is not None:
            return StackedOutputs(outs[0])
        return outs[0] if prog.single else list(outs)


class TapeModule(nn.Module):
    """Mixin: `forward` = traced launch lists; `.cuda()`/`.to()` invalidate the arena."""
    supports_input_grad = True

    def _runner(self):
        r = self.__dict__.get('_tape_runner')
        if r is None:
            r = Runner(self)
            self.__dict__['_tape_runner'] = r
        return r

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        r = self.__dict__.get('_tape_runner')
        # a no-op .cuda() / .float() (inference.generate_predictions calls model.cuda() every time) keeps the
        # arena and the traced programs; anything that really moved or retyped a parameter drops them
        if r is not None and r.arena is not None and not r.arena.valid(full=True):
            r.arena = None
            r.programs = {}
        return out

    def forward(self, x):
        return self._runner()(x)

    @property
    def arena(self):
        return self._runner().arena


# ====================================================================== the modules
class Bottleneck(TapeModule):
    """Pre-activation residual unit (reference hourglass.py:14-50)."""
    expansion = EXPANSION

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(inplanes)
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=True)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=True)
        self.bn3 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 2, kernel_size=1, bias=True)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.out_channels = planes * 2

    def trace(self, t, x, P):
        skip = x
        if self.downsample is not None:
            skip = t.conv(x, P.conv(self.downsample[0]), name='ds')
        y = t.conv(t.norm(x, P.bn(self.bn1)), P.conv(self.conv1), want_stats=True, name='c1')
        y = t.conv(t.norm(y, P.bn(self.bn2)), P.conv(self.conv2), want_stats=True, name='c2')
        return t.conv(t.norm(y, P.bn(self.bn3)), P.conv(self.conv3), res1=skip, want_stats=True,
                      name='c3')


def _trace_seq(seq, t, x, P):
    for m in seq:
        x = m.trace(t, x, P)
    return x


class Hourglass(TapeModule):
    """Depth-`depth` recursive hourglass (reference hourglass.py:53-93)."""

    def __init__(self, block, num_blocks, planes, depth):
        super().__init__()
        self.depth = depth
        self.block = block
        self.upsample = nn.Upsample(scale_factor=2, mode='nearest')
        levels = []
        for i in range(depth):
            groups = [nn.Sequential(*[block(planes * block.expansion, planes) for _ in range(num_blocks)])
                      for _ in range(4 if i == 0 else 3)]
            levels.append(nn.ModuleList(groups))
        self.hg = nn.ModuleList(levels)
        self.out_channels = planes * block.expansion

    def _level(self, n, t, x, P):
        g = self.hg[n - 1]
        # the full-resolution skip branch is independent of the whole low-resolution recursion:
        # trace it on the side lane so its few large kernels overlap the many small ones
        if t.record and t.use_lanes and t.fuse_join:
            # Training: the branch is traced LAST, so that its backward is emitted first and its gradient exists when
            # the pool's backward is emitted — the main lane then waits for the side lane, takes the branch's gradient
            # buffer over as x's and the pool's backward ACCUMULATES into it (no separate x.grad += branch.grad pass:
            # 400 MB at 64 x 64).  Launch order within a lane is unchanged.
            # (DSNT_SIDE_LANES side lanes, alternating by level: the inner levels' branches — which the main lane needs back first —
            # do not queue behind the outer level's large kernels)
            side = t.side_lanes[n % len(t.side_lanes)]
            xb = t.branch(x, join=False)
            t.sync(0, side, bwd=False)
            pooled = t.maxpool2(x)

            def take_branch_gradient():         # registered after the pool: runs right before the pool's backward
                t.sync_bwd(side, 0)
                if xb.grad is not None:
                    if x.grad is not None and x.pending_apply is None and t.share_grads:
                        # x holds a gradient already (the stack input of hourglass.py:175: the residual sum's came first) and the
                        # branch's arrived in a buffer of its own: the pool's backward, next on the list, adds it in its pass
                        t.add_later(x, xb.grad, xb.grad_amax)
                    else:
                        t.grad_identity(x, xb.grad, donate=True, g_amax=xb.grad_amax)
            t.on_backward(take_branch_gradient)
            low = _trace_seq(g[1], t, pooled, P)
            low = self._level(n - 1, t, low, P) if n > 1 else _trace_seq(g[3], t, low, P)
            low = _trace_seq(g[2], t, low, P)
            t.lane = side
            up1 = _trace_seq(g[0], t, xb, P)
            t.lane = 0
            t.sync(side, 0)
            return t.upsample2_add(up1, low)
        xb = t.branch(x)
        t.sync(0, 1)
        t.lane = 1 if t.use_lanes else 0
        up1 = _trace_seq(g[0], t, xb, P)
        t.lane = 0
        low = _trace_seq(g[1], t, t.maxpool2(x), P)
        low = self._level(n - 1, t, low, P) if n > 1 else _trace_seq(g[3], t, low, P)
        low = _trace_seq(g[2], t, low, P)
        t.sync(1, 0)
        return t.upsample2_add(up1, low)

    def trace(self, t, x, P):
        return self._level(self.depth, t, x, P)


class HourglassNet(TapeModule):
    """Hourglass model from Newell et al. ECCV 2016 (reference hourglass.py:96-177)."""
    supports_input_grad = True

    def __init__(self, block, num_stacks=2, num_blocks=4, num_classes=16):
        super().__init__()
        self.inplanes = 64
        self.num_feats = 128
        self.num_stacks = num_stacks
        self.num_classes = num_classes
        self.conv1 = nn.Conv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=True)
        self.bn1 = nn.BatchNorm2d(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._make_residual(block, self.inplanes, 1)
        self.layer2 = self._make_residual(block, self.inplanes, 1)
        self.layer3 = self._make_residual(block, self.num_feats, 1)
        self.maxpool = nn.MaxPool2d(2, stride=2)
        ch = self.num_feats * block.expansion
        hg, res, fc, score, fc_, score_ = [], [], [], [], [], []
        for i in range(num_stacks):
            hg.append(Hourglass(block, num_blocks, self.num_feats, 4))
            res.append(self._make_residual(block, self.num_feats, num_blocks))
            fc.append(nn.Sequential(nn.Conv2d(ch, ch, kernel_size=1, bias=True),
                                    nn.BatchNorm2d(ch), self.relu))
            score.append(nn.Conv2d(ch, num_classes, kernel_size=1, bias=True))
            if i < num_stacks - 1:
                fc_.append(nn.Conv2d(ch, ch, kernel_size=1, bias=True))
                score_.append(nn.Conv2d(num_classes, ch, kernel_size=1, bias=True))
        self.hg = nn.ModuleList(hg)
        self.res = nn.ModuleList(res)
        self.fc = nn.ModuleList(fc)
        self.score = nn.ModuleList(score)
        self.fc_ = nn.ModuleList(fc_)
        self.score_ = nn.ModuleList(score_)
        self.out_channels = num_classes

    def _make_residual(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion,
                                                 kernel_size=1, stride=stride, bias=True))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def param_bucket(self, name):
        """Gradient bucket of a parameter: 0 = stem, 1 + i = hourglass stack i."""
        parts = name.split('.')
        if parts[0] in ('hg', 'res', 'fc', 'score', 'fc_', 'score_'):
            return 1 + int(parts[1])
        return 0

    def trace(self, t, x, P):
        if x.H % 64 != 0 or x.W % 64 != 0:
            raise RuntimeError('dsnt: hourglass input must be a multiple of 64 pixels '
                               '(stem /4, four 2x2 poolings), got %dx%d' % (x.H, x.W))
        t.mark_bucket(0)
        stem = P.conv(self.conv1)
        # d loss / d image (the reference's autograd gives it; train.py never asks): the plain 7x7 / stride 2 convolution with its
        # zero-stuffed data gradient instead of the space-to-depth form
        gi = getattr(t, 'want_input_grad', False)
        x = (None if gi else t.stem_s2d(x, stem)) or t.conv(x, stem, want_stats=True, need_input_grad=gi, name='stem')
        x = t.bn_act(x, P.bn(self.bn1), relu=True, name='stem_act')
        x = _trace_seq(self.layer1, t, x, P)
        x = t.maxpool2(x)
        x = _trace_seq(self.layer3, t, _trace_seq(self.layer2, t, x, P), P)
        outs = []
        for i in range(self.num_stacks):
            t.mark_bucket(1 + i)
            y = self.hg[i].trace(t, x, P)
            y = _trace_seq(self.res[i], t, y, P)
            y = t.conv(y, P.conv(self.fc[i][0]), want_stats=True, name='fc')
            yn = t.norm(y, P.bn(self.fc[i][1]), relu=True)        # one BN, two consumers
            score = t.conv(yn, P.conv(self.score[i]), name='score')
            if i < self.num_stacks - 1:
                f = t.conv(yn, P.conv(self.fc_[i]), name='fc_')
                x = t.conv(score, P.conv(self.score_[i]), res1=x, res2=f, want_stats=True, name='score_')
            outs.append(score)
        return outs
