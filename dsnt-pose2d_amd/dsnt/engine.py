"""Static launch-list engine for the hourglass backbone on MI355X.

The reference runs the backbone as ~300 separate PyTorch modules per pass
(`/root/reference/src/dsnt/hourglass.py:30-50,78-90,155-177`) and lets autograd record them
every step.  Here the module tree is traced ONCE per (batch shape, mode) into two flat lists of
C-ABI kernel launches (forward, backward) over pre-allocated NHWC buffers; a step is a replay
of those lists on the current HIP stream (no allocation, no Python graph building, capturable
into a hipGraph).  Fusions decided at trace time:

* BatchNorm+ReLU in front of a conv is folded into the conv's operand load (scale/shift per
  input channel); its batch statistics come from the *producer's* epilogue (per-tile column
  sums written by the previous conv), so a pre-activation Bottleneck is 3 conv launches + 3
  tiny finalise launches instead of 10 elementwise passes;
* residual / skip adds ride in the conv epilogue (up to two addends);
* in backward, identity branches donate the incoming gradient buffer instead of copying, and
  every kernel that produces a gradient can accumulate in place, so fan-in costs no extra pass.

Gradients with respect to parameters are written straight into one flat arena (the same arena
the fused optimiser and the RCCL all-reduce operate on).
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ConvGeom, BnBwdEpilogue, BnTail, BnPrologue, BnBwdApply

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class Act:
    """An activation on the tape: NHWC buffer + lazily created gradient / batch statistics."""
    __slots__ = ('buf', 'N', 'H', 'W', 'C', '_grad', 'stats', 'name', 'grad_amax', 'amax_tail', 'fold_ok',
                 'pending_apply', 'uses', 'gives_away', 'is_branch', 'grad_shared', 'pending_add', 'base', 'base_amax')

    def __init__(self, buf, name=''):
        self.buf = buf
        self.N, self.H, self.W, self.C = buf.shape
        self._grad = None       # torch tensor once some backward op has written it (`grad`)
        # a gradient this activation's CONTINUES without owning it (Tape.conv's backward, `defer_res`): the next writer writes
        # base + its value into a buffer of its own (`Tape.grad_target`), so that the base stays intact for a deferred reader
        self.base, self.base_amax = None, None
        self.stats = None       # (partial tensor, ntiles)
        self.amax_tail = None   # fp16x3: the dsnt_out_bounds of the producing launch if it can leave max|buf| (operand_amax)
        self.grad_amax = None   # fp16x3: device scalar max|grad| when ONE bn-backward apply wrote the whole gradient
        self.fold_ok = False    # the producing 1x1 convolution can take the BatchNorm backward of its consumer into its own backward
        self.pending_apply = None   # ... and this is that BatchNorm backward, reduced but not applied (Tape._norm_backward)
        self.uses = 0           # forward consumers (each contributes once to the gradient)
        self.gives_away = False     # the producer's backward DONATES this gradient's buffer onwards (residual inputs, upsample + add)
        self.is_branch = False  # `Tape.branch`: an alias of another activation (its gradient is joined to that one's)
        self.grad_shared = False    # .grad is another activation's buffer, read in place (Tape.share_grads): readers must not be deferred
        self.pending_add = None     # (gradient tensor, its bound slot) the NEXT writer of .grad adds in its own pass (Tape.add_later)
        self.name = name

    @property
    def M(self):
        return self.N * self.H * self.W

    @property
    def grad(self):
        if self._grad is None and self.base is not None:
            # read before any further writer came: the gradient IS the base, read in place (`base` stays set: a later writer
            # still goes out of place, `Tape.grad_target`)
            self._grad, self.grad_amax, self.grad_shared = self.base, self.base_amax, True
        return self._grad

    @grad.setter
    def grad(self, g):
        self._grad = g


class ConvParams:
    """Views into the parameter / gradient arenas for one convolution (weights are OHWI)."""
    __slots__ = ('w', 'b', 'gw', 'gb', 'Cout', 'R', 'S', 'Cin', 'stride', 'pad', 'dil', 'wq', 'wq_stride', 'wq16', 'wb',
                 'post_reduce')

    def __init__(self, w, b, gw, gb, stride=1, pad=0, dil=1):
        self.w, self.b, self.gw, self.gb = w, b, gw, gb
        self.wq, self.wq_stride = None, 0      # bf16x6 planes of w (plane 0 view, plane stride)
        self.wq16, self.wb = None, None        # fp16x3: fp16 planes (same stride) and the device scalar max|w|
        self.post_reduce = None                # (name, args) launched after the slab reduction that writes gw (stem_s2d)
        self.Cout, self.R, self.S, self.Cin = w.shape
        self.stride, self.pad, self.dil = stride, pad, dil


class BnParams:
    __slots__ = ('gamma', 'beta', 'ggamma', 'gbeta', 'rmean', 'rvar', 'C', 'uses', 'momentum', 'eps')

    def __init__(self, gamma, beta, ggamma, gbeta, rmean, rvar, momentum=BN_MOMENTUM, eps=BN_EPS):
        self.gamma, self.beta, self.ggamma, self.gbeta = gamma, beta, ggamma, gbeta
        self.rmean, self.rvar = rmean, rvar
        self.momentum, self.eps = float(momentum), float(eps)
        self.C = gamma.numel()
        self.uses = 0           # backward accumulates dgamma/dbeta from the second use on


class Normed:
    """x seen through a BatchNorm(+ReLU): what a fused conv prologue needs."""
    __slots__ = ('x', 'bn', 'mean', 'invstd', 'scale', 'shift', 'relu', 'abound', 'pending', 'lane')


class Tape:
    def __init__(self, device, training, record=None):
        self.device = device
        # `training` is the BatchNorm mode (batch statistics vs running statistics, nn.Module.train / eval); `record` says
        # whether a backward list is built.  They differ for a backward through an EVAL-mode forward (frozen BatchNorm
        # statistics: dx = scale dz, the two reductions only feed dgamma / dbeta) — rare (bin/train.py never does it), so that
        # program runs on the bound-free kernels: bf16x6, none of the fp16x3-only fused kernels
        self.training = training
        self.record = training if record is None else bool(record)
        self.fwd = []           # (cfunc, args)
        self.bwd = []
        self._bwd_emitters = []
        self._scratch = {}
        self._keep = []         # keeps ctypes structs / tensors alive
        self.lib = _lib.load()
        self.nbytes = 0
        self.bytes_fwd, self.bytes_bwd, self.bytes_by_name, self._ws_ptrs = 0, 0, {}, set()
        # fp32-accurate split-bf16 matrix-core path for the large convolutions (DSNT_MFMA=f32 disables)
        self.use_bf16x6 = os.environ.get('DSNT_MFMA', 'bf16x6') != 'f32'
        # (16384 -> 8192, the 16x16 level at batch 32: -0.08 ms; -> 4096 once fwd1 took the 1x1 convolutions of that size: the 16x16
        # level at batch 16, hg8 -0.08 ms, nothing between 4096 and 8192 rows at batch 32; 2048: +0.07 ... +0.09)
        self.bf16x6_min_rows = int(os.environ.get('DSNT_BF16X6_MIN_ROWS', '4096'))
        # lanes: 0 = the caller's stream, 1 = a side stream for independent branches (the full-resolution
        # skip branch of every hourglass level runs beside the low-resolution recursion)
        self.lane = 0
        self.use_lanes = os.environ.get('DSNT_LANES', '1') != '0'
        self.fuse_join = True      # hourglass.Hourglass._level: branch gradients joined inside the pool's backward (-0.15 ms)
        self.side_stream = None
        self.wgrad_stream = None
        # lanes 3.. : more side lanes (the skip branches of the hourglass levels alternate over them: an inner level's
        # branch, which the main lane needs back first, does not queue behind the outer level's large kernels)
        self.side_lanes = (1, 3)   # two side lanes (one: +0.13 ms, four: +0.35 ms: DESIGN.md "round 2")
        self.n_lanes = 3 + len(self.side_lanes) - 1
        self.chain_lanes = (0,) + self.side_lanes
        self.more_streams = ()
        # weight gradients feed nothing downstream in backward.  The large ones (>= DSNT_WGRAD_LANE_ROWS output rows; 0 =
        # all) go to a third lane: the chip then has work while the dependency chain walks the launch-bound
        # low-resolution levels, and the main lane neither runs nor waits for the slab reductions (-0.9 ms/step on hg2).
        # DSNT_WGRAD_LANE_RES=1 also moves convolutions with residual inputs, whose dY buffer is donated onwards and
        # written again — the writer then has to wait for the lane (measured: +0.45 ms, off).
        self.wgrad_lane = 2 if (self.use_lanes and os.environ.get('DSNT_WGRAD_LANE', '1') != '0') else None
        self.wgrad_lane_rows = int(os.environ.get('DSNT_X_WGRAD_LANE_ROWS', '16000'))
        self.wgrad_lane_res = False
        self.wgrad_lane_from = self.chain_lanes
        # 3x3 forward / data gradient: the symmetric persistent kernel of csrc/conv3s.hip (stream-ordered weight planes)
        self.conv3s = 'conv3s' not in os.environ.get('DSNT_OFF', '').replace('+', ',').split(',')       # DSNT_OFF=conv3s,gemm1,wgrad3,wgrad1
        # the whole backward of a 1x1 convolution in one launch (csrc/bwd1.hip): data gradient with the BatchNorm-backward
        # epilogue + weight gradient + (conv1 of a Bottleneck) the BatchNorm backward of the layer behind, each tensor read once
        self.bwd1 = 'bwd1' not in os.environ.get('DSNT_OFF', '').replace('+', ',').split(',')
        self.fwd1 = 'fwd1' not in os.environ.get('DSNT_OFF', '').replace('+', ',').split(',')      # ... and their forward (csrc/fwd1.hip)
        # round 5: the BatchNorm backward behind a 3x3 convolution folded into that convolution's data gradient (conv3s.hip MODE 4)
        self.fold3 = 'fold3' not in os.environ.get('DSNT_OFF', '').replace('+', ',').split(',')
        self.fold3_rows = int(os.environ.get('DSNT_X_FOLD3_ROWS', '16384'))
        self.stem4 = 'stem4' not in os.environ.get('DSNT_OFF', '').replace('+', ',').split(',')    # the stem's forward (csrc/stem4.hip)
        # round 6: runs of small dependent launches of one lane as ONE persistent launch (csrc/stage.h).  OFF by default — built,
        # bit-identical to the launches it replaces (tests/test_stage_gpu.py), and measured SLOWER: hg2 batch 32 11.06 -> 11.97 ms,
        # hg8 batch 16 24.1 -> 25.7 ms with every run fused; 11.17 / 24.04 (no gain) with only the <= 64-workgroup launches of the
        # 4 x 4 level inside (profiles/r06_stage_ab.txt).  DSNT_STAGE=1 turns it on (A/B, tests).
        self.stage = os.environ.get('DSNT_STAGE', '0') == '1'
        self.stage_min_run = int(os.environ.get('DSNT_X_STAGE_MIN_RUN', '3'))
        self.stage_max_vgrid = int(os.environ.get('DSNT_X_STAGE_MAX_VGRID', '512'))
        self.stage_grid = int(os.environ.get('DSNT_X_STAGE_GRID', '64'))
        self.stage_census = []      # (stage launches, recorded launches inside) per compiled list
        self._f16_w_stream = {}
        # DSNT_CONV_SHARE_CHIP on the side lanes' launches of the two persistent kernels (3x3: 3/2 workgroups per CU; 1x1: half of
        # the CUs): both hold most of a CU's LDS for the whole launch, and the chain's kernels need LDS too (-0.2 ms)
        self.conv_share = True
        self.wgrad_share = True    # DSNT_WGRAD_SHARE_CHIP on every launch of that lane (-0.2 ms)
        # DSNT_X=<name>=<value>,...: A/B overrides of scheduling constants (tools/ab_env.sh); not product switches
        self._x = dict(kv.split('=') for kv in os.environ.get('DSNT_X', '').split(',') if '=' in kv)
        self.wgrad_narrow = self._x.get('wgrad_narrow', '1')      # DSNT_WGRAD_NARROW: 0 never, 1 always, 2 stem bucket only, 3 stacks only
        # a gradient shared instead of copied (A/B: DSNT_X=share_grads=0): a second residual input with ONE consumer reads dL/dy in
        # place (hourglass.py:175: x + fc_ + score_ — fc_'s gradient IS the sum's): one 268 MB pass per intermediate supervision,
        # 7 per hg8 step.  (The OTHER pass there — x.grad += the skip branch's gradient, Hourglass._level — stays: both buffers
        # arrive by donation, and folding the sum into the branch's last apply would need a five-stream form of that kernel.)
        self.share_grads = self._x.get('share_grads', '1') != '0'
        # the weight gradient of a low-resolution convolution WITH residual inputs (conv3 of a Bottleneck: 2-32 workgroups, 17-25 us
        # on the dependency chain each; 15 per hg2 step, 63 per hg8 step) joins its bucket's grouped launch too: its dL/dy is not
        # donated to the residual input but becomes the `base` that input's gradient continues out of place (Act.base)
        self.defer_res = self._x.get('defer_res', '1') != '0'
        self.flush_points = self._x.get('flush_points', '1') != '0'       # `flush_point` (A/B)
        # the per-step preparation in two halves: what the FORWARD reads (arena planes, weight planes, BatchNorm bounds) and what only
        # the backward reads (re-packed + split data-gradient weights); the main lane waits for the first half only (A/B: prep_split=0)
        self.prep_split = self._x.get('prep_split', '1') != '0'
        self.cur_bucket = 0         # parameter bucket of the layers being traced (mark_bucket)
        self._wgrad_lane_reads = set()
        self._writer_base = None
        self._ident = None
        # ... and they are HELD BACK (launches collected, not yet on the list) until the chain enters a launch-bound
        # phase: a `release point` is the backward of an up-sampling whose low-resolution operand has at most
        # DSNT_WGRAD_RELEASE_ROWS rows.  Issued as they come, the weight gradients share the chip with the large
        # data-gradient kernels and are gone by the time the small levels start; held back, they run beside them.
        # Only while a release point is still ahead in the backward order (a ResNet has none: nothing is held back).
        # (8192 -> 4096 in round 6: at batch 32 the 16 x 16 level — 8192 rows — still fills the chip by itself; released one level
        # further in, the held weight gradients run beside the 8 x 8 / 4 x 4 chains only: hg2 -0.07 ms, hg8 batch 16 unchanged — its
        # 16 x 16 level has 4096 rows; profiles/r06_ab_switches.txt box A)
        self.release_rows = int(os.environ.get('DSNT_X_RELEASE_ROWS', '4096'))
        self._release_total, self._release_left = 0, 0
        self._held = []
        self.acts = []          # every activation in creation order (debugging / introspection)
        self.dgrad_slots = []   # (conv params, dst offset) of every conv whose data gradient is needed
        self.dgrad_total = 0
        self.dgrad_f32 = None
        self.dgrad_planes = None
        self.param_arena = None  # flat fp32 parameter arena (set by the Program) for the one-launch pack
        # weight-gradient slabs are kept per convolution and reduced once per parameter bucket (one launch
        # instead of one per convolution); DSNT_DEFER_REDUCE=0 reduces inside every dsnt_conv_wgrad call
        self.defer_reduce = os.environ.get('DSNT_DEFER_REDUCE', '1') != '0'
        self._pending_reduce = []   # table rows of the slabs written since the last flush
        # weight gradients of the low-resolution levels (few workgroups each, nothing downstream in backward
        # needs them) wait for the end of their parameter bucket and run side by side in one grouped launch;
        # DSNT_WGRAD_GROUP_ROWS = largest N*Ho*Wo that is deferred (0 disables)
        self.group_rows = int(os.environ.get('DSNT_X_GROUP_ROWS', '8192')) if self.defer_reduce else 0
        self._pending_group = []    # (descriptor bytes, workgroups) since the last flush
        # max-pool / upsample+add write the BatchNorm statistics of their output themselves (DSNT_FUSE_OP_STATS=0:
        # a separate dsnt_bn_stats pass when a BatchNorm asks for them)
        self.fuse_op_stats = True
        # BatchNorm finalisation of few-tile statistics inside the consumer's prologue (DSNT_FUSE_FINALIZE=0: separate launches)
        self.fuse_finalize = os.environ.get('DSNT_FUSE_FINALIZE', '1') != '0'
        # ... up to this many (tiles x channels) of partial sums.  Measured (tools/bench_bn_prologue.py, MI355X): a finalise
        # launch costs the chain 4-5 us; the prologue costs every workgroup of the consumer 2.8 us at 16 KB of partials,
        # 3.2 us at 32 KB, 4.5 us at 64 KB (no gain), 9 us at 128 KB (a loss): fused up to 32 KB
        # (round 4, with one statistics row per workgroup from the 1x1 kernels: hg2 is indifferent between 0 and 4096 — 12.00-12.03
        # ms — and hg8 at batch 16 is best at 2048: 26.64 vs 26.84 at 4096, 26.76 at 1024, 26.99 with separate launches only)
        self.fuse_finalize_max = int(os.environ.get('DSNT_X_FUSE_FINALIZE_MAX', '2048'))
        # fp16x3 (default; DSNT_SPLIT=bf16x6 turns it off): two fp16 planes + three MFMAs instead of three bf16 planes + six, where an operand
        # bound is available without a host round-trip: train-mode BN+ReLU operands (bound from the BN parameters) and
        # weights (amax in the per-step prep launch)
        self.use_f16x3 = self.use_bf16x6 and os.environ.get('DSNT_SPLIT', 'f16x3') == 'f16x3'
        if self.record and not self.training:
            self.use_f16x3 = False      # (its backward operand bounds come from BATCH statistics: |xhat| <= sqrt(M))
        self._f16_w_rows, self._f16_w_seen = [], set()
        self._f16_bn_rows = []
        self._eval_bn_rows = []     # eval mode: every BatchNorm's vectors come from ONE table-driven launch per forward
        self._amax_buf, self._amax_used = None, 0      # zeroed at the start of every backward
        self.raw_f16 = True        # bounds of raw operands from the producers' epilogues (-0.24 ms)
        self._planar_src, self._prep_exempt, self._post_reduce = {}, set(), []
        self.stem_s2d_on = True    # the stem as a space-to-depth convolution (-0.08 ms)
        self._famax_buf, self._famax_used, self._famax_of, self._famax_bn_of = None, 0, {}, {}   # forward activations: zeroed at the start of every forward
        # DSNT_AMAX_ALL=0: only single-writer BN-backward outputs get a bound (A/B switch); default: bounds follow the
        # gradient through every writer that can report one (apply, axpy, pool / upsample backward) and through donations
        self.amax_all = True
        self._f16_dw_rows = []      # fp16x3 planes of the re-packed data-gradient weights
        self._dgrad_pack = None     # arguments of the one re-packing launch (set by finish, emitted by emit_f16_prep)
        self.prep_on_side_lane = self.use_lanes      # per-step preparation beside the stem (-0.15 ms)
        self.dgrad_planes16, self.dgrad_bounds = None, None
        # every fp16x3 launch with the tensors behind its operands and bounds: (list entry, {...}) — lets a test walk a
        # step launch by launch and hold each bound against the operand it must dominate (tests/test_bounds_gpu.py)
        self.f16_uses = []
        self._pending_group_uses = []
        # replay from C (dsnt_list_*): a launch list is recorded once into the library and then issued by ONE call per
        # segment instead of one ctypes call per launch (DSNT_C_REPLAY=0: the Python loop below)
        self.c_replay = True
        self._clists = {}

    # ------------------------------------------------------------------ buffers
    def empty(self, *shape, dtype=torch.float32):
        t = torch.empty(*shape, device=self.device, dtype=dtype)
        self.nbytes += t.numel() * t.element_size()
        self._keep.append(t)
        return t

    def scratch(self, key, numel):
        """Shared scratch (valid only within one op's launches on the single stream)."""
        if False:      # (debugging aid: private buffers instead of the shared scratch)
            return self.empty(max(numel, 1))
        key = (key, self.lane)
        t = self._scratch.get(key)
        if t is None or t.numel() < numel:
            if t is not None:
                self._keep.append(t)   # earlier launches recorded the old pointer
            t = torch.empty(max(numel, 1), device=self.device, dtype=torch.float32)
            self.nbytes += t.numel() * 4
            self._scratch[key] = t
        return t

    def scratch_bf16(self, key, numel):
        key = (key, self.lane)
        t = self._scratch.get(key)
        if t is None or t.numel() < numel:
            if t is not None:
                self._keep.append(t)
            t = torch.empty(max(numel, 8), device=self.device, dtype=torch.bfloat16)
            self.nbytes += t.numel() * 2
            self._scratch[key] = t
        return t

    def f16_weights(self, p, wq16=None, wsrc=None, wb=None, stream=False):
        """Register conv weights (default: the arena's forward layout) for the per-step fp16x3 preparation.
        stream: the planes of a 3x3 filter in the K-step order of dsnt_conv_fwd_f16x3_stream (csrc/conv3s.hip)."""
        wq16 = p.wq16 if wq16 is None else wq16
        key = wq16.data_ptr()
        if key not in self._f16_w_seen:
            self._f16_w_seen.add(key)
            wsrc = p.w if wsrc is None else wsrc
            wb = p.wb if wb is None else wb
            n = wsrc.numel()
            assert n % 4 == 0
            self._f16_w_rows.append([wsrc.data_ptr(), wq16.data_ptr(), wb.data_ptr(), n, p.wq_stride] +
                                    ([p.Cout, p.Cin] if stream else [0, 0]))
            self._f16_w_stream[key] = stream
        assert self._f16_w_stream.get(key, False) == stream, 'one weight tensor, two plane layouts'

    def stream_ok(self, p, g, rows, res2):
        """A 3x3 convolution the symmetric persistent kernel runs (weights in stream order): at most one residual."""
        return (self.conv3s and p.R == 3 and p.S == 3 and res2 is None and
                bool(self.lib.dsnt_conv_fwd_stream_ok(C.byref(g))))

    def f16_bn_bound(self, n):
        """Device scalar >= |relu?(bn(x))| of the train-mode BatchNorm seen through Normed n."""
        if n.abound is None:
            n.abound = self.empty(64)            # a bound = 64 slots (DSNT_BOUND_SLOTS)
            import struct
            bits = struct.unpack('<I', struct.pack('<f', float(n.x.M) ** 0.5))[0]
            self._f16_bn_rows.append([n.bn.gamma.data_ptr(), n.bn.beta.data_ptr(), n.abound.data_ptr(), n.bn.C, bits])
        return n.abound

    def amax_slot(self):
        if self._amax_buf is None:
            self._amax_buf = self.empty(64 * 4096)
        assert self._amax_used + 64 <= self._amax_buf.numel(), 'backward bound slots exhausted'
        self._amax_used += 64
        return self._amax_buf[self._amax_used - 64:self._amax_used]

    def operand_amax(self, x):
        """Bound slot for activation x used as a RAW fp16x3 operand (no BatchNorm in between: skip projections, `lin`
        convolutions): the launch that produced x is asked — through its dsnt_out_bounds, read at launch time — to leave
        max|x| there.  None if the producer cannot."""
        t = x.amax_tail
        if t is None or not self.use_f16x3 or not self.raw_f16:
            return None
        slot = self._famax_of.get(id(t))
        if slot is None:
            slot = self._famax_slot()
            t.amax = slot.data_ptr()
            self._famax_of[id(t)] = slot
            self._keep.append(t)
        return slot

    def _famax_slot(self):
        if self._famax_buf is None:
            self._famax_buf = self.empty(64 * 2048)      # hg8 with every level on fp16x3 claims ~800 slots
        assert self._famax_used + 64 <= self._famax_buf.numel(), 'forward bound slots exhausted'
        self._famax_used += 64
        return self._famax_buf[self._famax_used - 64:self._famax_used]

    def operand_amax_bn(self, n):
        """Eval mode: bound slot for relu?(bn(x)) as an fp16x3 operand.  The BatchNorm vectors come from running
        statistics (one dsnt_bn_eval_prep launch at the head of the forward), so the launch that produces x can form
        the operand itself and leave its exact maximum (dsnt_out_bounds.amax_bn).  One BatchNorm per producer; None if
        the producer cannot or is already taken by another BatchNorm."""
        t = n.x.amax_tail
        if t is None or not self.use_f16x3 or not self.raw_f16 or self.training:
            return None
        got = self._famax_bn_of.get(id(t))
        if got is not None:
            return got[1] if got[0] is n else None
        slot = self._famax_slot()
        t.amax_bn, t.amax_scale, t.amax_shift = slot.data_ptr(), n.scale.data_ptr(), n.shift.data_ptr()
        t.amax_relu = 1 if n.relu else 0
        self._famax_bn_of[id(t)] = (n, slot)
        self._keep.append(t)
        return slot

    def f16_bn_bound_bwd(self, n):
        """The same bound for a backward consumer.  The forward list computes it every step (emit_f16_prep) only if
        some forward conv asked for it before the list was closed; rows added later would be lost."""
        assert n.abound is not None, 'fp16x3 backward needs the BN bound its forward conv registered'
        return n.abound

    # launches that read weight planes or operand bounds: the preparation must have finished before the first of them
    _PREP_CONSUMERS = ('dsnt_conv_fwd_f16x3_ex', 'dsnt_conv_fwd_f16x3_stream', 'dsnt_conv1x1_fwd_f16x3', 'dsnt_stem4_fwd_f16x3',
                       'dsnt_conv_fwd_bf16x6_ex', 'dsnt_conv_fwd_bf16x6')

    def emit_f16_prep(self, pos, head=()):
        """Insert the per-step preparation launches at position `pos` of the forward list: fp16x3 weight planes and BN
        operand bounds, eval-mode BN vectors and — weights do not change between a forward and its backward — the
        re-packed / split data-gradient weights.  `head` = entries already on the list at `pos` that belong to the
        preparation (the bf16 split of the arena).  With lanes, all of it runs on the side lane beside the stem
        convolution (which reads fp32 weights) and the main lane waits right before the first launch that needs it."""
        # (eval mode: the BN vectors are needed at once, and there is nothing to hide the launches behind)
        side = 1 if (self.use_lanes and self.prep_on_side_lane and self.training) else 0
        saved, self.fwd = self.fwd, []
        saved_lane, self.lane = self.lane, side
        for fn, args, name, _ in head:
            self.fwd.append((fn, args, name, side))
        if self._f16_w_rows:
            t = torch.tensor(self._f16_w_rows, dtype=torch.int64).to(self.device)
            self._keep.append(t)
            self.f('dsnt_f16_prep_weights', t, len(self._f16_w_rows), 7)
        if self._f16_bn_rows:
            t = torch.tensor(self._f16_bn_rows, dtype=torch.int64).to(self.device)
            self._keep.append(t)
            self.f('dsnt_f16_prep_bn_bounds', t, len(self._f16_bn_rows))
        if self._eval_bn_rows:
            t = torch.tensor(self._eval_bn_rows, dtype=torch.int64).to(self.device)
            self._keep.append(t)
            self.f('dsnt_bn_eval_prep', t, len(self._eval_bn_rows))
        fwd_half = len(self.fwd)                # everything from here on is read by the backward list only
        if self._dgrad_pack is not None:
            self.f('dsnt_conv_pack_dgrad_all', *self._dgrad_pack)
        if self._f16_dw_rows:
            t = torch.tensor(self._f16_dw_rows, dtype=torch.int64).to(self.device)
            self._keep.append(t)
            self.f('dsnt_f16_prep_weights', t, len(self._f16_dw_rows), 7)
        self.lane = 0
        if self._famax_used:
            self.f('dsnt_fill_zero', self._famax_buf, self._famax_used)
            self.fwd.insert(0, self.fwd.pop())      # main lane, first: every producer comes after it
        self.lane = saved_lane
        prep, self.fwd = self.fwd, saved
        # (the fill_zero entry was moved to the front: the boundary index moves with it)
        fwd_half = fwd_half + 1 if self._famax_used else fwd_half
        relay = None
        if side and self.prep_split and self.n_lanes > 3 and 0 < fwd_half < len(prep):
            # the backward-only half keeps running on the side lane while the forward proceeds: an idle lane takes the dependency
            # on the FIRST half here (a list entry records and waits at one position), the main lane waits for that lane below
            relay = self.n_lanes - 1
            prep.insert(fwd_half, (None, (side, relay, torch.cuda.Event()), 'sync', 0))
        del self.fwd[pos:pos + len(head)]
        self.fwd[pos:pos] = prep
        if side and prep:
            first = next((i for i in range(pos + len(prep), len(self.fwd))
                          if self.fwd[i][0] is not None and self.fwd[i][2] in self._PREP_CONSUMERS
                          and id(self.fwd[i]) not in self._prep_exempt), len(self.fwd))
            if relay is not None:
                # the relay lane is taken to be idle up to here: an op (or a wait) traced onto it before `first` would put the main
                # lane behind that op without anyone noticing — refuse at trace time instead
                for ent in self.fwd[pos + len(prep):first]:
                    busy = ent[3] == relay if ent[0] is not None else (ent[2] == 'sync' and relay in ent[1][:2])
                    assert not busy, 'prep relay lane %d is in use before the first prep consumer (%s)' % (relay, ent[2])
            self.fwd.insert(first, (None, (side if relay is None else relay, 0, torch.cuda.Event()), 'sync', 0))

    def _use6(self, g):
        return (self.use_bf16x6 and g.N * g.Ho * g.Wo >= self.bf16x6_min_rows and
                bool(self.lib.dsnt_conv_bf16x6_ok(C.byref(g))))

    def act(self, N, H, W, Cc, name=''):
        a = Act(self.empty(N, H, W, Cc), name)
        self.acts.append(a)
        return a

    # ------------------------------------------------------------------ emission helpers
    def _emit(self, lst, name, *args):
        fn = getattr(self.lib, name)
        conv = []
        for a in args:
            if isinstance(a, torch.Tensor):
                # algorithmic HBM bytes of the step (bench.py `step_bounds`): every tensor a launch names is read or
                # written once by it; weight-gradient slabs (workspace, not algorithm) are left out
                if a.data_ptr() not in self._ws_ptrs:
                    nb = a.numel() * a.element_size()
                    self.bytes_by_name[name] = self.bytes_by_name.get(name, 0) + nb
                    if lst is self.fwd:
                        self.bytes_fwd += nb
                    else:
                        self.bytes_bwd += nb
                conv.append(_lib.ptr(a))
            elif isinstance(a, (ConvGeom, BnBwdEpilogue, BnTail, BnPrologue, BnBwdApply)):
                self._keep.append(a)
                conv.append(C.byref(a))
            else:
                conv.append(a)
        entry = (fn, tuple(conv), name, self.lane)
        lst.append(entry)
        return entry

    def f(self, name, *args):
        return self._emit(self.fwd, name, *args)

    def b(self, name, *args):
        return self._emit(self.bwd, name, *args)

    def on_backward(self, fn):
        lane = self.lane

        def emit():
            saved, self.lane = self.lane, lane
            try:
                fn()
            finally:
                self.lane = saved
        self._bwd_emitters.append(emit)

    # ------------------------------------------------------------------ lanes
    def sync(self, src, dst, bwd=True):
        """`dst` lane waits for everything emitted so far on `src`; mirrored in backward unless bwd=False."""
        if not self.use_lanes:
            return
        self.fwd.append((None, (src, dst, torch.cuda.Event()), 'sync', 0))
        if self.record and bwd:
            self._bwd_emitters.append(
                lambda: self.bwd.append((None, (dst, src, torch.cuda.Event()), 'sync', 0)))

    def sync_bwd(self, src, dst):
        """Backward-list only: `dst` lane waits for what has been emitted on `src` so far."""
        if self.use_lanes and src != dst:
            self.bwd.append((None, (src, dst, torch.cuda.Event()), 'sync', 0))

    def branch(self, x, join=True):
        """Alias of activation x for a branch traced on the side lane: same buffer and statistics,
        private gradient (the two lanes must not accumulate into one buffer concurrently); the
        private gradient is added to x's after the lanes have joined in backward (join=False: the
        caller hands it over itself — Hourglass._level)."""
        if not self.use_lanes:
            return x
        xb = Act(x.buf, x.name + '/branch')
        xb.stats, xb.amax_tail = x.stats, x.amax_tail
        xb.is_branch = True
        x.uses += 1
        if self.record and join:
            def join_grad():
                if xb.grad is not None:
                    self.grad_identity(x, xb.grad, donate=False, g_amax=xb.grad_amax)
            self.on_backward(join_grad)        # registered first -> runs after the branch's backward
        return xb

    def finish(self):
        """Emit the backward list (reverse order of the forward ops)."""
        if self.dgrad_slots:
            # one launch re-packs (tap-flipped, transposed) and bf16-splits every dgrad weight
            total = (self.dgrad_total + 7) // 8 * 8
            self.dgrad_f32 = self.empty(total)
            self.dgrad_planes = self.empty(3 * total, dtype=torch.bfloat16)
            rows = []
            for p, dst in self.dgrad_slots:
                src = (p.w.data_ptr() - self.param_arena.data_ptr()) // 4
                rows.append([src, dst, p.Cout, p.R, p.S, p.Cin])
            table = torch.tensor(rows, dtype=torch.int32).to(self.device)
            self._keep.append(table)
            # launched by the forward list's preparation (emit_f16_prep), off the backward's critical path
            self._dgrad_pack = (table, len(rows), self.param_arena, self.dgrad_f32, self.dgrad_planes, total)
            self.dgrad_total = total
            if self.use_f16x3:
                self.dgrad_planes16 = self.empty(2 * total, dtype=torch.float16)
                self.dgrad_bounds = self.empty(64 * len(rows))
        prep_pos = len(self.bwd)
        self._release_left = self._release_total
        for fn in reversed(self._bwd_emitters):
            fn()
        self._bwd_emitters = []
        # fp16x3: zero the amax slots (the re-packed data-gradient weights are split by the forward list's preparation)
        saved, self.bwd = self.bwd, []
        if self._amax_used:
            self.b('dsnt_fill_zero', self._amax_buf, self._amax_used)
        prep, self.bwd = self.bwd, saved
        self.bwd[prep_pos:prep_pos] = prep
        if self._pending_reduce:          # convolutions outside every parameter bucket (stand-alone modules)
            self.lane = self._flush_lane()
            self.flush_wgrad()

    def release_wgrads(self):
        """Put the held-back weight-gradient launches on the backward list here."""
        if self._held:
            self.bwd.extend(self._held)
            self._held = []

    def _flush_lane(self):
        """The lane that reduces a bucket's weight-gradient slabs, made to wait for every lane that wrote one.  With a
        weight-gradient lane it is that lane — the main lane neither waits for the backlog of weight gradients nor
        runs the reduction; without, the main lane."""
        self.release_wgrads()
        fl = self.wgrad_lane if self.wgrad_lane is not None else 0
        for src in self.chain_lanes:
            self.sync_bwd(src, fl)
        return fl

    def flush_wgrad(self):
        """Emit the one-launch reduction of every weight-gradient slab written since the last flush."""
        rows = self._pending_reduce
        if not rows:
            return
        self._pending_reduce = []
        if self._pending_group:
            descs, self._pending_group = self._pending_group, []
            blob = b''.join(d for d, _ in descs)
            gtable = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(self.device)
            self._keep.append(gtable)
            blocks = (max(n for _, n in descs) + 7) // 8 * 8
            e = self.b('dsnt_conv_wgrad_group', gtable, len(descs), blocks)
            for u in self._pending_group_uses:
                self.f16_uses.append((e, u))
            self._pending_group_uses = []
        table = torch.tensor(rows, dtype=torch.int64).to(self.device)
        self._keep.append(table)
        blocks = max((r[4] // 4 + (r[5] + 3) // 4 + 63) // 64 for r in rows)
        self.b('dsnt_wgrad_reduce_all', table, len(rows), blocks)
        for nm, args in self._post_reduce:          # gradients that are reduced in another layout than the parameter's
            self.b(nm, *args)
        self._post_reduce = []

    def mark_bucket(self, k):
        """Forward position where parameter bucket k starts being used: in the (reversed) backward
        list the marker lands right after the last launch that writes bucket k's gradients."""
        self.cur_bucket = k
        if self.record:
            def mark():
                lane, self.lane = self.lane, self._flush_lane()
                self.flush_wgrad()
                # the marker carries the lane on which the bucket's gradients become complete: `run` calls the hook
                # with that lane's stream current, so a collective started there orders itself after the reduction
                self.bwd.append((None, k, 'bucket', self.lane))
                self.lane = lane
            self.on_backward(mark)

    def flush_point(self):
        """Forward position behind which (in backward: before which) the weight-gradient slabs written so far are reduced and the
        grouped low-resolution weight gradients launched, without closing the parameter bucket.  A ResNet is ONE bucket: all of its
        grouped weight gradients (every convolution of <= 8192 rows: 337 us at batch 8) and its one slab reduction (128 us) used to
        run behind the last data gradient, with nothing beside them; flushed after every stage they run on the weight-gradient
        lane beside the next stage's launch-bound chain.  (On the hourglass the same cut inside the stem's bucket gains nothing:
        profiles/r05_ab_switches.txt box B.)"""
        if self.record and self.defer_reduce and self.flush_points:
            def flush():
                lane, self.lane = self.lane, self._flush_lane()
                self.flush_wgrad()
                self.lane = lane
            self.on_backward(flush)

    def _compile(self, lst):
        """Record `lst` into a C launch list: (handle, bucket ids after each segment)."""
        lib = self.lib
        h = lib.dsnt_list_create()
        if not h:
            raise RuntimeError('dsnt_list_create failed')
        marks = []
        try:
            if self.use_lanes:
                for lane in range(1, self.n_lanes):
                    self._rc(lib.dsnt_list_sync(h, 0, lane), 'dsnt_list_sync')
            self._rc(lib.dsnt_list_begin(h), 'dsnt_list_begin')
            try:
                for fn, args, name, lane in lst:
                    if fn is None:
                        if name == 'sync':
                            self._rc(lib.dsnt_list_sync(h, args[0], args[1]), 'dsnt_list_sync')
                        else:
                            marks.append((args, lane))
                            lib.dsnt_list_mark(h)
                        continue
                    self._rc(fn(*args, C.c_void_p(lane)), name)          # recorded, not launched: `stream` = lane index
            finally:
                lib.dsnt_list_end()
            if self.use_lanes:
                for lane in range(1, self.n_lanes):
                    self._rc(lib.dsnt_list_sync(h, lane, 0), 'dsnt_list_sync')
            self._fuse_stages(h)
        except Exception:
            lib.dsnt_list_destroy(h)
            raise
        return h, marks

    def _fuse_stages(self, h):
        """Runs of small dependent launches of one lane (the 8 x 8 / 4 x 4 hourglass levels) -> persistent stage launches
        (include/dsnt_hip.h: dsnt_list_fuse).  The tables and counters live in a tensor this tape owns."""
        if not self.stage or self.device.type != 'cuda':
            return
        lib = self.lib
        need = lib.dsnt_list_fuse_bytes(h, self.stage_min_run, self.stage_max_vgrid)
        if need <= 0:
            return
        ws = torch.zeros((need + 63) // 64 * 64, dtype=torch.uint8, device=self.device)
        self._keep.append(ws)
        rc = lib.dsnt_list_fuse(h, C.c_void_p(ws.data_ptr()), ws.numel(), self.stage_min_run, self.stage_max_vgrid, self.stage_grid)
        if rc < 0:
            self._rc(rc, 'dsnt_list_fuse')
        inside = C.c_int(0)
        n = lib.dsnt_list_stages(h, C.byref(inside))
        self.stage_census.append((n, inside.value))

    def stage_errors(self):
        """Persistent stages of this tape's compiled lists whose barrier gave up (a workgroup never arrived) since they were
        built: 0 in a healthy process.  Synchronises the device (diagnostic / tests)."""
        torch.cuda.synchronize()
        return sum(max(0, self.lib.dsnt_list_stage_errors(h)) for h, _ in self._clists.values())

    def _rc(self, rc, name):
        if rc != 0:
            raise RuntimeError('%s failed (%d): %s' % (name, rc, self.lib.dsnt_last_error().decode()))

    def __del__(self):
        try:
            for h, _ in self._clists.values():
                self.lib.dsnt_list_destroy(h)
        except Exception:
            pass

    def run(self, lst, bucket_hook=None, probe=None):
        """Replay a launch list.  `probe(entry)` (diagnostics / tests) is called before every launch."""
        main = torch.cuda.current_stream()
        if self.use_lanes and self.side_stream is None:
            # (stream priorities were tried — chain on high-priority streams, weight gradients on a normal one — and
            # change nothing on this hardware)
            self.side_stream = torch.cuda.Stream()
            self.wgrad_stream = torch.cuda.Stream()
            self.more_streams = tuple(torch.cuda.Stream() for _ in range(self.n_lanes - 3))
        return self._run(lst, main, bucket_hook, probe)

    def _run(self, lst, main, bucket_hook, probe):
        streams = (main, self.side_stream, self.wgrad_stream) + self.more_streams
        ptrs = tuple(st.cuda_stream if st is not None else 0 for st in streams)
        if self.c_replay and probe is None:
            cl = self._clists.get(id(lst))
            if cl is None:
                cl = self._clists[id(lst)] = self._compile(lst)
            h, marks = cl
            arr = (C.c_void_p * len(ptrs))(*ptrs)
            for seg in range(len(marks) + 1):
                rc = self.lib.dsnt_list_replay(h, seg, arr, len(ptrs))
                if rc != 0:
                    torch.cuda.synchronize()
                    self._rc(rc, 'dsnt_list_replay')
                if seg < len(marks) and bucket_hook is not None:
                    k, lane = marks[seg]
                    with torch.cuda.stream(streams[lane]):
                        bucket_hook(k)
            return
        if self.use_lanes:
            self.side_stream.wait_stream(main)
            self.wgrad_stream.wait_stream(main)
            for st in self.more_streams:
                st.wait_stream(main)
        for entry in lst:
            fn, args, name, lane = entry
            if fn is None:
                if name == 'sync':
                    src, dst, ev = args
                    ev.record(streams[src])
                    streams[dst].wait_event(ev)
                elif bucket_hook is not None:
                    with torch.cuda.stream(streams[lane]):
                        bucket_hook(args)
                continue
            if probe is not None:
                probe(entry)
            rc = fn(*args, ptrs[lane])
            if rc != 0:
                torch.cuda.synchronize()
                raise RuntimeError('%s failed (%d): %s' % (
                    name, rc, _lib.load().dsnt_last_error().decode()))
        if self.use_lanes:
            main.wait_stream(self.side_stream)
            main.wait_stream(self.wgrad_stream)
            for st in self.more_streams:
                main.wait_stream(st)

    # ------------------------------------------------------------------ gradient plumbing
    def add_later(self, a, g, g_amax=None):
        """a.grad += g, left to the next writer of a.grad if that writer can add a second tensor in its own pass
        (`maxpool2`'s backward: dsnt_maxpool2_bwd_add); any other writer first emits the addition as a launch of its own."""
        assert a.grad is not None and a.pending_add is None
        a.pending_add = (g, g_amax)

    def _flush_add(self, a):
        (g, g_amax), a.pending_add = a.pending_add, None
        self.grad_identity(a, g, donate=False, g_amax=g_amax)

    def identity_bn(self, Cc):
        """(ones, zeros) of >= Cc channels: the BatchNorm prologue that changes nothing (a raw operand for a launch form that only
        exists with a prologue — the grouped weight gradients)."""
        if self._ident is None or self._ident[0].numel() < Cc:
            n = max(Cc, 2048)
            self._ident = (torch.ones(n, device=self.device), torch.zeros(n, device=self.device))
        return self._ident

    def take_base(self):
        """After grad_target(..., base_ok=True): the tensor the writer has to add to its value (or None)."""
        b, self._writer_base = self._writer_base, None
        return b

    def grad_target(self, a, amax=False, take_add=False, base_ok=False):
        """(buffer, accumulate flag) for a kernel about to write a's gradient.  `amax`: the kernel leaves max|written|
        in a.grad_amax (fp16x3 operand bound).  Every writer rewrites the whole tensor, so the slot stays a valid bound
        while all writers since its creation report into it; a writer that cannot invalidates it."""
        if a.pending_apply is not None:          # a BatchNorm backward left to a's producer, and now a second contribution
            self.materialize_apply(a)
        if a.pending_add is not None and not take_add:
            self._flush_add(a)                   # this writer cannot add a second tensor in its pass: the separate launch after all
        self._writer_base = None
        if a.base is not None:
            # a's gradient continues one it does not own: base + this writer's value go into a NEW buffer.  A writer that can
            # read the base itself (base_ok: `take_base`) does that in its own pass; any other finds a copy in place
            base, base_amax, a.base, a.base_amax = a.base, a.base_amax, None, None
            a._grad, a.grad_shared = self.empty(a.N, a.H, a.W, a.C), False
            if self.use_f16x3 and amax and self.amax_all:
                a.grad_amax = self.amax_slot()
            else:
                a.grad_amax = None
            if base_ok:
                self._writer_base = base
                return a._grad, 0
            if a.grad_amax is not None:
                self.b('dsnt_axpy_amax', base, a._grad, 1.0, 0, base.numel(), a.grad_amax)
            else:
                self.b('dsnt_axpy', base, a._grad, 1.0, 0, base.numel())
            return a._grad, 1
        acc = 1
        if a.grad is None:
            a.grad = self.empty(a.N, a.H, a.W, a.C)
            acc = 0
        elif a.grad.data_ptr() in self._wgrad_lane_reads:
            self._wgrad_lane_reads.discard(a.grad.data_ptr())
            self.release_wgrads()                            # a held-back launch that reads it must be on the list first
            self.sync_bwd(self.wgrad_lane, self.lane)        # a weight gradient on its own lane still reads this buffer
        if self.use_f16x3 and amax and (self.amax_all or (amax == 'apply' and acc == 0)):
            if a.grad_amax is None:
                a.grad_amax = self.amax_slot()
        else:
            a.grad_amax = None
        return a.grad, acc

    def grad_identity(self, a, g, donate, g_amax=None):
        """a.grad (+)= g.  With `donate`, g's buffer is handed over when a has no gradient yet
        (the caller guarantees g is dead after its own launches).  Returns True if donated.
        g_amax: the bound slot of g, if it has one (it moves with a donated buffer)."""
        if a.pending_apply is not None:
            self.materialize_apply(a)
        if a.pending_add is not None:
            self._flush_add(a)
        if a.grad is None and donate:
            a.grad = g
            a.grad_amax = g_amax if self.amax_all else None
            return True
        buf, acc = self.grad_target(a, amax=True)
        if a.grad_amax is not None:
            self.b('dsnt_axpy_amax', g, buf, 1.0, acc, g.numel(), a.grad_amax)
        else:
            self.b('dsnt_axpy', g, buf, 1.0, acc, g.numel())
        return False

    # ------------------------------------------------------------------ ops
    def geom(self, x, p):
        Ho = (x.H + 2 * p.pad - p.dil * (p.R - 1) - 1) // p.stride + 1
        Wo = (x.W + 2 * p.pad - p.dil * (p.S - 1) - 1) // p.stride + 1
        assert x.C == p.Cin, (x.C, p.Cin)
        return ConvGeom(x.N, x.H, x.W, p.Cin, Ho, Wo, p.Cout, p.R, p.S, p.stride, p.pad, p.dil)

    def ensure_stats(self, a):
        if a.stats is None:
            tiles = (a.M + 127) // 128
            part = self.empty(tiles, 2, a.C)
            self.f('dsnt_bn_stats', a.buf, part, a.M, a.C)
            a.stats = (part, tiles)
        return a.stats

    def norm(self, x, bn, relu=True):
        """BatchNorm bookkeeping for x under `bn`: finalise kernel -> scale/shift vectors."""
        if os.environ.get('DSNT_DEBUG_NO_RELU'):   # smooth network for exact gradient checks
            relu = False
        n = Normed()
        n.x, n.bn, n.relu = x, bn, relu
        n.abound = None
        n.mean, n.invstd, n.scale, n.shift = (self.empty(bn.C) for _ in range(4))
        n.pending = None
        n.lane = self.lane      # the lane whose launches write scale / shift: every consumer must run there (`check_lane`)
        if self.training:
            part, tiles = self.ensure_stats(x)
            if self.fuse_finalize and bn.C <= 256 and tiles * bn.C <= self.fuse_finalize_max:
                # few tiles (the 8x8 / 4x4 levels): the launch that CONSUMES this BatchNorm finalises it in its prologue
                # (`conv`); any other first reader materialises it with the usual launch (`materialize`)
                n.pending = (part, tiles)
            else:
                self.f('dsnt_bn_finalize', part, tiles, x.M, bn.C, bn.gamma, bn.beta, bn.rmean, bn.rvar,
                       bn.momentum, bn.eps, 1, n.mean, n.invstd, n.scale, n.shift)
        else:
            import struct
            bits = struct.unpack('<I', struct.pack('<f', float(bn.eps)))[0]
            self._eval_bn_rows.append([bn.gamma.data_ptr(), bn.beta.data_ptr(), bn.rmean.data_ptr(), bn.rvar.data_ptr(),
                                       n.mean.data_ptr(), n.invstd.data_ptr(), n.scale.data_ptr(), n.shift.data_ptr(),
                                       bn.C, bits])
        return n

    def check_lane(self, n):
        """A consumer of Normed n is about to be emitted on the current lane.  The vectors of n are written by launches of ONE
        lane (the finalise launch at `norm` time, or — deferred — the first consumer's prologue / `materialize`), and nothing
        orders another lane's reader behind them: a second consumer elsewhere would race silently."""
        if n.pending is not None:
            n.lane = self.lane              # this consumer finalises it, here
        elif n.lane != self.lane:
            raise RuntimeError('dsnt: BatchNorm vectors finalised on lane %d are read on lane %d without a synchronisation '
                               '(trace norm() and its consumers on one lane, or sync the lanes)' % (n.lane, self.lane))

    def materialize(self, n):
        """The separate finalise launch of a BatchNorm whose finalisation was left to its consumer (`norm`)."""
        self.check_lane(n)
        if n.pending is not None:
            part, tiles = n.pending
            n.pending = None
            bn = n.bn
            self.f('dsnt_bn_finalize', part, tiles, n.x.M, bn.C, bn.gamma, bn.beta, bn.rmean, bn.rvar,
                   bn.momentum, bn.eps, 1, n.mean, n.invstd, n.scale, n.shift)

    def fold_ok(self, n):
        """The BatchNorm backward of Normed n can be left to the backward of the convolution that produced n.x."""
        x = n.x
        return bool(self.bwd1 and x.fold_ok and x.grad is None and x.pending_apply is None)

    def materialize_apply(self, x):
        """The apply launch of a BatchNorm backward that was left to x's producer (`_norm_backward` with dz_amax) after all:
        x got a second gradient contribution, so its gradient has to exist in memory."""
        ap, x.pending_apply = x.pending_apply, None
        n = ap['n']
        buf, acc = self.grad_target(x, amax='apply')
        if x.grad_amax is not None:
            self.b('dsnt_bn_act_bwd_apply_amax', ap['dz'], x.buf, n.scale, n.shift, n.mean, n.invstd, ap['coef'], 0,
                   buf, acc, x.M, n.bn.C, x.grad_amax)
        else:
            self.b('dsnt_bn_act_bwd_apply', ap['dz'], x.buf, n.scale, n.shift, n.mean, n.invstd, ap['coef'], 0,
                   buf, acc, x.M, n.bn.C)

    def _norm_backward(self, n, da, reduced=None, finalised=False, dz_amax=None):
        """Given da = dL/d relu(bn(x)), accumulate dx into n.x.grad and dgamma/dbeta.  With
        `reduced` = (partials, ntiles) the producer already masked da by the ReLU and reduced it; with `finalised` it
        also wrote dgamma / dbeta / coef.  With dz_amax (the slot in which the producer left max|da|; the
        caller has asked `fold_ok`) dx is NOT written: the finalise launch also leaves the bound of dx, and the backward
        of the 1x1 convolution that produced x forms dx in registers (dsnt_conv1x1_bwd_f16x3)."""
        x, bn = n.x, n.bn
        coef = self.scratch('bncoef', 2 * bn.C)
        relu = 1 if n.relu else 0
        if reduced is not None:
            part, tiles = reduced
            relu = 0                      # da is already dz
        else:
            tiles = (x.M + 127) // 128
            part = self.scratch('bnpart', tiles * 2 * bn.C).view(-1)
            self.b('dsnt_bn_act_bwd_reduce', da, x.buf, n.scale, n.shift, n.mean, n.invstd, relu, part,
                   x.M, bn.C)
        if dz_amax is not None:
            assert reduced is not None and not finalised and x.grad is None
            acc_p = 1 if bn.uses > 0 else 0
            bn.uses += 1
            coef = self.empty(2 * bn.C)          # lives until the producer's backward
            bound = self.amax_slot()
            self.b('dsnt_bn_bwd_finalize_bound', part, tiles, x.M, bn.C, bn.ggamma, bn.gbeta, acc_p, coef, n.scale,
                   dz_amax, bound)
            x.pending_apply = dict(dz=da, n=n, coef=coef, bound=bound, dz_amax=dz_amax)
            return
        # eval-mode forward: the statistics are constants — dx = scale dz, i.e. the apply with both coefficients zero
        # (DSNT_BN_FROZEN); the two sums still are dbeta / dgamma
        frozen = not self.training
        fused = (not finalised) and self.fuse_finalize and bn.C <= 256 and tiles * bn.C <= self.fuse_finalize_max and not frozen
        assert not (frozen and finalised)
        if not finalised:
            acc_p = 1 if bn.uses > 0 else 0
            bn.uses += 1
            if not fused:
                self.b('dsnt_bn_bwd_finalize', part, tiles, x.M, bn.C, bn.ggamma, bn.gbeta, acc_p | (2 if frozen else 0), coef)
        buf, acc = self.grad_target(x, amax='apply', base_ok=True)
        base = self.take_base()         # x's gradient continues one it does not own: dx = base + value (`conv`, defer_res)
        if fused:
            # few tiles: the apply launch sums them itself in its prologue (coef) and writes dgamma / dbeta
            coef = self.empty(2 * bn.C)          # private: the launch writes it (a shared scratch could still be in use)
            if base is not None:
                self.b('dsnt_bn_act_bwd_apply_pro_base', da, x.buf, n.scale, n.shift, n.mean, n.invstd, part, tiles,
                       bn.ggamma, bn.gbeta, acc_p, coef, relu, base, buf, x.M, bn.C, x.grad_amax)
            else:
                self.b('dsnt_bn_act_bwd_apply_pro', da, x.buf, n.scale, n.shift, n.mean, n.invstd, part, tiles,
                       bn.ggamma, bn.gbeta, acc_p, coef, relu, buf, acc, x.M, bn.C, x.grad_amax)
        elif base is not None:
            self.b('dsnt_bn_act_bwd_apply_base', da, x.buf, n.scale, n.shift, n.mean, n.invstd, coef, relu,
                   base, buf, x.M, bn.C, x.grad_amax)
        elif x.grad_amax is not None:
            self.b('dsnt_bn_act_bwd_apply_amax', da, x.buf, n.scale, n.shift, n.mean, n.invstd, coef, relu,
                   buf, acc, x.M, bn.C, x.grad_amax)
        else:
            self.b('dsnt_bn_act_bwd_apply', da, x.buf, n.scale, n.shift, n.mean, n.invstd, coef, relu,
                   buf, acc, x.M, bn.C)

    def conv(self, src, p, res1=None, res2=None, want_stats=False, need_input_grad=True, name=''):
        """y = conv(src) + bias [+ res1 + res2]; src is an Act (raw) or a Normed (BN+ReLU folded)."""
        normed = isinstance(src, Normed)
        x = src.x if normed else src
        g = self.geom(x, p)
        y = self.act(x.N, g.Ho, g.Wo, p.Cout, name)
        x.uses += 1
        for r in (res1, res2):
            if r is not None:
                r.uses += 1
                y.gives_away = True             # dL/dy's buffer is handed to a residual input in backward
        sc = src.scale if normed else None
        sh = src.shift if normed else None
        relu = 1 if (normed and src.relu) else 0
        if normed:
            self.check_lane(src)
        part, tail = None, None
        use6 = self._use6(g) and p.wq is not None
        # the large 1x1 convolutions with both operand bounds: the LDS-staged streaming kernel (csrc/fwd1.hip) — its statistics
        # come one row per WORKGROUP, and how many that is depends on the launch's share flag (below)
        x_amax_pre = None
        if use6 and p.wq16 is not None and p.R == 1 and res2 is None and self.fwd1:
            x_amax_pre = self.operand_amax_bn(src) if normed else self.operand_amax(x)
        f1 = bool(use6 and self.use_f16x3 and p.wq16 is not None and p.R == 1 and res2 is None and self.fwd1 and
                  ((self.training and normed) or x_amax_pre is not None) and self.lib.dsnt_conv1x1_fwd_ok(C.byref(g)))
        f1_shr = 2 if (f1 and self.lane != 0 and self.conv_share) else 0
        # the stem's space-to-depth convolution (4x4, 16 -> 64 channels, a raw operand): the halo kernel of csrc/stem4.hip
        s4 = bool(use6 and self.use_f16x3 and self.stem4 and p.wq16 is not None and not normed and res1 is None and res2 is None and
                  self.lib.dsnt_stem4_fwd_ok(C.byref(g)))
        if want_stats and self.training:
            bm = 128 if use6 else self.lib.dsnt_conv_fwd_bm(C.byref(g))
            tiles = (self.lib.dsnt_conv1x1_fwd_stats_rows(C.byref(g), f1_shr) if f1 else
                     self.lib.dsnt_stem4_fwd_stats_rows(C.byref(g)) if s4 else (y.M + bm - 1) // bm)
            part = self.empty(tiles, 2, p.Cout)
            y.stats = (part, tiles)
        if use6 and self.use_f16x3:
            # the large-tile kernels can leave max|y| behind: a later consumer of the raw y claims it (operand_amax)
            y.amax_tail = tail = BnTail()
        r1 = res1.buf if res1 is not None else None
        r2 = res2.buf if res2 is not None else None
        if self.use_f16x3 and self.training and normed:
            self.f16_bn_bound(src)              # also for the weight gradient of convs whose forward is not fp16x3
        # fp16x3 needs a bound of the A operand: train-mode BatchNorm parameters, or the producer's max|x| for a raw x
        x_amax = x_amax_pre
        if x_amax is None and use6 and p.wq16 is not None:
            x_amax = self.operand_amax_bn(src) if normed else self.operand_amax(x)
        use16 = use6 and self.use_f16x3 and p.wq16 is not None and ((self.training and normed) or x_amax is not None)
        if normed and (use16 or use6):
            self.materialize(src)
        if use16:
            st = self.stream_ok(p, g, y.M, res2)
            self.f16_weights(p, stream=st)
            ab = self.f16_bn_bound(src) if (normed and self.training) else x_amax
            # (bit 1 of in_relu: a persistent kernel on a side lane leaves CUs with free LDS for the chain's kernels)
            shr = 2 if ((st or p.R == 1) and self.lane != 0 and self.conv_share) else 0
            if f1:
                e = self.f('dsnt_conv1x1_fwd_f16x3', x.buf, p.wq16, p.wq_stride, p.wb, ab, p.b, y.buf, sc, sh, relu | f1_shr, r1,
                           part, g, tail)
            elif s4:
                e = self.f('dsnt_stem4_fwd_f16x3', x.buf, p.wq16, p.wq_stride, p.wb, ab, p.b, y.buf, part, g, tail)
            else:
                e = self.f('dsnt_conv_fwd_f16x3_stream' if st else 'dsnt_conv_fwd_f16x3_ex', x.buf, p.wq16, p.wq_stride, p.wb, ab,
                           p.b, y.buf, sc, sh, relu | shr, r1, r2, part, g, None, tail)
            self.f16_uses.append((e, dict(kind='fwd', name=name, x=x.buf, sc=sc, sh=sh, relu=relu, a_bound=ab,
                                          w=p.w, w_bound=p.wb)))
        elif use6:
            self.f('dsnt_conv_fwd_bf16x6_ex', x.buf, p.wq, p.wq_stride, p.b, y.buf, sc, sh, relu, r1, r2, part, g,
                   None, tail)
        elif normed and src.pending is not None and self.lib.dsnt_conv_fwd_pro_ok(C.byref(g), src.pending[1], src.bn.C):
            # the BatchNorm of this operand has few statistics tiles: finalised in this launch's prologue
            spart, stiles = src.pending
            src.pending = None
            bn = src.bn
            pro = BnPrologue(_lib.ptr(spart), stiles, bn.C, x.M, _lib.ptr(bn.gamma), _lib.ptr(bn.beta), _lib.ptr(bn.rmean),
                             _lib.ptr(bn.rvar), bn.momentum, bn.eps, _lib.ptr(src.mean), _lib.ptr(src.invstd),
                             _lib.ptr(src.scale), _lib.ptr(src.shift))
            self._keep.extend((spart, src.mean, src.invstd, src.scale, src.shift))
            self.f('dsnt_conv_fwd_pro', x.buf, p.w, p.b, y.buf, pro, relu, r1, r2, part, g, tail)
        else:
            if normed:
                self.materialize(src)
            self.f('dsnt_conv_fwd_ex', x.buf, p.w, p.b, y.buf, sc, sh, relu, r1, r2, part, g, None, tail)
        if not self.record:
            return y
        slot = None
        if need_input_grad and self.param_arena is not None:
            slot = self.dgrad_total
            slot_k = len(self.dgrad_slots)
            self.dgrad_slots.append((p, slot))
            self.dgrad_total += (p.w.numel() + 7) // 8 * 8
        # the whole backward of this convolution as ONE launch (csrc/bwd1.hip): 1x1 behind a train-mode BatchNorm, both
        # operand bounds known on the device
        bucket = self.cur_bucket
        fuse1 = bool(self.bwd1 and (normed or x_amax is not None) and self.use_f16x3 and self.defer_reduce and
                     slot is not None and p.R == 1 and p.S == 1 and p.post_reduce is None and
                     self.lib.dsnt_conv1x1_bwd_ok(C.byref(g)))
        # ... and without residual inputs (whose gradient IS dL/dy) it can also take over the BatchNorm backward of its consumer
        y.fold_ok = fuse1 and normed and res1 is None and res2 is None
        # the same for a 3x3 convolution on the persistent fp16x3 kernel (conv2 of a Bottleneck: bn3's backward rides in the operand
        # load of conv2's data gradient, csrc/conv3s.hip MODE 4), from DSNT_X_FOLD3_ROWS output rows
        # (the predicate is the one emit_dgrad's fp16x3 stream branch needs — `d16 and d_stream and not native` — so that a geometry
        # that branch refuses is decided HERE, while the consumer BatchNorm can still be told to apply by itself)
        gd3 = ConvGeom(x.N, g.Ho, g.Wo, p.Cout, x.H, x.W, p.Cin, p.R, p.S, 1, p.dil * (p.R - 1) - p.pad, p.dil)
        fold3_ok = bool(self.fold3 and self.bwd1 and normed and self.training and res1 is None and res2 is None and slot is not None and
                        self.use_f16x3 and self.defer_reduce and p.R == 3 and p.S == 3 and p.stride == 1 and use16 and
                        y.M >= self.fold3_rows and self._use6(gd3) and self.stream_ok(p, gd3, x.M, None))
        y.fold_ok = y.fold_ok or fold3_ok

        def backward():
            gy = y.grad
            ap = y.pending_apply
            fused = fuse1 and self.dgrad_planes16 is not None
            fold3 = bool(fold3_ok and ap is not None and gy is None and self.dgrad_planes16 is not None and need_input_grad and
                         ap['bound'] is not None)
            if ap is not None and (not fused or gy is not None) and not fold3:
                self.materialize_apply(y)
                ap, gy = None, y.grad
            assert gy is not None or ap is not None, 'no gradient reached conv output ' + name
            if fused and ap is None and y.grad_amax is None:
                fused = False
            cur = wl = self.lane
            defer_res = False
            if fused:
                nw = p.w.numel()
                shr = 2 if (self.lane != 0 and self.conv_share) else 0
                nsp = self.lib.dsnt_conv1x1_bwd_splits(C.byref(g), shr)
                ws = self.empty(self.lib.dsnt_conv1x1_bwd_ws_floats(C.byref(g), shr))      # lives until the bucket's reduction
                self._ws_ptrs.add(ws.data_ptr())
                wd = self.dgrad_f32[slot:slot + nw]
                wq16 = self.dgrad_planes16[slot:slot + nw]
                wbd = self.dgrad_bounds[64 * slot_k:64 * slot_k + 64]
                self._f16_dw_rows.append([wd.data_ptr(), wq16.data_ptr(), wbd.data_ptr(), nw, self.dgrad_total, 0, 0])
                if ap is not None:
                    n2 = ap['n']
                    aps = BnBwdApply(_lib.ptr(y.buf), _lib.ptr(n2.scale), _lib.ptr(n2.mean), _lib.ptr(n2.invstd), _lib.ptr(ap['coef']))
                    dy, gb = ap['dz'], ap['bound']
                    y.pending_apply = None
                else:
                    aps, dy, gb = None, gy, y.grad_amax
                if normed:
                    ab = self.f16_bn_bound_bwd(src)
                    # (the BatchNorm in front of THIS convolution may in turn be left to the 1x1 convolution before it)
                    fold = self.fold_ok(src)
                    dz = self.empty(x.M * x.C) if fold else self.scratch('da', x.M * x.C).view(-1)[:x.M * x.C]
                    dz_amax = self.amax_slot() if fold else None
                    part = self.scratch('bnpart', nsp * 2 * x.C).view(-1)
                    xs = BnBwdEpilogue(_lib.ptr(x.buf), _lib.ptr(src.scale), _lib.ptr(src.shift), _lib.ptr(src.mean),
                                       _lib.ptr(src.invstd), 1 if src.relu else 0)
                    e = self.b('dsnt_conv1x1_bwd_f16x3', xs, dy, aps, wq16, self.dgrad_total, wbd, ab, gb, dz, part, ws,
                               dz_amax, shr, g)
                    u = dict(kind='bwd1', name=name, x=x.buf, sc=src.scale, sh=src.shift, relu=1 if src.relu else 0,
                             a_bound=ab, w=wd, w_bound=wbd)
                else:
                    # no BatchNorm in front of it (projection shortcuts, the `fc` convolutions): dL/dx itself is written, or
                    # added to what x's gradient holds already
                    ab = x_amax
                    buf, acc = self.grad_target(x, amax=self.raw_f16)
                    xs = BnBwdEpilogue(_lib.ptr(x.buf), None, None, None, None, 0)
                    e = self.b('dsnt_conv1x1_bwd_f16x3', xs, dy, None, wq16, self.dgrad_total, wbd, ab, gb, buf, None, ws,
                               x.grad_amax, shr | acc, g)
                    u = dict(kind='bwd1', name=name, x=x.buf, sc=None, sh=None, relu=0, a_bound=ab, w=wd, w_bound=wbd)
                nb = 4 * (x.buf.numel() + (y.buf.numel() if ap is not None else 0))      # (tensors named inside the structs)
                self.bytes_bwd += nb
                self.bytes_by_name['dsnt_conv1x1_bwd_f16x3'] = self.bytes_by_name.get('dsnt_conv1x1_bwd_f16x3', 0) + nb
                if ap is not None:
                    u.update(g_apply=dict(dz=dy, y=y.buf, scale=n2.scale, mean=n2.mean, invstd=n2.invstd, coef=ap['coef']),
                             g_bound=gb)
                else:
                    u.update(g=dy, g_bound=gb)
                self.f16_uses.append((e, u))
                self._pending_reduce.append([ws.data_ptr(), p.gw.data_ptr(), p.gb.data_ptr() if p.gb is not None else 0,
                                             nsp, p.Cout * g.Cin, p.Cout, 0])
                if normed:
                    self._norm_backward(src, dz, reduced=(part, nsp), finalised=False, dz_amax=dz_amax)
            else:
                # the BatchNorm backward of the layer behind a 3x3 convolution, left pending by that layer's own backward
                # (`fold_ok`), rides in this convolution's data gradient (csrc/conv3s.hip MODE 4): the launch forms dL/dy from
                # (dz, y) while it stages its operand and writes it out for the weight gradient, which therefore comes SECOND
                if fold3:
                    gy = y._grad = self.empty(y.N, y.H, y.W, y.C)
                    y.grad_amax, y.pending_apply = ap['bound'], None

                def emit_wgrad():
                    defer_res = False
                    # parameter gradients (flat arena, overwritten every step)
                    # the weight gradient feeds nothing downstream in backward: run it on its own lane so the
                    # data-gradient chain never waits for it
                    cur = self.lane
                    wl = cur
                    if self.wgrad_lane is not None:
                        # DSNT_WGRAD_LANE_ROWS > 0: only the large main-lane convolutions whose dY nobody writes again (no
                        # residual inputs: the gradient buffer is not donated onwards) — their weight gradients then fill the
                        # chip while the main lane walks the launch-bound low-resolution levels
                        if self.wgrad_lane_rows == 0 or (cur in self.wgrad_lane_from and (self.wgrad_lane_res or (res1 is None and res2 is None)) and
                                                         g.N * g.Ho * g.Wo >= self.wgrad_lane_rows):
                            wl = self.wgrad_lane
                    hold = wl != cur and self._release_left > 0
                    if hold:
                        listed, self.bwd = self.bwd, self._held
                    self.sync_bwd(cur, wl)
                    self.lane = wl
                    if wl != cur:
                        # gy may be donated onwards and accumulated into by a later launch of another lane: that writer waits
                        # for the weight-gradient lane first (grad_target)
                        self._wgrad_lane_reads.add(gy.data_ptr())
                    nws = self.lib.dsnt_conv_wgrad_ws_floats(C.byref(g))
                    # DSNT_WGRAD_SHARE_CHIP: one workgroup per CU beside the chain — except for the first convolution of the
                    # network (no data gradient: it is the LAST launch of backward and has the chip to itself)
                    share = 2 if (wl != cur and self.wgrad_share and need_input_grad) else 0
                    # DSNT_WGRAD_NARROW for the hourglass stacks' 3x3 weight gradients: the lane has slack there (the 1x1 weight
                    # gradients left it), the stem's bucket is the tail of backward and keeps the wider plan
                    if share and self.wgrad_narrow in ('1', {0: '2'}.get(bucket, '3')):
                        share |= 4
                    w6 = self.use_bf16x6 and bool(self.lib.dsnt_conv_wgrad_bf16x6_ok(C.byref(g)))
                    if self.defer_reduce:
                        # deferred to the bucket's grouped launch: x, gy and the BN vectors are written once per
                        # step and gy is not donated onwards (no residual inputs), so they are intact at the flush
                        # (... nor shared with an activation whose gradient is accumulated into later: `share_grads`)
                        # (a RAW operand — conv1 of a torchvision BasicBlock, resnet.py — rides with the identity prologue: scale 1, shift 0)
                        groupable = (w6 and (normed or self.share_grads) and wl == cur and not y.grad_shared and
                                     0 < g.N * g.Ho * g.Wo <= self.group_rows)
                        # ... or, WITH residual inputs (conv3 of the low-resolution Bottlenecks): dL/dy is not donated to them but
                        # handed over as the `base` their own gradient continues out of place, and so stays intact as well
                        residuals = [r for r in (res1, res2) if r is not None]
                        defer_res = bool(groupable and residuals and self.share_grads and self.defer_res and all(
                            r._grad is None and r.base is None and r.pending_apply is None and r.pending_add is None
                            for r in residuals))
                        grouped = groupable and (not residuals or defer_res)
                        # fp16x3: both operand bounds exist (A: train-mode BN parameters, dY: one bn-backward apply wrote it)
                        w16 = w6 and self.use_f16x3 and (normed or x_amax is not None) and y.grad_amax is not None
                        ab = (self.f16_bn_bound_bwd(src) if normed else x_amax) if w16 else None
                        splits = self.lib.dsnt_conv_wgrad_splits(C.byref(g))
                        if w16 and not grouped:
                            # dsnt_conv_wgrad_f16x3 cuts the pixels of a 3x3 convolution into its own slabs (halo kernel)
                            nws = self.lib.dsnt_conv_wgrad_f16x3_ws_floats(C.byref(g), share)
                            splits = self.lib.dsnt_conv_wgrad_f16x3_splits(C.byref(g), share)
                        ws = self.empty(nws)         # lives until the bucket's reduction
                        self._ws_ptrs.add(ws.data_ptr())
                        if grouped:
                            desc = C.create_string_buffer(self.lib.dsnt_conv_wgrad_desc_bytes())
                            gsc, gsh = (sc, sh) if normed else self.identity_bn(x.C)
                            if w16:
                                nblk = self.lib.dsnt_conv_wgrad_desc_f16x3(_lib.ptr(x.buf), _lib.ptr(gsc), _lib.ptr(gsh), relu,
                                                                           _lib.ptr(gy), _lib.ptr(ws), _lib.ptr(ab),
                                                                           _lib.ptr(y.grad_amax), C.byref(g), desc)
                            else:
                                nblk = self.lib.dsnt_conv_wgrad_desc(_lib.ptr(x.buf), _lib.ptr(gsc), _lib.ptr(gsh), relu,
                                                                     _lib.ptr(gy), _lib.ptr(ws), C.byref(g), desc)
                            if nblk <= 0:
                                raise RuntimeError('dsnt_conv_wgrad_desc failed: %s' % self.lib.dsnt_last_error().decode())
                            self._pending_group.append((desc.raw, nblk))
                            if w16:
                                self._pending_group_uses.append(dict(kind='wgrad', name=name, x=x.buf, sc=sc if normed else None,
                                                                     sh=sh if normed else None, relu=relu,
                                                                     a_bound=ab, g=gy, g_bound=y.grad_amax))
                        elif w16:
                            e = self.b('dsnt_conv_wgrad_f16x3', x.buf, sc, sh, relu, gy, ws, None, None, share, ab, y.grad_amax, g)
                            self.f16_uses.append((e, dict(kind='wgrad', name=name, x=x.buf, sc=sc, sh=sh, relu=relu, a_bound=ab,
                                                          g=gy, g_bound=y.grad_amax)))
                        else:
                            self.b('dsnt_conv_wgrad_bf16x6' if w6 else 'dsnt_conv_wgrad', x.buf, sc, sh, relu, gy, ws,
                                   None, None, share if w6 else 0, g)
                        if p.post_reduce is not None:
                            self._post_reduce.append(p.post_reduce)
                        self._pending_reduce.append([ws.data_ptr(), p.gw.data_ptr(), p.gb.data_ptr() if p.gb is not None else 0,
                                                     splits, p.Cout * g.R * g.S * g.Cin, p.Cout, 0])
                    else:
                        ws = self.scratch('wgrad', nws)
                        self.b('dsnt_conv_wgrad_bf16x6' if w6 else 'dsnt_conv_wgrad', x.buf, sc, sh, relu, gy, ws,
                               p.gw, p.gb, 0, g)
                        if p.post_reduce is not None:      # the stem's space-to-depth gradient -> the parameter's 7x7 layout
                            self.b(p.post_reduce[0], *p.post_reduce[1])
                    self.lane = cur
                    if hold:
                        self.bwd = listed
                    return wl, defer_res

                def emit_dgrad():
                    if not need_input_grad:
                        return
                    nw = p.w.numel()
                    pad_d = p.dil * (p.R - 1) - p.pad
                    assert pad_d >= 0, 'data gradient needs pad <= dil * (R - 1)'
                    native = p.stride != 1 and slot is not None and self.lib.dsnt_conv_dgrad_strided_ok(C.byref(g))
                    if p.stride == 1:
                        gd = ConvGeom(x.N, g.Ho, g.Wo, p.Cout, x.H, x.W, p.Cin, p.R, p.S, 1, pad_d, p.dil)
                    elif native:
                        # strided convolution (ResNet stage transitions): dsnt_conv_dgrad_strided computes the pixels of dX
                        # phase by phase straight from dY (csrc/dgrad_up.hip)
                        gd = None
                    else:
                        # ... or, for the shapes that kernel refuses: the stride-1 data gradient of dY with stride-1
                        # zeros stuffed between the pixels
                        Hs = x.H + 2 * p.pad - p.dil * (p.R - 1)
                        Ws = x.W + 2 * p.pad - p.dil * (p.S - 1)
                        stuffed = self.scratch('stuffed', x.N * Hs * Ws * p.Cout).view(-1)[:x.N * Hs * Ws * p.Cout]
                        self.b('dsnt_zero_insert', gy, stuffed, x.N, g.Ho, g.Wo, p.Cout, Hs, Ws, p.stride)
                        gy_d = stuffed
                        gd = ConvGeom(x.N, Hs, Ws, p.Cout, x.H, x.W, p.Cin, p.R, p.S, 1, pad_d, p.dil)
                    if slot is not None:
                        wd = self.dgrad_f32[slot:slot + nw]
                        wq, wq_stride = self.dgrad_planes[slot:slot + nw], self.dgrad_total
                        d6 = gd is not None and self._use6(gd)
                    else:       # stand-alone use without a parameter arena
                        wd = self.scratch('wdgrad', nw)
                        self.b('dsnt_conv_pack_dgrad', p.w, wd, p.Cout, p.R, p.S, p.Cin)
                        d6 = self._use6(gd) and nw % 8 == 0
                        if d6:
                            wq, wq_stride = self.scratch_bf16('wdgrad6', 3 * nw), nw
                            self.b('dsnt_split_bf16x3', wd, wq, nw)

                    gsrc = gy if (p.stride == 1 or native) else gy_d

                    g_amax = y.grad_amax if (self.use_f16x3 and p.stride == 1) else None
                    d16 = d6 and g_amax is not None and slot is not None and self.dgrad_planes16 is not None
                    if d16:
                        wq16 = self.dgrad_planes16[slot:slot + nw]
                        wbd = self.dgrad_bounds[64 * slot_k:64 * slot_k + 64]
                        # (the data gradient's filter is [Cin][3][3][Cout]: its "Cout" is this convolution's Cin)
                        d_stream = self.stream_ok(p, gd, x.M, None)
                        self._f16_dw_rows.append([wd.data_ptr(), wq16.data_ptr(), wbd.data_ptr(), nw, self.dgrad_total] +
                                                 ([p.Cin, p.Cout] if d_stream else [0, 0]))

                    def dgrad(out, res, part=None, bnb=None, tail=None):
                        if native:
                            self.b('dsnt_conv_dgrad_strided', gy, wd, out, res, part, g, bnb, tail)
                        elif d16:
                            e = self.b('dsnt_conv_fwd_f16x3_stream' if d_stream else 'dsnt_conv_fwd_f16x3_ex', gsrc, wq16,
                                       self.dgrad_total, wbd, g_amax, None, out, None,
                                       None, 2 if ((d_stream or p.R == 1) and self.lane != 0 and self.conv_share) else 0, res, None, part, gd, bnb, tail)
                            self.f16_uses.append((e, dict(kind='dgrad', name=name, g=gsrc, g_bound=g_amax, w=wd, w_bound=wbd)))
                        elif d6:
                            self.b('dsnt_conv_fwd_bf16x6_ex', gsrc, wq, wq_stride, None, out, None, None, 0, res, None,
                                   part, gd, bnb, tail)
                        else:
                            self.b('dsnt_conv_fwd_ex', gsrc, wd, None, out, None, None, 0, res, None, part, gd, bnb, tail)
                    if normed:
                        # the ReLU mask and the two per-channel sums of the BatchNorm backward ride in the
                        # data-gradient epilogue; only finalise + apply remain as separate launches
                        # (fold: the 1x1 convolution that produced x forms this BatchNorm's dx in its own backward — dz then has
                        # to outlive this op's launches, and its maximum is what the bound of dx is made from)
                        fold = not native and self.fold_ok(src)
                        dz = self.empty(x.M * x.C) if fold else self.scratch('da', x.M * x.C).view(-1)[:x.M * x.C]
                        if native:
                            tiles = self.lib.dsnt_conv_dgrad_strided_tiles(C.byref(g))
                        else:
                            bm = 128 if d6 else self.lib.dsnt_conv_fwd_bm(C.byref(gd))
                            tiles = (x.M + bm - 1) // bm
                        part = self.scratch('bnpart', tiles * 2 * x.C).view(-1)
                        bnb = BnBwdEpilogue(_lib.ptr(x.buf), _lib.ptr(src.scale), _lib.ptr(src.shift),
                                            _lib.ptr(src.mean), _lib.ptr(src.invstd), 1 if src.relu else 0)
                        tl, dz_amax = None, None
                        if fold:
                            tl, dz_amax = BnTail(), self.amax_slot()
                            tl.amax = dz_amax.data_ptr()
                        if fold3:
                            assert d16 and d_stream and not native
                            n2 = ap['n']
                            aps = BnBwdApply(_lib.ptr(y.buf), _lib.ptr(n2.scale), _lib.ptr(n2.mean), _lib.ptr(n2.invstd), _lib.ptr(ap['coef']))
                            shr = 2 if (self.lane != 0 and self.conv_share) else 0
                            e = self.b('dsnt_conv_dgrad_f16x3_stream_apply', ap['dz'], aps, gy, wq16, self.dgrad_total, wbd, ap['bound'],
                                       dz, part, shr, gd, bnb, tl)
                            nb = 4 * y.buf.numel()          # (the tensor named inside the struct)
                            self.bytes_bwd += nb
                            self.bytes_by_name['dsnt_conv_dgrad_f16x3_stream_apply'] = self.bytes_by_name.get('dsnt_conv_dgrad_f16x3_stream_apply', 0) + nb
                            self.f16_uses.append((e, dict(kind='dgrad', name=name, w=wd, w_bound=wbd, g_bound=ap['bound'],
                                                          g_apply=dict(dz=ap['dz'], y=y.buf, scale=n2.scale, mean=n2.mean,
                                                                       invstd=n2.invstd, coef=ap['coef']))))
                        else:
                            dgrad(dz, None, part, bnb, tl)
                        self._norm_backward(src, dz, reduced=(part, tiles), dz_amax=dz_amax)
                    else:
                        # (d6: the large-tile kernels; their epilogue can leave max|written gradient| as the next bound)
                        buf, acc = self.grad_target(x, amax=(bool(d6) or bool(native)) and self.raw_f16, base_ok=True)
                        base = self.take_base()         # (x's gradient continues one it does not own: the residual operand)
                        tl = None
                        if x.grad_amax is not None:
                            tl = BnTail()
                            tl.amax = x.grad_amax.data_ptr()
                        dgrad(buf, base if base is not None else (buf if acc else None), tail=tl)

                if fold3:
                    emit_dgrad()
                    wl, defer_res = emit_wgrad()
                else:
                    wl, defer_res = emit_wgrad()
                    emit_dgrad()
            # identity branches last: gy is dead after the launches above (the weight-gradient lane
            # must have read it before anyone accumulates into the donated buffer)
            if res1 is not None or res2 is not None:
                self.sync_bwd(wl, cur)
            donated = False
            for r in (res1, res2):
                if r is not None and defer_res:
                    r.base, r.base_amax = gy, (y.grad_amax if self.amax_all else None)
                    continue
                if r is not None:
                    if (donated and self.share_grads and r.grad is None and r.pending_apply is None and r.uses == 1 and
                            not r.gives_away and not r.is_branch):
                        # the buffer went to the first residual input; this one has a single consumer, so its gradient is dL/dy
                        # and nothing else, and its producer only READS it (it gives no buffer away): shared, not copied.  Every
                        # later writer of the buffer comes after that reader in lane order, or waits for the lane that reads
                        # (`_wgrad_lane_reads` is keyed by the buffer)
                        r.grad = gy
                        r.grad_amax = y.grad_amax if self.amax_all else None
                        r.grad_shared = True
                        continue
                    donated = self.grad_identity(r, gy, donate=not donated, g_amax=y.grad_amax) or donated

        self.on_backward(backward)
        return y

    def bn_act(self, x, bn, relu=True, name=''):
        """Materialised y = relu?(bn(x)) (the stem: hourglass.py:157-159)."""
        n = self.norm(x, bn, relu)
        self.materialize(n)
        y = self.act(x.N, x.H, x.W, x.C, name)
        x.uses += 1
        big = self.use_f16x3 and y.M >= self.bf16x6_min_rows
        if self.training and self.fuse_op_stats:
            # the statistics of y (the first Bottleneck's BatchNorm reads it next) and the bound of its raw consumer
            # (that Bottleneck's skip projection) ride in the same pass
            tiles = (y.M + 127) // 128
            part = self.empty(tiles, 2, x.C)
            y.amax_tail = BnTail() if big else None
            self.f('dsnt_bn_act_fwd_stats', x.buf, n.scale, n.shift, 1 if n.relu else 0, y.buf, part, x.M, x.C, y.amax_tail)
            y.stats = (part, tiles)
        elif not self.training and big:
            y.amax_tail = BnTail()
            self.f('dsnt_bn_act_fwd_stats', x.buf, n.scale, n.shift, 1 if n.relu else 0, y.buf, None, x.M, x.C, y.amax_tail)
        else:
            self.f('dsnt_bn_act_fwd', x.buf, n.scale, n.shift, 1 if n.relu else 0, y.buf, x.M, x.C)
        if self.record:
            def backward():
                self._norm_backward(n, y.grad)
            self.on_backward(backward)
        return y

    def maxpool2(self, x, name=''):
        y = self.act(x.N, x.H // 2, x.W // 2, x.C, name)
        x.uses += 1
        idx = self.empty(x.N, x.H // 2, x.W // 2, x.C, dtype=torch.uint8)
        if self.training and self.fuse_op_stats:
            # the consumer is a BatchNorm (hourglass.py:33): its batch statistics ride in the same pass
            tiles = (y.M + 127) // 128
            part = self.empty(tiles, 2, x.C)
            y.amax_tail = BnTail() if (self.use_f16x3 and y.M >= self.bf16x6_min_rows) else None
            self.f('dsnt_maxpool2_fwd_stats', x.buf, y.buf, idx, part, x.N, x.H, x.W, x.C, y.amax_tail)
            y.stats = (part, tiles)
        elif not self.training and self.use_f16x3 and y.M >= self.bf16x6_min_rows:
            y.amax_tail = BnTail()        # no statistics in eval mode, but the next convolution's operand bound
            self.f('dsnt_maxpool2_fwd_stats', x.buf, y.buf, idx, None, x.N, x.H, x.W, x.C, y.amax_tail)
        else:
            self.f('dsnt_maxpool2_fwd', x.buf, y.buf, idx, x.N, x.H, x.W, x.C)
        if self.record:
            def backward():
                add, x.pending_add = x.pending_add, None
                buf, acc = self.grad_target(x, amax=True, take_add=True)
                if add is not None:
                    # a second gradient of x that arrived in a buffer of its own (the skip branch's: Hourglass._level) rides along
                    assert acc == 1
                    self.b('dsnt_maxpool2_bwd_add', y.grad, idx, buf, acc, add[0], x.N, x.H, x.W, x.C, x.grad_amax)
                elif x.grad_amax is not None:
                    self.b('dsnt_maxpool2_bwd_amax', y.grad, idx, buf, acc, x.N, x.H, x.W, x.C, x.grad_amax)
                else:
                    self.b('dsnt_maxpool2_bwd', y.grad, idx, buf, acc, x.N, x.H, x.W, x.C)
            self.on_backward(backward)
        return y

    def maxpool3s2(self, x, name=''):
        """3x3 / stride 2 / pad 1 max-pool (the torchvision ResNet stem)."""
        Ho, Wo = (x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1
        y = self.act(x.N, Ho, Wo, x.C, name)
        x.uses += 1
        idx = self.empty(x.N, Ho, Wo, x.C, dtype=torch.uint8)
        self.f('dsnt_maxpool3s2_fwd', x.buf, y.buf, idx, x.N, x.H, x.W, x.C)
        if self.record:
            def backward():
                buf, acc = self.grad_target(x)
                self.b('dsnt_maxpool3s2_bwd', y.grad, idx, buf, acc, x.N, x.H, x.W, x.C)
            self.on_backward(backward)
        return y

    def bn_add_act(self, x, bn, skip, relu=True, name=''):
        """y = relu(bn(x) + skip): the tail of a torchvision residual block (conv -> bn -> += identity -> relu)."""
        n = self.norm(x, bn, relu=False)
        y = self.act(x.N, x.H, x.W, x.C, name)
        x.uses += 1
        skip.uses += 1
        if os.environ.get('DSNT_DEBUG_NO_RELU'):
            relu = False
        self.materialize(n)
        self.f('dsnt_bn_add_act_fwd', x.buf, n.scale, n.shift, skip.buf, 1 if relu else 0, y.buf, x.M, x.C)
        if self.record:
            def backward():
                # the block input's gradient can START as dz and continue out of place (below): dz is then handed out as a `base`
                # that later launches — possibly of another lane, possibly held back — still read, so it gets a buffer of its own
                # (a shared scratch would be overwritten by the next block's backward with nothing ordering its readers first)
                share = bool(self.share_grads and skip._grad is None and skip.base is None and skip.pending_apply is None and
                             skip.pending_add is None)
                if relu:
                    # the ReLU mask (from the stored y), dz and the BatchNorm's two reductions in ONE pass
                    dz = (self.empty(x.M * x.C) if share else self.scratch('dz_tail', x.M * x.C).view(-1)[:x.M * x.C])
                    tiles = (x.M + 127) // 128
                    part = self.scratch('bnpart', tiles * 2 * bn.C).view(-1)
                    self.b('dsnt_bn_add_act_bwd_reduce', y.grad, y.buf, x.buf, n.mean, n.invstd, 1, dz, part, x.M, bn.C)
                    self._norm_backward(n, dz, reduced=(part, tiles))
                else:
                    dz = y.grad
                    self._norm_backward(n, dz)
                if share and skip._grad is None and skip.base is None:
                    # the block input's gradient starts as dz and continues out of place (the data gradient of conv1 adds it as its
                    # residual operand): no copy
                    skip.base, skip.base_amax = dz, None
                else:
                    self.grad_identity(skip, dz, donate=False)
            self.on_backward(backward)
        return y

    def upsample2_add(self, up, low, name=''):
        out = self.act(up.N, up.H, up.W, up.C, name)
        up.uses += 1
        low.uses += 1
        out.gives_away = True                   # dL/d out's buffer becomes dL/d up in backward
        if self.training and self.fuse_op_stats:
            tiles = (out.M + 127) // 128
            part = self.empty(tiles, 2, up.C)
            out.amax_tail = BnTail() if (self.use_f16x3 and out.M >= self.bf16x6_min_rows) else None
            self.f('dsnt_upsample2_add_fwd_stats', up.buf, low.buf, out.buf, part, up.N, up.H, up.W, up.C, out.amax_tail)
            out.stats = (part, tiles)
        elif not self.training and self.use_f16x3 and out.M >= self.bf16x6_min_rows:
            out.amax_tail = BnTail()
            self.f('dsnt_upsample2_add_fwd_stats', up.buf, low.buf, out.buf, None, up.N, up.H, up.W, up.C, out.amax_tail)
        else:
            self.f('dsnt_upsample2_add_fwd', up.buf, low.buf, out.buf, up.N, up.H, up.W, up.C)
        if self.record:
            point = self.wgrad_lane is not None and low.M <= self.release_rows
            self._release_total += point

            def backward():
                if point:
                    self._release_left -= 1
                    self.release_wgrads()
                buf, acc = self.grad_target(low, amax=True)
                if low.grad_amax is not None:
                    self.b('dsnt_upsample2_bwd_amax', out.grad, buf, acc, up.N, up.H, up.W, up.C, low.grad_amax)
                else:
                    self.b('dsnt_upsample2_bwd', out.grad, buf, acc, up.N, up.H, up.W, up.C)
                self.grad_identity(up, out.grad, donate=True, g_amax=out.grad_amax)
            self.on_backward(backward)
        return out

    def to_planar(self, x, C_logical, name='', out=None, gin=None):
        """NHWC activation -> logical NCHW tensor [N, C, H, W] (model surface); returns the tensor.
        In backward the incoming NCHW gradient is transposed into x.grad (first writer).
        out / gin: the caller's buffers (slices of one slab when a model has several outputs of one shape)."""
        out = self.empty(x.N, C_logical, x.H, x.W) if out is None else out
        x.uses += 1
        if gin is None:
            gin = self.empty(x.N, C_logical, x.H, x.W) if self.record else None
        self.f('dsnt_nhwc_to_nchw', x.buf, out, x.N, C_logical, x.H * x.W, x.C)
        if self.record:
            def backward():
                assert x.grad is None, 'planar output must be the first gradient writer'
                buf, _ = self.grad_target(x)
                self.b('dsnt_nchw_to_nhwc', gin, buf, x.N, C_logical, x.H * x.W, x.C)
            self.on_backward(backward)
        return out, gin

    def from_planar(self, src_nchw, Cpad, name=''):
        """Logical NCHW input [N, C, H, W] -> NHWC Act with channels zero-padded to Cpad."""
        N, Cc, H, W = src_nchw.shape
        a = self.act(N, H, W, Cpad, name)
        e = self.f('dsnt_nchw_to_nhwc', src_nchw, a.buf, N, Cc, H * W, Cpad)
        self._planar_src[id(a)] = (src_nchw, e)
        return a

    def stem_s2d(self, x, p, name='stem'):
        """The 7x7 / stride 2 / pad 3 convolution of a planar (<= 4 channel) input as a 4x4 / stride 1 / pad 1 convolution
        on its space-to-depth form (csrc/elementwise.hip: dsnt_s2d_input / dsnt_s2d_weights): K = 256 in 16-channel steps,
        so the stem runs on the split-precision matrix-core kernels like every other large convolution instead of the
        fp32 MFMA (245 -> ~80 us at batch 32).  Returns None if it does not apply (the caller then uses `conv`)."""
        got = self._planar_src.get(id(x))
        if (got is None or not self.use_f16x3 or not self.stem_s2d_on or p.R != 7 or p.S != 7 or p.stride != 2 or p.pad != 3
                or p.dil != 1 or x.C != 4 or x.H % 2 or x.W % 2 or x.N * (x.H // 2) * (x.W // 2) < self.bf16x6_min_rows):
            return None
        src, entry = got
        self.fwd.remove(entry)                      # the NHWC copy of the image is not needed
        xs = self.act(x.N, x.H // 2 + 1, x.W // 2 + 1, 16, 'input_s2d')
        xs.amax_tail = BnTail()
        self.f('dsnt_s2d_input', src, xs.buf, x.N, src.shape[1], x.H, x.W, xs.amax_tail)
        n = p.Cout * 256
        w2, gw2 = self.empty(p.Cout, 4, 4, 16), self.empty(p.Cout, 4, 4, 16)
        p2 = ConvParams(w2, p.b, gw2, p.gb, 1, 1, 1)
        p2.wq, p2.wq_stride = self.empty(3 * n, dtype=torch.bfloat16), n
        p2.wq16, p2.wb = self.empty(2 * n, dtype=torch.float16), self.empty(64)
        p2.post_reduce = ('dsnt_s2d_weights', (gw2, p.gw, p.Cout, 1))
        # the re-packed filter and its planes: one tiny launch on the MAIN lane right before the convolution (the table-driven
        # preparation of all other weights runs on the side lane beside it)
        self.f('dsnt_s2d_weights_prep', p.w, w2, p2.wq16, p2.wq, p2.wb, p.Cout)
        self._f16_w_seen.add(p2.wq16.data_ptr())    # (prepared here, not by the table-driven launch)
        mark = len(self.fwd)
        y = self.conv(xs, p2, want_stats=True, need_input_grad=False, name=name)
        self._prep_exempt.update(id(e) for e in self.fwd[mark:])
        return y
