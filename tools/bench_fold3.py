"""Time the 3x3 data gradient with the BatchNorm backward folded into its operand load (dsnt_conv_dgrad_f16x3_stream_apply,
csrc/conv3s.hip MODE 4) against the two launches it replaces (dsnt_bn_act_bwd_apply + dsnt_conv_fwd_f16x3_stream with the
BatchNorm-backward epilogue).  usage: python tools/bench_fold3.py [N H W C]   (default 32 64 64 128)"""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
import torch  # noqa: E402
from dsnt import _lib  # noqa: E402
from dsnt._lib import ptr, call, ConvGeom, BnBwdEpilogue, BnBwdApply, BnTail  # noqa: E402

N, H, W, Cc = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (32, 64, 64, 128)
dev = torch.device('cuda:0')
g = ConvGeom(N, H, W, Cc, H, W, Cc, 3, 3, 1, 1, 1)
M = N * H * W
torch.manual_seed(0)
w = (torch.randn(Cc, 3, 3, Cc, device=dev) * 0.03).contiguous()
n = w.numel()
strm = torch.empty(2 * n, dtype=torch.float16, device=dev)
wb = torch.zeros(64, device=dev)
table = torch.tensor([[w.data_ptr(), strm.data_ptr(), wb.data_ptr(), n, n, Cc, Cc]], dtype=torch.int64).to(dev)
call('dsnt_f16_prep_weights', ptr(table), 1, 7)
y = torch.randn(N, H, W, Cc, device=dev)
dz = torch.randn(N, H, W, Cc, device=dev) * (torch.rand(N, H, W, Cc, device=dev) > 0.5)
mu, invstd, scale, shift = torch.randn(Cc, device=dev) * 0.1, torch.rand(Cc, device=dev) + 0.5, torch.rand(Cc, device=dev) + 0.5, torch.zeros(Cc, device=dev)
coef = torch.randn(2 * Cc, device=dev) * 0.01
xin = torch.randn(N, H, W, Cc, device=dev)
bnb = BnBwdEpilogue(ptr(xin), ptr(scale), ptr(shift), ptr(mu), ptr(invstd), 1)
dy = torch.empty(N, H, W, Cc, device=dev)
dyo = torch.empty(N, H, W, Cc, device=dev)
out = torch.empty(N, H, W, Cc, device=dev)
stats = torch.empty(M // 128, 2, Cc, device=dev)
ab = torch.full((64,), 64.0, device=dev)
amax = torch.zeros(64, device=dev)
tail = BnTail()
tail.amax = amax.data_ptr()
ap = BnBwdApply(ptr(y), ptr(scale), ptr(mu), ptr(invstd), ptr(coef))


def apply_():
    call('dsnt_bn_act_bwd_apply_amax', ptr(dz), ptr(y), ptr(scale), ptr(shift), ptr(mu), ptr(invstd), ptr(coef), 0, ptr(dy), 0, M, Cc, ptr(amax))


def stream():
    call('dsnt_conv_fwd_f16x3_stream', ptr(dy), ptr(strm), n, ptr(wb), ptr(ab), None, ptr(out), None, None, 0, None, None, ptr(stats),
         C.byref(g), C.byref(bnb), C.byref(tail))


def folded():
    call('dsnt_conv_dgrad_f16x3_stream_apply', ptr(dz), C.byref(ap), ptr(dyo), ptr(strm), n, ptr(wb), ptr(ab), ptr(out), ptr(stats), 0,
         C.byref(g), C.byref(bnb), C.byref(tail))


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        s.append(e0.elapsed_time(e1) * 1e3)
        time.sleep(1e-3)
    s.sort()
    return s[len(s) // 2]


bias = torch.zeros(Cc, device=dev)


def forward():      # the same kernel as a forward convolution: BatchNorm + ReLU prologue, bias, statistics (MODE 0, PRO)
    call('dsnt_conv_fwd_f16x3_stream', ptr(y), ptr(strm), n, ptr(wb), ptr(ab), ptr(bias), ptr(out), ptr(scale), ptr(mu), 1, None, None,
         ptr(stats), C.byref(g), None, C.byref(tail))


ta, ts, tf = timed(apply_), timed(stream), timed(folded)
print('forward (MODE 0, prologue) %.1f us' % timed(forward))
print('%dx%dx%dx%d: apply %.1f us + stream (MODE 3) %.1f us = %.1f us;  folded (MODE 4) %.1f us' % (N, H, W, Cc, ta, ts, ta + ts, tf))
