"""Time the DSNT head kernels alone at a large batch (HBM roofline of the path's namesake kernels):
   python tools/bench_head.py [batch]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rows, h, w = B * 16, 64, 64
logits = torch.randn(rows, h * w, device=dev) * 3
hm = torch.empty_like(logits); g0 = torch.empty_like(logits)
coords = torch.empty(rows, 2, device=dev); target = torch.rand(rows, 2, device=dev) * 2 - 1
mask = torch.ones(rows, device=dev); dist = torch.empty(rows, device=dev); reg = torch.empty(rows, device=dev)
denom2 = torch.empty(2, device=dev); gd = torch.full((rows,), 1.0 / rows, device=dev)
st = torch.cuda.current_stream().cuda_stream
assert _lib.fn('dsnt_mask_denom')(ptr(mask), ptr(denom2), rows, st) == 0
def timeit(name, fn, args, nbytes, iters=20):
    for _ in range(3): assert fn(*args, st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn(*args, st)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    print('%-28s %8.1f us  %7.1f GB/s (%4.1f%% of 8 TB/s)' % (name, t * 1e6, nbytes / t / 1e9, nbytes / t / 8e10))
    return t
one = rows * h * w * 4.0
t = 0
t += timeit('head_fwd', _lib.fn('dsnt_head_fwd'), (ptr(logits), ptr(hm), ptr(coords), rows, h, w), 2 * one)
t += timeit('head_loss_grad js', _lib.fn('dsnt_head_loss_grad'), (ptr(hm), ptr(coords), ptr(target), ptr(mask), ptr(denom2), ptr(dist), ptr(reg), ptr(g0), rows, h, w, 2.0 / 64, 0, 1.0), 2 * one)
print('train-step head, 4 passes: %.1f us -> %.1f GB/s' % (t * 1e6, 4 * one / t / 1e9))
timeit('head_loss_grad none', _lib.fn('dsnt_head_loss_grad'), (ptr(hm), ptr(coords), ptr(target), ptr(mask), ptr(denom2), ptr(dist), None, ptr(g0), rows, h, w, 2.0 / 64, -1, 1.0), 2 * one)
timeit('head_loss_rows js (old)', _lib.fn('dsnt_head_loss_rows'), (ptr(hm), ptr(coords), ptr(target), ptr(dist), ptr(reg), rows, h, w, 2.0 / 64, 0), one)
timeit('head_bwd js (old)', _lib.fn('dsnt_head_bwd'), (ptr(hm), ptr(coords), ptr(target), ptr(dist), ptr(gd), ptr(gd), ptr(g0), rows, h, w, 2.0 / 64, 0), 2 * one)
timeit('copy (torch)', lambda a, b, s: (b.copy_(a), 0)[1], (logits, hm), 2 * one)
