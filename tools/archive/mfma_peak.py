import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr
fn = _lib.fn('dsnt_debug_mfma_peak')
out = torch.empty(256 * 16 * 1024, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for blocks, threads, dep in [(256, 256, 1), (256, 256, 4), (512, 256, 1), (512, 256, 4), (1024, 256, 1), (256, 512, 1), (256, 512, 4)]:
    iters = 4000
    fn(ptr(out), blocks, threads, 100, dep, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(ptr(out), blocks, threads, iters, dep, st); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3
    flops = blocks * (threads // 64) * iters * 16 * (32 * 32 * 2 * 2)
    print('blocks %4d threads %3d dep %d: %.1f TFLOP/s  (%.2f ms)' % (blocks, threads, dep, flops / t / 1e12, t * 1e3))
