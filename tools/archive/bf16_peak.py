import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr
fn = _lib.fn('dsnt_debug_bf16_peak')
out = torch.empty(1024 * 512, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def t(blocks, threads, mi, vi):
    fn(ptr(out), blocks, threads, max(mi // 10, 1), max(vi // 10, 0), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(ptr(out), blocks, threads, mi, vi, st); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3
mi = 8000
for blocks, threads in [(256, 256), (512, 256), (1024, 256)]:
    tt = t(blocks, threads, mi, 0)
    fl = blocks * 4 * mi * 16 * (32 * 32 * 16 * 2)
    print('bf16 32x32x16: blocks %d x %d threads: %.1f TFLOP/s' % (blocks, threads, fl / tt / 1e12))
for vi in (0, 4000, 8000, 16000):
    print('coexec (256 blocks x 512): mfma %d valu %d: both %.3f ms | mfma alone %.3f | valu alone %.3f' % (
        mi, vi, t(256, 512, mi, vi) * 1e3, t(256, 512, mi, 0) * 1e3, t(256, 512, 0, vi) * 1e3 if vi else 0))
