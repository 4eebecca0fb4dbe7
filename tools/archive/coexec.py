import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr
fn = _lib.fn('dsnt_debug_coexec')
out = torch.empty(256 * 512, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def t(mi, vi):
    fn(ptr(out), 256, max(mi // 10, 1) if mi else 0, max(vi // 10, 1) if vi else 0, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(ptr(out), 256, mi, vi, st); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
mi = 2000          # 2000*16 MFMA * 64 cyc = 2.05M cycles ~ 0.85 ms
for vi in (0, 2000, 4000, 8000, 16000):   # vi*64 v_fma * ~4 cyc issue
    print('mfma_iters %d valu_iters %5d : both %.3f ms | mfma alone %.3f | valu alone %.3f' % (mi, vi, t(mi, vi), t(mi, 0), t(0, vi) if vi else 0))
