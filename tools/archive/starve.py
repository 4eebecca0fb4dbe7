"""Does a saturated matrix pipe starve VALU issue on the same SIMD?  (dsnt_debug_starve)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr
dev = torch.device('cuda:0')
blocks = 256
out = torch.empty(blocks * 512, device=dev)
cyc = torch.zeros(blocks * 8, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
fn = _lib.fn('dsnt_debug_starve')
iters = 400
for valu_n in (8, 64, 1024):
    for mfma_iters in (0, iters):
        for pad in ((0, 1, 2, 3, 4, 5, 6, 7) if mfma_iters else (0,)):
            for prio in (0,):
                for _ in range(2):
                    assert fn(ptr(out), ptr(cyc), blocks, mfma_iters, valu_n, prio | (pad << 4), st) == 0
                torch.cuda.synchronize()
                c = cyc[:blocks * 4].float(); m = cyc[blocks * 4:].float()
                print('valu %4d instr | mfma %-4s pad %d | prio %d | burst %7.0f cycles = %6.1f cycles/instr | mfma wave %.1f cycles/mfma'
                      % (valu_n * 16, 'busy' if mfma_iters else 'idle', pad, prio, c.mean(), c.mean() / (valu_n * 16),
                         m.mean() / (16 * iters) if mfma_iters else 0))
