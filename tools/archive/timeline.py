import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr, ConvGeom
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
block = int(sys.argv[2]) if len(sys.argv) > 2 else 300
H, Cin, Cout, k = 64, 128, 128, 3
g = ConvGeom(B, H, H, Cin, H, H, Cout, k, k, 1, 1, 1)
x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
b = torch.zeros(Cout, device=dev); sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
y = torch.empty(B, H, H, Cout, device=dev)
M = B * H * H
stats = torch.empty((M + 127) // 128, 2, Cout, device=dev)
st = torch.cuda.current_stream().cuda_stream
fn = _lib.fn('dsnt_conv_fwd')
args = (ptr(x), ptr(w), ptr(b), ptr(y), ptr(sc), ptr(sh), 1, None, None, ptr(stats), C.byref(g))
for _ in range(3): fn(*args, st)
torch.cuda.synchronize()
buf = torch.zeros(8 * 128, dtype=torch.int64, device=dev)
assert _lib.fn('dsnt_debug_set_timeline')(ptr(buf), block) == 0
fn(*args, st); torch.cuda.synchronize()
_lib.fn('dsnt_debug_set_timeline')(None, 0)
t = buf.cpu().view(8, 128)
t0 = t[t > 0].min().item()
r = (t - t0)
print('MFMA wave 0: step-start stamps (slot 1+2s), barrier-arrive (2+2s)  [cycles @100MHz memtime? raw units]')
mf = r[0]
print('start', mf[0].item(), 'end-loop', mf[120].item(), 'end', mf[121].item())
steps = [(mf[1 + 2 * s].item(), mf[2 + 2 * s].item()) for s in range(36)]
print('step durations (start->start):', [steps[i + 1][0] - steps[i][0] for i in range(35)])
print('start->barrier-arrive:', [b - a for a, b in steps])
ld = r[4]
print('loader wave 4: prologue done', ld[1].item())
print('loader lstore dur (even s):', [ld[3 + 3 * s].item() - ld[2 + 3 * s].item() for s in range(0, 34, 2)])
print('loader gload dur (even s):', [ld[4 + 3 * s].item() - ld[3 + 3 * s].item() for s in range(0, 34, 2)])
print('loader barrier wait (even s):', [ld[5 + 3 * s].item() - ld[4 + 3 * s].item() for s in range(0, 34, 2)])
print('loader 2nd half lstore:', [ld[6 + 3 * s].item() - ld[5 + 3 * s].item() for s in range(0, 34, 2)])
print('loader 2nd half gload:', [ld[7 + 3 * s].item() - ld[6 + 3 * s].item() for s in range(0, 32, 2)])
