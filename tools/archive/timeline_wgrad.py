"""Wave timeline of the unified bf16x6 weight-gradient kernel (needs a DSNT_TIMELINE=1 build:
DSNT_TIMELINE=1 python dsnt-pose2d_amd/build.py --force).  Per 16-row step every wave stamps s_memtime at
1+3s (step start = barrier released), 2+3s (fragments read, staging of the next step and the 24 MFMAs issued),
3+3s (prefetch issued); slot 0 = kernel start, 125 = loop end.  Usage: python tools/timeline_wgrad.py [k] [Cin] [Cout] [H]"""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr, ConvGeom
dev = torch.device('cuda:0')
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
Cin = int(sys.argv[2]) if len(sys.argv) > 2 else 128
Cout = int(sys.argv[3]) if len(sys.argv) > 3 else 128
H = int(sys.argv[4]) if len(sys.argv) > 4 else 64
B = 32
g = ConvGeom(B, H, H, Cin, H, H, Cout, k, k, 1, k // 2, 1)
x = torch.randn(B, H, H, Cin, device=dev); gy = torch.randn(B, H, H, Cout, device=dev)
sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
ws = torch.empty(_lib.fn('dsnt_conv_wgrad_ws_floats')(C.byref(g)), device=dev)
st = torch.cuda.current_stream().cuda_stream
fn = _lib.fn('dsnt_conv_wgrad_bf16x6')
args = (ptr(x), ptr(sc), ptr(sh), 1, ptr(gy), ptr(ws), None, None, 0, C.byref(g))
for _ in range(3):
    assert fn(*args, st) == 0
torch.cuda.synchronize()
splits = _lib.fn('dsnt_conv_wgrad_splits')(C.byref(g))
nwg = splits * ((k * k * Cin + 127) // 128) * ((Cout + 127) // 128)
buf = torch.zeros(nwg * 1024, dtype=torch.int64, device=dev)
assert _lib.fn('dsnt_debug_set_timeline')(ptr(buf), -1) == 0
fn(*args, st); torch.cuda.synchronize()
_lib.fn('dsnt_debug_set_timeline')(None, 0)
t = buf.cpu().numpy().reshape(nwg, 8, 128)[:, :4, :]
if not t[:, :, 0].any():
    sys.exit('no stamps: rebuild with DSNT_TIMELINE=1 python dsnt-pose2d_amd/build.py --force')
t0 = t[:, :, 0][t[:, :, 0] > 0].min()
print('workgroups %d, kernel span %d cycles, loop span mean %d' % (nwg, (t[:, :, 125] - t0).max(), (t[:, :, 125] - t[:, :, 1]).mean()))
nst = 41
sl = np.arange(nst)
a = t[:, :, 1 + 3 * sl].astype(np.int64); b = t[:, :, 2 + 3 * sl].astype(np.int64); c = t[:, :, 3 + 3 * sl].astype(np.int64)
def stat(v, waves):
    v = v[:, waves, 4:]
    return '%7.0f (p10 %5.0f p90 %5.0f)' % (v.mean(), np.percentile(v, 10), np.percentile(v, 90))
for name, waves in (('A-staging waves 0,1', [0, 1]), ('dY-staging waves 2,3', [2, 3])):
    print(name)
    print('  step period            ', stat(np.diff(a, axis=2), waves))
    print('  rd + stage + 24 MFMA   ', stat((b - a)[:, :, :-1], waves))
    print('  prefetch issue         ', stat((c - b)[:, :, :-1], waves))
    print('  barrier wait           ', stat(a[:, :, 1:] - c[:, :, :-1], waves))
i = nwg // 2
for s in range(8, 12):
    print('  wg %d step %d: ' % (i, s) + ' | '.join('w%d start %d body %d pf %d' % (w, a[i, w, s] - t0, b[i, w, s] - a[i, w, s], c[i, w, s] - b[i, w, s]) for w in range(4)))
