"""Time dsnt_conv_wgrad_f16x3 alone (slabs only): python tools/bench_wgrad.py [H Cin Cout k [B]].
DSNT_OFF=wgrad3 routes 3x3 shapes through the implicit-GEMM kernel instead of the halo kernel (A/B)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr, ConvGeom
dev = torch.device('cuda:0')
H, Cin, Cout, k = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (64, 128, 128, 3)
B = int(sys.argv[5]) if len(sys.argv) > 5 else 32
g = ConvGeom(B, H, H, Cin, H, H, Cout, k, k, 1, k // 2, 1)
x = torch.randn(B, H, H, Cin, device=dev)
sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
gy = torch.randn(B, H, H, Cout, device=dev)
st = torch.cuda.current_stream().cuda_stream
nws = _lib.fn('dsnt_conv_wgrad_f16x3_ws_floats')(C.byref(g), 0)
ws = torch.empty(nws, device=dev)
ab, gb = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
_lib.fn('dsnt_amax')(ptr(gy), gy.numel(), ptr(gb), st)
ab.fill_(float(torch.relu(x * sc + sh).max()) * 4.0)
f = _lib.fn('dsnt_conv_wgrad_f16x3')
def run():
    e = f(ptr(x), ptr(sc), ptr(sh), 1, ptr(gy), ptr(ws), None, None, 0, ptr(ab), ptr(gb), C.byref(g), st)
    assert e == 0, _lib.load().dsnt_last_error()
for _ in range(5):
    run()
torch.cuda.synchronize()
n = 30
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / n
fl = 2.0 * B * H * H * k * k * Cin * Cout
print('wgrad %dx%d %d->%d k%d B%d halo=%d slabs=%d ws=%.1f MB: %.1f us  %.1f TFLOP/s (algorithmic)' % (
    H, H, Cin, Cout, k, B, _lib.fn('dsnt_conv_wgrad_halo_ok')(C.byref(g)),
    _lib.fn('dsnt_conv_wgrad_f16x3_splits')(C.byref(g), 0), nws * 4 / 1e6, us, fl / us / 1e6))
