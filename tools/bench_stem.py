"""The stem of the hourglass as the engine runs it (4x4 / stride 1 / pad 1 convolution of the 16-channel space-to-depth image,
hourglass.py:157 `conv1`): forward and weight gradient alone, HIP events.  python3 tools/bench_stem.py [batch]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr, ConvGeom
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, Cin, Cout, k = 129, 16, 64, 4
g = ConvGeom(B, H, H, Cin, 128, 128, Cout, k, k, 1, 1, 1)
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, H, H, Cin, device=dev)
w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
b = torch.zeros(Cout, device=dev)
y = torch.empty(B, 128, 128, Cout, device=dev)
gy = torch.randn(B, 128, 128, Cout, device=dev) * 1e-3
M = B * 128 * 128
stats = torch.empty((M + 127) // 128, 2, Cout, device=dev)
wq16 = torch.empty(2 * w.numel(), dtype=torch.float16, device=dev)
wb, ab, gb = torch.zeros(64, device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev)
for t, s in ((w, wb), (x, ab), (gy, gb)):
    assert _lib.fn('dsnt_amax')(ptr(t), t.numel(), ptr(s), st) == 0
assert _lib.fn('dsnt_split_f16x2')(ptr(w), ptr(wq16), w.numel(), w.numel(), ptr(wb), st) == 0
fwd = _lib.fn('dsnt_conv_fwd_f16x3_ex')
wg = _lib.fn('dsnt_conv_wgrad_f16x3')
ws = torch.empty(_lib.fn('dsnt_conv_wgrad_f16x3_ws_floats')(C.byref(g), 0), device=dev)
def timed(fn, n=10):
    for _ in range(3):
        assert fn() == 0
    e = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    e[0].record()
    for i in range(n):
        fn(); e[i + 1].record()
    torch.cuda.synchronize()
    return sorted(e[i].elapsed_time(e[i + 1]) * 1e3 for i in range(n))[n // 2]
f4 = None
if _lib.fn('dsnt_stem4_fwd_ok')(C.byref(g)):
    rows = _lib.fn('dsnt_stem4_fwd_stats_rows')(C.byref(g))
    stats4 = torch.empty(rows, 2, Cout, device=dev)
    y4 = torch.empty_like(y)
    s4 = _lib.fn('dsnt_stem4_fwd_f16x3')
    f4 = timed(lambda: s4(ptr(x), ptr(wq16), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y4), ptr(stats4), C.byref(g), None, st))
f = timed(lambda: fwd(ptr(x), ptr(wq16), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y), None, None, 0, None, None, ptr(stats), C.byref(g), None, None, st))
wgt = timed(lambda: wg(ptr(x), None, None, 0, ptr(gy), ptr(ws), None, None, 0, ptr(ab), ptr(gb), C.byref(g), st))
# the weight gradient against torch (fp64) on a slice of the batch
dw, db = torch.empty_like(w), torch.empty(Cout, device=dev)
assert wg(ptr(x), None, None, 0, ptr(gy), ptr(ws), ptr(dw), ptr(db), 0, ptr(ab), ptr(gb), C.byref(g), st) == 0
torch.cuda.synchronize()
ref = torch.nn.grad.conv2d_weight(x.double().permute(0, 3, 1, 2), (Cout, Cin, k, k), gy.double().permute(0, 3, 1, 2), stride=1, padding=1)
ref = ref.permute(0, 2, 3, 1)
print('weight gradient: %d slabs; max |dW - fp64| = %.3e of max |dW| %.3e; max |db - fp64| = %.3e of %.3e'
      % (_lib.fn('dsnt_conv_wgrad_f16x3_splits')(C.byref(g), 0), (dw.double() - ref).abs().max().item(), ref.abs().max().item(),
         (db.double() - gy.double().sum((0, 1, 2))).abs().max().item(), gy.double().sum((0, 1, 2)).abs().max().item()))
mb_f = 4e-6 * (x.numel() + y.numel())
mb_w = 4e-6 * (x.numel() + gy.numel())
if f4 is not None:
    torch.cuda.synchronize()
    print('stem4 forward %6.1f us (%.2f TB/s); max |y - y_tiled| = %.3e (max |y| %.3e); column sums: %.3e vs %.3e'
          % (f4, mb_f / f4, (y4 - y).abs().max().item(), y.abs().max().item(),
             stats4[:, 0].sum().item(), stats[:, 0].sum().item()))
print('stem forward  %6.1f us (%.0f MB: %.2f TB/s)   weight gradient %6.1f us (%.0f MB + %.1f MB of slabs: %.2f TB/s)'
      % (f, mb_f, mb_f / f, wgt, mb_w, ws.numel() * 4e-6, mb_w / wgt))
