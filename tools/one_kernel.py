"""Run only the dominant conv kernel a few times (for rocprofv3 --pmc)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr, ConvGeom
dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
B = 32
H, Cin, Cout, k = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (64, 128, 128, 3)
g = ConvGeom(B, H, H, Cin, H, H, Cout, k, k, 1, k // 2, 1)
x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
b = torch.zeros(Cout, device=dev); sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
y = torch.empty(B, H, H, Cout, device=dev); gy = torch.randn(B, H, H, Cout, device=dev)
M = B * H * H
stats = torch.empty((M + 63) // 64, 2, Cout, device=dev)     # (room for 64-row tiles: kernel experiments)
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(max(_lib.fn('dsnt_conv_wgrad_ws_floats')(C.byref(g)), _lib.fn('dsnt_conv_wgrad_f16x3_ws_floats')(C.byref(g), 0)), device=dev); dw = torch.empty_like(w); db = torch.empty(Cout, device=dev)
planes = torch.empty(3 * w.numel(), dtype=torch.bfloat16, device=dev)
_lib.fn('dsnt_split_bf16x3')(ptr(w), ptr(planes), w.numel(), st)
planes16 = torch.empty(2 * w.numel(), dtype=torch.float16, device=dev)
wb, ab, gb = torch.zeros(64, device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev)
_lib.fn('dsnt_amax')(ptr(w), w.numel(), ptr(wb), st)
_lib.fn('dsnt_split_f16x2')(ptr(w), ptr(planes16), w.numel(), w.numel(), ptr(wb), st)
_lib.fn('dsnt_amax')(ptr(gy), gy.numel(), ptr(gb), st)
ab.fill_(float(torch.relu(x * sc + sh).max()) * 4.0)
planes16s = torch.empty(2 * w.numel(), dtype=torch.float16, device=dev)
if k == 3 and _lib.fn('dsnt_conv_fwd_stream_ok')(C.byref(g)):
    tab = torch.tensor([[w.data_ptr(), planes16s.data_ptr(), wb.data_ptr(), w.numel(), w.numel(), Cout, Cin]], dtype=torch.int64).to(dev)
    _lib.fn('dsnt_f16_prep_weights')(ptr(tab), 1, 7, st)
reps = int(os.environ.get('ONE_KERNEL_REPS', '5'))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
for it in range(reps):
    ev[it].record()
    if which == 'fwd6':
        _lib.fn('dsnt_conv_fwd_bf16x6')(ptr(x), ptr(planes), w.numel(), ptr(b), ptr(y), ptr(sc), ptr(sh), 1, None, None, ptr(stats), C.byref(g), st)
    elif which == 'fwd16':
        _lib.fn('dsnt_conv_fwd_f16x3_ex')(ptr(x), ptr(planes16), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y), ptr(sc), ptr(sh), 1, None, None, ptr(stats), C.byref(g), None, None, st)
    elif which == 'fwd16s':
        _lib.fn('dsnt_conv_fwd_f16x3_stream')(ptr(x), ptr(planes16s), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y), ptr(sc), ptr(sh), 1, None, None, ptr(stats), C.byref(g), None, None, st)
    elif which == 'wgrad16':
        _lib.fn('dsnt_conv_wgrad_f16x3')(ptr(x), ptr(sc), ptr(sh), 1, ptr(gy), ptr(ws), None, None, 0, ptr(ab), ptr(gb), C.byref(g), st)
    elif which == 'wgrad6':
        _lib.fn('dsnt_conv_wgrad_bf16x6')(ptr(x), ptr(sc), ptr(sh), 1, ptr(gy), ptr(ws), None, None, 0, C.byref(g), st)
    elif which == 'fwd':
        _lib.fn('dsnt_conv_fwd')(ptr(x), ptr(w), ptr(b), ptr(y), ptr(sc), ptr(sh), 1, None, None, ptr(stats), C.byref(g), st)
    else:
        _lib.fn('dsnt_conv_wgrad')(ptr(x), ptr(sc), ptr(sh), 1, ptr(gy), ptr(ws), ptr(dw), ptr(db), 0, C.byref(g), st)
ev[reps].record()
torch.cuda.synchronize()
print(which, H, Cin, Cout, k, 'us per launch:', ' '.join('%.1f' % (1e3 * ev[i].elapsed_time(ev[i + 1])) for i in range(reps)))
