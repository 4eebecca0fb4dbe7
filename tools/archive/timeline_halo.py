"""Wave timeline of the fp16x3 halo-tile 3x3 kernel (needs a DSNT_TIMELINE=1 build, e.g.
DSNT_TIMELINE=1 DSNT_LIB_NAME=libdsnt_tl.so python dsnt-pose2d_amd/build.py; DSNT_HIP_LIB=.../libdsnt_tl.so python tools/timeline_halo.py).
Stamps (first 18 K-steps of every workgroup): 1+3j step start, 2+3j work issued (MFMA waves: the 12 MFMAs; loaders: staging
+ loads), 3+3j through the barrier; 0 kernel start, 124 / 125 around the epilogue."""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr, ConvGeom
dev = torch.device('cuda:0')
B, H, Cin, Cout, k = 32, 64, 128, 128, 3
g = ConvGeom(B, H, H, Cin, H, H, Cout, k, k, 1, 1, 1)
x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
b = torch.zeros(Cout, device=dev); sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
y = torch.empty(B, H, H, Cout, device=dev)
M = B * H * H
stats = torch.empty((M + 127) // 128, 2, Cout, device=dev)
st = torch.cuda.current_stream().cuda_stream
wq16 = torch.empty(2 * w.numel(), dtype=torch.float16, device=dev)
wb, ab = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
assert _lib.fn('dsnt_amax')(ptr(w), w.numel(), ptr(wb), st) == 0
assert _lib.fn('dsnt_split_f16x2')(ptr(w), ptr(wq16), w.numel(), w.numel(), ptr(wb), st) == 0
ab.fill_(float(torch.relu(x * sc + sh).max()) * 4.0)
fn = _lib.fn('dsnt_conv_fwd_f16x3_ex')
args = (ptr(x), ptr(wq16), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y), ptr(sc), ptr(sh), 1, None, None, ptr(stats), C.byref(g), None, None)
for _ in range(3):
    assert fn(*args, st) == 0
torch.cuda.synchronize()
nwg = (M // 128) * ((Cout + 127) // 128)
buf = torch.zeros(nwg * 1024, dtype=torch.int64, device=dev)
assert _lib.fn('dsnt_debug_set_timeline')(ptr(buf), -1) == 0
fn(*args, st); torch.cuda.synchronize()
_lib.fn('dsnt_debug_set_timeline')(None, 0)
t = buf.cpu().numpy().reshape(nwg, 8, 128)
if not t[:, :, 0].any():
    sys.exit('no stamps: build with DSNT_TIMELINE=1')
t0 = t[:, :, 0][t[:, :, 0] > 0].min()
start = t[:, 0, 0] - t0
first = start < np.percentile(start, 40)            # workgroups of the first round
sl = np.arange(18)
def stat(a):
    a = a[first][:, :, 3:]
    return '%7.0f (p10 %5.0f p50 %5.0f p90 %5.0f)' % (a.mean(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90))
for name, waves in (('MFMA waves', slice(0, 4)), ('loader waves', slice(4, 8))):
    q = t[:, waves, :] - t0
    s0, s1, s2 = q[:, :, 1 + 3 * sl], q[:, :, 2 + 3 * sl], q[:, :, 3 + 3 * sl]
    print(name, '(cycles per K-step, s_memtime units):')
    print('  step period        ', stat(np.diff(s0, axis=2)))
    print('  work (start->issued)', stat((s1 - s0)[:, :, :-1]))
    print('  barrier wait        ', stat((s2 - s1)[:, :, :-1]))
    print('  barrier -> next     ', stat((s0[:, :, 1:] - s2[:, :, :-1])))
dur = t[:, 0, 125] - t[:, 0, 0]
pro = t[:, 0, 1] - t[:, 0, 0]; epi = t[:, 0, 125] - t[:, 0, 124]; loop18 = t[:, 0, 3 + 3 * 17] - t[:, 0, 1]
print('workgroup lifetime mean %d p10 %d p90 %d; prologue %d; first 18 steps %d (x4 = %d); epilogue %d' % (
    dur.mean(), np.percentile(dur, 10), np.percentile(dur, 90), pro.mean(), loop18.mean(), 4 * loop18.mean(), epi.mean()))
end = t[:, 0, 125] - t0
print('kernel span %d; start pct (0,25,50,75,100) %s; end pct %s' % (end.max(), [int(np.percentile(start, q)) for q in (0, 25, 50, 75, 100)],
                                                                   [int(np.percentile(end, q)) for q in (0, 25, 50, 75, 100)]))
# one workgroup in detail
i = int(np.where(first)[0][5])
for i in [int(v) for v in np.where(first)[0][5:8]]:
    base = t[i, :, 1 + 3 * 4].min()
    for j in range(4, 12):
        # arrival at the barrier (stamp 2) relative to the step's earliest start, and barrier exit (stamp 3)
        arr = [int(t[i, wv, 2 + 3 * j] - base) for wv in range(8)]
        ext = [int(t[i, wv, 3 + 3 * j] - base) for wv in range(8)]
        st_ = [int(t[i, wv, 1 + 3 * j] - base) for wv in range(8)]
        print('  wg%d step %2d start %s | arrive %s | exit %s | last arriver w%d' % (i, j, st_, arr, ext, int(np.argmax(arr))))
# who arrives last, over all first-round workgroups and steps 3..17
q = t[first][:, :, 2 + 3 * np.arange(3, 18)]
last = np.argmax(q, axis=1)
print('last arriver histogram (waves 0-3 MFMA, 4-7 loaders):', np.bincount(last.ravel(), minlength=8).tolist())
lag = np.sort(q, axis=1)
print('arrival spread: last - first %d, last - second-last %d (mean cycles)' % ((lag[:, -1] - lag[:, 0]).mean(), (lag[:, -1] - lag[:, -2]).mean()))
