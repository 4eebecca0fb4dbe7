"""Wave timeline of the bf16x6 forward kernel (needs a DSNT_TIMELINE=1 build).

Every workgroup stamps s_memtime per K-step: MFMA waves (0-3): 1+3s step start, 2+3s fragments in
registers, 3+3s MFMAs issued; loader waves (4-7): 1+3s start, 2+3s LDS stores issued, 3+3s global loads
issued, 4+3s after the barrier, ...  Slots 126/127 hold HW_ID / XCC_ID, so co-resident workgroups
can be paired up.  Usage: python tools/timeline6.py [k] [Cin] [Cout] [H]
"""
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
x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
b = torch.zeros(Cout, device=dev); sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
y = torch.empty(B, H, H, Cout, device=dev)
M = B * H * H
stats = torch.empty((M + 127) // 128, 2, Cout, device=dev)
wq = torch.empty(3 * w.numel(), dtype=torch.int16, device=dev)
st = torch.cuda.current_stream().cuda_stream
assert _lib.fn('dsnt_split_bf16x3')(ptr(w), ptr(wq), w.numel(), st) == 0
if os.environ.get('TL_F16'):      # fp16x3 form of the same kernel
    wq16 = torch.empty(2 * w.numel(), dtype=torch.float16, device=dev)
    wb, ab = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    assert _lib.fn('dsnt_amax')(ptr(w), w.numel(), ptr(wb), st) == 0
    assert _lib.fn('dsnt_split_f16x2')(ptr(w), ptr(wq16), w.numel(), w.numel(), ptr(wb), st) == 0
    ab.fill_(float(torch.relu(x * sc + sh).max()) * 4.0)
    fn = _lib.fn('dsnt_conv_fwd_f16x3_ex')
    args = (ptr(x), ptr(wq16), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y), ptr(sc), ptr(sh), 1, None, None, ptr(stats), C.byref(g), None, None)
else:
    fn = _lib.fn('dsnt_conv_fwd_bf16x6')
    args = (ptr(x), ptr(wq), w.numel(), ptr(b), ptr(y), ptr(sc), ptr(sh), 1, None, None, ptr(stats), C.byref(g))
for _ in range(3):
    assert fn(*args, st) == 0
torch.cuda.synchronize()
nwg = ((M + 127) // 128) * ((Cout + 127) // 128)
buf = torch.zeros(nwg * 1024, dtype=torch.int64, device=dev)
assert _lib.fn('dsnt_debug_set_timeline')(ptr(buf), -1) == 0
fn(*args, st); torch.cuda.synchronize()
_lib.fn('dsnt_debug_set_timeline')(None, 0)
t = buf.cpu().numpy().reshape(nwg, 8, 128)
if not t[:, :, 0].any():
    sys.exit('no stamps: rebuild with DSNT_TIMELINE=1 python dsnt-pose2d_amd/build.py --force')
hw = t[:, 0, 126]; xcc = t[:, 0, 127] & 0xf
cu = (hw >> 8) & 0xf; sh_ = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc * 1000 + se * 100 + sh_ * 10 + cu
t0 = t[:, :, 0][t[:, :, 0] > 0].min()
start = t[:, 0, 0] - t0; end = t[:, 0, 125] - t0
print('workgroups %d, distinct CUs %d, kernel span %d cycles' % (nwg, len(set(key.tolist())), end.max()))
nst = min(41, (k * k * Cin) // 16)
nthreads = int(os.environ.get('TL_WAVES', '8'))
mf = t[:, 0:4, :] - t0; ld = (t[:, 4:8, :] if nthreads == 8 else t[:, 0:4, :]) - t0
sl = np.arange(nst)
step_start = mf[:, :, 1 + 3 * sl]; rd_done = mf[:, :, 2 + 3 * sl]; mm_done = mf[:, :, 3 + 3 * sl]
# steady-state workgroups only: started in the first wave of the launch
first = start < np.percentile(start, 40)
def stat(a):
    a = a[first][:, :, 4:]
    return '%7.0f (p10 %5.0f p90 %5.0f)' % (a.mean(), np.percentile(a, 10), np.percentile(a, 90))
print('MFMA waves, cycles per K-step:')
print('  step period      ', stat(np.diff(step_start, axis=2)))
print('  fragment read    ', stat((rd_done - step_start)[:, :, :-1]))
print('  mfma issue       ', stat((mm_done - rd_done)[:, :, :-1]))
print('  barrier wait     ', stat((step_start[:, :, 1:] - mm_done[:, :, :-1])))
l_start = ld[:, :, 1 + 3 * sl]; l_st = ld[:, :, 2 + 3 * sl]; l_gl = ld[:, :, 3 + 3 * sl]
print('loader waves:')
print('  lstore           ', stat((l_st - l_start)[:, :, :-1]))
print('  gload            ', stat((l_gl - l_st)[:, :, :-1]))
print('  barrier wait     ', stat((l_start[:, :, 1:] - l_gl[:, :, :-1])))
# one CU in detail: two co-resident workgroups
ks, cnt = np.unique(key[first], return_counts=True)
kk = ks[cnt >= 2][0]
ids = np.where((key == kk) & first)[0][:2]
print('CU key', kk, 'workgroups', ids.tolist(), 'start', start[ids].tolist())
for s in range(8, 14):
    row = []
    for i in ids:
        row.append('wg%d mf0 [%d rd %d mm %d] ld4 [%d st %d gl %d]' % (
            i, mf[i, 0, 1 + 3 * s], mf[i, 0, 2 + 3 * s] - mf[i, 0, 1 + 3 * s], mf[i, 0, 3 + 3 * s] - mf[i, 0, 2 + 3 * s],
            ld[i, 0, 1 + 3 * s], ld[i, 0, 2 + 3 * s] - ld[i, 0, 1 + 3 * s], ld[i, 0, 3 + 3 * s] - ld[i, 0, 2 + 3 * s]))
    print('  s=%d ' % s + ' | '.join(row))
# launch-level view: when do workgroups start / end, how long do they live
dur = (t[:, 0, 125] - t[:, 0, 0])
print('workgroup lifetime: mean %d  p10 %d  p90 %d cycles' % (dur.mean(), np.percentile(dur, 10), np.percentile(dur, 90)))
print('start time percentiles (0,10,25,50,75,90,100):', [int(np.percentile(start, q)) for q in (0, 10, 25, 50, 75, 90, 100)])
print('end   time percentiles (0,10,25,50,75,90,100):', [int(np.percentile(end, q)) for q in (0, 10, 25, 50, 75, 90, 100)])
pro = t[:, 0, 1] - t[:, 0, 0]; epi = t[:, 0, 125] - t[:, 0, 124]
print('prologue (start -> first step) mean %d p90 %d ; epilogue mean %d p90 %d' % (pro.mean(), np.percentile(pro, 90), epi.mean(), np.percentile(epi, 90)))
