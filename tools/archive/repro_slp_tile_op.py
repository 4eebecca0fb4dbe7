"""Stand-alone reproducer attempt for the round-2 "lost term" in the pool / upsample statistics kernel
(csrc/elementwise.hip: tile_op_stats_kernel, batches of UB = 4 rows, SLP vectoriser ON, beside a GEMM stream).

Build the suspect library first (on the build host):
    DSNT_SLP=1 DSNT_CXXFLAGS=-DDSNT_TILE_UB=4 DSNT_LIB_NAME=libdsnt_slp4.so python dsnt-pose2d_amd/build.py
then, on the GPU box:
    DSNT_HIP_LIB=dsnt-pose2d_amd/csrc/libdsnt_slp4.so python tools/repro_slp_tile_op.py [launches]
It runs dsnt_maxpool2_fwd_stats / dsnt_upsample2_add_fwd_stats on the hourglass's shapes `launches` times on one stream
while a second stream issues 1x1 GEMM launches back to back (the co-residency under which the term was lost), and
compares outputs and statistics partials bit for bit with the first launch.  Prints the number of differing launches."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr, ConvGeom
dev = torch.device('cuda:0')
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lib = _lib.load()
print('library:', _lib.LIB_PATH)
torch.manual_seed(0)
side = torch.cuda.Stream()
# the GEMM stream's work: 1x1 256 -> 128 at 64x64, batch 32 (fp32-MFMA entry point: no bounds needed)
g = ConvGeom(32, 64, 64, 256, 64, 64, 128, 1, 1, 1, 0, 1)
gx = torch.randn(32, 64, 64, 256, device=dev); gw = torch.randn(128, 1, 1, 256, device=dev) * 0.05
gb = torch.zeros(128, device=dev); gy = torch.empty(32, 64, 64, 128, device=dev)
planes = torch.empty(3 * gw.numel(), dtype=torch.bfloat16, device=dev)
assert lib.dsnt_split_bf16x3(ptr(gw), ptr(planes), gw.numel(), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
total_bad = 0
kinds = {}
for (N, H, Cc) in ((32, 64, 256), (32, 32, 256), (32, 16, 256), (32, 8, 256)):
    x = torch.randn(N, H, H, Cc, device=dev)
    low = torch.randn(N, H // 2, H // 2, Cc, device=dev)
    yp = torch.empty(N, H // 2, H // 2, Cc, device=dev); idx = torch.empty(N, H // 2, H // 2, Cc, dtype=torch.uint8, device=dev)
    yu = torch.empty(N, H, H, Cc, device=dev)
    tp, tu = (N * (H // 2) ** 2 + 127) // 128, (N * H * H + 127) // 128
    pp, pu = torch.empty(tp, 2, Cc, device=dev), torch.empty(tu, 2, Cc, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def once():
        assert lib.dsnt_maxpool2_fwd_stats(ptr(x), ptr(yp), ptr(idx), ptr(pp), N, H, H, Cc, None, st) == 0
        assert lib.dsnt_upsample2_add_fwd_stats(ptr(x), ptr(low), ptr(yu), ptr(pu), N, H, H, Cc, None, st) == 0
    once()
    torch.cuda.synchronize()
    ref = (yp.clone(), pp.clone(), yu.clone(), pu.clone())
    bad = 0
    for it in range(launches):
        with torch.cuda.stream(side):
            for _ in range(3):
                lib.dsnt_conv_fwd_bf16x6(ptr(gx), ptr(planes), gw.numel(), ptr(gb), ptr(gy), None, None, 0, None, None, None,
                                         C.byref(g), side.cuda_stream)
        pp.fill_(float('nan')); pu.fill_(float('nan'))
        once()
        torch.cuda.synchronize()
        cur = (yp, pp, yu, pu)
        diff = [n for n, a, b in zip(('pool y', 'pool stats', 'up y', 'up stats'), ref, cur) if not torch.equal(a, b)]
        if diff:
            bad += 1
            for i, nm in ((1, 'pool stats'), (3, 'up stats')):
                if nm not in diff:
                    continue
                pos = (ref[i] != cur[i]).nonzero()
                for q in pos.tolist():
                    kinds[(nm, q[1], q[2] % 4)] = kinds.get((nm, q[1], q[2] % 4), 0) + 1
                if bad <= 3:
                    t_, k_, c_ = pos[0].tolist()
                    out = (yp if i == 1 else yu).view(-1, Cc)[t_ * 128:(t_ + 1) * 128, c_]
                    delta = float(ref[i][t_, k_, c_]) - float(cur[i][t_, k_, c_])
                    vals = out if k_ == 0 else out * out
                    near = float((vals - delta).abs().min())
                    print('  %dx%d launch %d differs in %s; %d entries; first [tile %d, kind %d, channel %d]: ref %r now %r; '
                          'ref - now = %.6f, nearest single term of that tile / channel: off by %.2e' % (
                              H, H, it, diff, len(pos), t_, k_, c_, float(ref[i][t_, k_, c_]), float(cur[i][t_, k_, c_]), delta, near))
    print('%3dx%-3d x %d channels: %d of %d launches differ' % (H, H, Cc, bad, launches))
    total_bad += bad
print('TOTAL differing launches:', total_bad)
print('differing entries by (kernel, kind 0 = sum / 1 = sum of squares, channel % 4):', sorted(kinds.items()))
