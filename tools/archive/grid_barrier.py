"""Chip-wide grid barrier vs kernel boundary on MI355X: us per barrier for 16..256 co-resident
workgroups, against us per dependent launch of an empty / tiny kernel in one stream.   python tools/grid_barrier.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]
from dsnt import _lib
from dsnt._lib import ptr
dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
out = torch.zeros(4, device=dev)
bar = _lib.fn('dsnt_debug_grid_barrier')
emp = _lib.fn('dsnt_debug_empty')
def timed(f, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best
iters = 2000
for threads in (256, 512):
    for blocks in (16, 64, 128, 256):
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        def go():
            cnt.zero_()
            assert bar(ptr(cnt), blocks, threads, iters, ptr(out), st) == 0
        go()
        torch.cuda.synchronize()
        assert int(cnt.item()) == iters * blocks, (int(cnt.item()), iters * blocks)
        t = timed(go)
        print('grid barrier: %3d workgroups x %4d threads: %.2f us per barrier (polls by workgroup 0: %.0f per barrier)' % (
            blocks, threads, 1e6 * t / iters, float(out[0]) / iters))
bar2 = _lib.fn('dsnt_debug_grid_barrier2')
for blocks in (64, 128, 256):
    cnt = torch.zeros(16 * 9, dtype=torch.int32, device=dev)
    def go():
        cnt.zero_()
        assert bar2(ptr(cnt), blocks, 256, iters, ptr(out), st) == 0
    go()
    torch.cuda.synchronize()
    assert int(cnt[0].item()) == iters * min(blocks, 8), (int(cnt[0].item()), iters)
    t = timed(go)
    print('two-level grid barrier (one counter per XCD): %3d workgroups: %.2f us per barrier' % (blocks, 1e6 * t / iters))
for blocks, threads in ((1, 64), (8, 1024), (64, 256), (256, 256)):
    n = 2000
    def go():
        for _ in range(n):
            emp(blocks, threads, ptr(out), st)
    go()
    t = timed(go, 3)
    print('dependent launches: %3d x %4d threads: %.2f us per launch (host-bound if the host cannot keep up: ~3.6 us per call)' % (
        blocks, threads, 1e6 * t / n))
