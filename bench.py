#!/usr/bin/env python
"""Benchmark of the dsnt-pose2d hot path on MI355X: images/sec of one TRAIN STEP.

A step is the reference's `bin/train.py:355-384` minus data loading, telemetry and PCKh:
forward -> forward_loss -> zero_grad -> backward (-> gradient all-reduce) -> optimiser step,
on synthetic 256x256x3 crops already resident in HBM, 16 joints, 64x64 heat-maps.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hg2_js|hg1|hg8_js|resnet34] [--batch B]
                  [--global-batch G]

For N > 1 either `python bench.py --gpus N` (it starts its own N ranks as a child `python -m torch.distributed.run`, forwards
rank 0's line and exit code) or `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` directly (one rank
per GPU, RCCL).  Rank 0 prints ONE JSON line.  Default scaling is weak: the per-GPU batch is
fixed (32; hg8: 16) as N grows.  `--global-batch G` fixes the TOTAL batch instead (per-GPU batch G / N,
"scaling": "strong") — north_star quotes efficiency at global batch 256: `--global-batch 256` at N = 1 and 8.
Besides the whole-job throughput the line carries
  roofline     — the dominant kernel (3x3 128->128 halo convolution at 64x64, the shape that holds ~half of the
                 backbone FLOPs) timed live with HIP events on its stream: algorithmic FLOPs / launch time vs the
                 split-precision MFMA peak; `roofline.by_time`: the step's largest kernel families (weight gradients,
                 1x1 GEMM, 3x3 halo, BatchNorm-backward apply), each timed the same way with its algorithmic FLOPs AND
                 bytes against whichever bound applies;
  step_bounds  — the whole step against its two floors: GFLOP/step / the split-precision MFMA peak and the
                 launch lists' algorithmic bytes / the 6.29 TB/s copy peak;
  cpu_baseline — the CPU oracle (plain PyTorch ops, proven equal to the reference) timed on this
                 node's host cores on a bounded sample of the same workload (rank 0, N = 1): thread-count
                 sweep, then 2 warm-up + 5 timed steps at the best count;
  parity       — max |dcoord| and PCKh@0.5 of the HIP path vs the CPU oracle on a small fixed batch with
                 the weights the timed steps ended on (outside the timed region);
  head_roofline — the DSNT head kernels alone on a B = 1024 batch: achieved HBM GB/s.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, 'dsnt-pose2d_amd')]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: (base, reg, per-GPU batch, train GFLOP per image [BASELINE.md])
    'hg2_js': ('hg2', 'js', 32, 53.6),
    'hg1': ('hg1', 'none', 32, 34.5),
    'hg8_js': ('hg8', 'js', 16, 168.3),
    # BASELINE config 1 (the reference's CPU-runnable case) on the HIP path: ResNet-34 + DSNT, 8x8 heat-maps
    'resnet34': ('resnet34', 'none', 8, 28.4),
    # SURVEY 8(f) f-2, NOT the headline metric: eval-mode forward (inference.py:33-48), plain and with flip augmentation
    'hg2_infer': ('hg2', 'none', 32, 17.98),
}
PEAK_F32_MFMA = 157.3  # TFLOP/s, MI355X_MICROARCH.md


PEAK_BF16_MFMA = 2500.0  # TFLOP/s dense bf16, MI355X_MICROARCH.md


def committed_baseline(workload, batch_1gpu):
    """images/sec of the committed 1-GPU line for `workload` at per-GPU batch `batch_1gpu` (weak scaling: the run's own
    per-GPU batch; strong scaling: the global batch on one GPU) and where it was read from — so that the driver's N > 1
    runs, which pass no --baseline-ips, still carry `dp.efficiency`.  Latest round first; (None, None) if no line fits.
    A number measured on ANOTHER box (boxes differ by +-2 %): the driver computes its own efficiency from its own N = 1."""
    import glob
    import re
    cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_bench_line*.json')), reverse=True)
    for path in cands:
        try:
            for ln in open(path):
                ln = ln.strip()
                if not ln.startswith('{'):
                    continue
                d = json.loads(ln)
                cfg = d.get('config', {})
                wl = cfg.get('workload', '')
                base, reg, _, _ = WORKLOADS[workload]
                m = re.search(r'batch (\d+) per GPU', wl)
                if (d.get('n_gpus') == 1 and wl.startswith(base + ' + DSNT') and ('JS reg' in wl) == (reg == 'js')
                        and m and int(m.group(1)) == batch_1gpu and d.get('value')):
                    return float(d['value']), os.path.relpath(path, ROOT)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def dominant_kernel_roofline(batch, iters=20):
    """3x3 128->128 conv with fused BN+ReLU prologue and stats epilogue at [B,64,64,128] — the
    shape that holds most of the backbone FLOPs — on the kernel the step actually launches for it:
    the persistent fp16x3 kernel of csrc/conv3s.hip (`dsnt_conv_fwd_f16x3_stream`: two fp16 planes per
    operand after a power-of-two scale, 3 MFMAs per fp32-grade product).  `achieved` is in ALGORITHMIC
    (fp32) TFLOP/s; `peak` is the dense fp16 MFMA peak / 3.  The bf16x6 (peak / 6), tiled fp16x3 and
    fp32-MFMA kernels of the same contract are timed beside it (`twins`)."""
    from dsnt import _lib
    from dsnt._lib import ptr, ConvGeom
    dev = torch.device('cuda', torch.cuda.current_device())
    g = ConvGeom(batch, 64, 64, 128, 64, 64, 128, 3, 3, 1, 1, 1)
    x = torch.randn(batch, 64, 64, 128, device=dev)
    w = torch.randn(128, 3, 3, 128, device=dev) * 0.03
    b = torch.zeros(128, device=dev)
    sc = torch.rand(128, device=dev) + 0.5
    sh = torch.randn(128, device=dev) * 0.1
    y = torch.empty(batch, 64, 64, 128, device=dev)
    M = batch * 64 * 64
    stats = torch.empty((M + 127) // 128, 2, 128, device=dev)
    planes = torch.empty(3 * w.numel(), dtype=torch.bfloat16, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    assert _lib.fn('dsnt_split_bf16x3')(ptr(w), ptr(planes), w.numel(), stream) == 0

    def timed(fn, args):
        # ISOLATED launches, a millisecond of idle between samples: 20 launches back to back measure the clock droop of the
        # loop (DVFS: MI355X_MICROARCH.md) as kernel time — 119.9 us against 107.8 us under rocprofv3 for the same launch in
        # round 3.  HIP events on the stream the kernel runs on; median.
        for _ in range(3):
            assert fn(*args, stream) == 0
        torch.cuda.synchronize()
        samples = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(*args, stream)
            e1.record()
            torch.cuda.synchronize()
            samples.append(e0.elapsed_time(e1))
            time.sleep(1e-3)
        samples.sort()
        return samples[len(samples) // 2]

    def timed_loop(fn, args):                         # the back-to-back loop of rounds 1-3, reported beside it
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn(*args, stream)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    common = (ptr(sc), ptr(sh), 1, None, None, ptr(stats), C.byref(g))
    ms6 = timed(_lib.fn('dsnt_conv_fwd_bf16x6'), (ptr(x), ptr(planes), w.numel(), ptr(b), ptr(y)) + common)
    ms32 = timed(_lib.fn('dsnt_conv_fwd'), (ptr(x), ptr(w), ptr(b), ptr(y)) + common)
    flops = 2.0 * M * (3 * 3 * 128) * 128
    traffic, pmc = None, {}
    tj = os.path.join(ROOT, 'profiles', 'traffic.json')
    if batch == 32 and os.path.exists(tj):      # PMC passes of this exact launch (profiles/)
        pmc = json.load(open(tj))
        traffic = pmc.get('traffic_bytes_per_launch')
    f16 = os.environ.get('DSNT_SPLIT', 'f16x3') == 'f16x3' and os.environ.get('DSNT_MFMA', 'bf16x6') != 'f32'
    twins = {'fp32_mfma_kernel': {'achieved': round(flops / (ms32 * 1e-3) / 1e12, 2), 'peak': PEAK_F32_MFMA,
                                  'us_per_launch': round(ms32 * 1e3, 1)}}
    if f16:
        # the kernel the train step runs for this shape: fp16x3 (two fp16 planes, 3 MFMAs per product); the operand
        # bounds are what the step's producers leave in device memory (weights: amax; activations: here the amax x 4)
        planes16 = torch.empty(2 * w.numel(), dtype=torch.float16, device=dev)
        wb, ab = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
        assert _lib.fn('dsnt_amax')(ptr(w), w.numel(), ptr(wb), stream) == 0
        assert _lib.fn('dsnt_split_f16x2')(ptr(w), ptr(planes16), w.numel(), w.numel(), ptr(wb), stream) == 0
        ab.fill_(float(torch.relu(x * sc + sh).max()) * 4.0)
        ms16h = timed(_lib.fn('dsnt_conv_fwd_f16x3_ex'),
                      (ptr(x), ptr(planes16), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y)) + common + (None, None))
        # ... with the weight planes in stream order: the persistent symmetric kernel (csrc/conv3s.hip) the step launches
        stream16 = torch.empty(2 * w.numel(), dtype=torch.float16, device=dev)
        tab = torch.tensor([[w.data_ptr(), stream16.data_ptr(), wb.data_ptr(), w.numel(), w.numel(), 128, 128]],
                           dtype=torch.int64).to(dev)
        assert _lib.fn('dsnt_f16_prep_weights')(ptr(tab), 1, 7, stream) == 0
        streamed = 'conv3s' not in os.environ.get('DSNT_OFF', '').replace('+', ',').split(',') and bool(_lib.fn('dsnt_conv_fwd_stream_ok')(C.byref(g)))
        dom = (_lib.fn('dsnt_conv_fwd_f16x3_stream'),
               (ptr(x), ptr(stream16), w.numel(), ptr(wb), ptr(ab), ptr(b), ptr(y)) + common + (None, None))
        ms16 = timed(*dom) if streamed else ms16h
        ms_loop = timed_loop(*dom) if streamed else None
        achieved, peak, ms = flops / (ms16 * 1e-3) / 1e12, PEAK_BF16_MFMA / 3.0, ms16
        twins['bf16x6_kernel'] = {'achieved': round(flops / (ms6 * 1e-3) / 1e12, 2), 'peak': round(PEAK_BF16_MFMA / 6.0, 1),
                                  'us_per_launch': round(ms6 * 1e3, 1)}
        twins['halo_tile_kernel_f16x3'] = {'achieved': round(flops / (ms16h * 1e-3) / 1e12, 2), 'peak': round(PEAK_BF16_MFMA / 3.0, 1),
                                           'us_per_launch': round(ms16h * 1e3, 1)}
        kernel = ('conv3s_kernel<128,true,0,32,MF> (fp16x3, 16x16x32 MFMAs, weight ring by LDS-DMA, persistent) 3x3 128->128 @64x64 B=%d' if streamed else
                  'conv3x3_bf16x6_kernel<2,true,F16> (fp16x3) 3x3 128->128 @64x64 B=%d') % batch
        note = ('algorithmic fp32 FLOPs; peak = 2500 TFLOP/s dense fp16 MFMA / 3 MFMAs per product '
                '(= %.0f fp16 TFLOP/s executed)' % (3 * achieved))
    else:
        achieved, peak, ms = flops / (ms6 * 1e-3) / 1e12, PEAK_BF16_MFMA / 6.0, ms6
        kernel = 'conv3x3_bf16x6_kernel<2,true> 3x3 128->128 @64x64 B=%d' % batch
        note = ('algorithmic fp32 FLOPs; peak = 2500 TFLOP/s dense bf16 MFMA / 6 MFMAs per product '
                '(= %.0f bf16 TFLOP/s executed)' % (6 * achieved))
    out = {
        'bound': 'mfma', 'kernel': kernel,
        'achieved': round(achieved, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
        'frac': round(achieved / peak, 4), 'traffic': traffic, 'peak_note': note,
        'flops_per_launch': flops, 'us_per_launch': round(ms * 1e3, 1),
        # achieved / us_per_launch are measured in THIS run (HIP events); traffic and the matrix-pipe figure are
        # NOT: they are copied from the committed rocprofv3 PMC passes of this launch
        'traffic_source': ('profiles/traffic.json <- ' + str(pmc.get('source'))) if traffic is not None else None,
    }
    if f16 and pmc.get('matrix_pipe_busy_frac') is not None:      # PMC passes of this launch (profiles/): SQ_VALU_MFMA_BUSY_CYCLES
        out['matrix_pipe_busy_frac_pmc'] = pmc['matrix_pipe_busy_frac']
    out['timing'] = 'median of %d isolated launches (HIP events on the launch stream, 1 ms idle between samples)' % iters
    if f16 and ms_loop is not None:
        out['us_per_launch_back_to_back'] = round(ms_loop * 1e3, 1)
    if pmc.get('us_per_launch_rocprof') is not None:               # the committed kernel trace's average for the same launch
        out['us_per_launch_rocprof'] = pmc['us_per_launch_rocprof']
        out['us_per_launch_rocprof_source'] = pmc.get('us_per_launch_rocprof_source')
    out.update(twins)
    return out


HBM_PEAK, HBM_COPY_PEAK = 8000.0, 6290.0      # GB/s: spec and measured float4 copy (MI355X_MICROARCH.md)
# conv-in + conv-out activation elements per image (probed on the reference): the algorithmic HBM volume of
# a train step is 4 B x 3 x this
SURVEY_CONV_ELEMS = {'hg1': 36.95e6, 'hg2': 55.93e6, 'hg8': 169.87e6}


def measured_step_traffic(workload, batch):
    """HBM bytes per step as the TCC counters saw them: copied from the committed rocprofv3 --pmc passes of the bare train loop
    (tools/collect_profiles.sh -> profiles/r05_<workload>_b<batch>_step_traffic.txt; the latest round that has the file wins),
    never measured by this run; None for workloads / batches without a committed file."""
    import glob
    import re
    tag = {'hg2_js': 'hg2', 'hg8_js': 'hg8', 'hg1': 'hg1', 'resnet34': 'resnet34'}.get(workload)
    if tag is None:
        return None
    cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_%s_b%d_step_traffic.txt' % (tag, batch))), reverse=True)
    legacy = {('hg2_js', 32): 'r04_step_traffic.txt', ('hg8_js', 16): 'r04_hg8_b16_step_traffic.txt'}.get((workload, batch))
    if legacy:
        cands.append(os.path.join(ROOT, 'profiles', legacy))
    for path in cands:
        if os.path.exists(path):
            mt = re.search(r'=\s*([0-9.]+) GB per step', open(path).read())
            if mt:
                name = os.path.basename(path)
                return {'gbytes': float(mt.group(1)),
                        'source': 'profiles/%s (FETCH_SIZE x 2 + WRITE_SIZE over every kernel of a step)' % name}
    return None


def family_rooflines(batch, iters=20):
    """The step's largest kernel families at their 64x64 shapes, each timed alone with HIP events on its stream:
    algorithmic FLOPs and bytes per launch, and the fraction of the bound that applies (MFMA: 2500 TFLOP/s / 3 fp16
    MFMAs per product; HBM: 8.0 TB/s spec, the 6.29 TB/s copy peak beside it).  `share_of_step` comes from the
    serialised kernel profile committed under profiles/ (what each family costs when it has the chip to itself)."""
    from dsnt import _lib
    from dsnt._lib import ptr, ConvGeom
    dev = torch.device('cuda', torch.cuda.current_device())
    st = torch.cuda.current_stream().cuda_stream
    H, M = 64, batch * 64 * 64

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3          # us

    def entry(kernel, us, flops, nbytes, bound, note=None):
        tf, gbs = flops / us / 1e6, nbytes / us / 1e3
        e = {'kernel': kernel, 'us_per_launch': round(us, 1), 'flops_per_launch': flops, 'bytes_per_launch': nbytes,
             'tflops': round(tf, 1), 'gbs': round(gbs, 1), 'bound': bound}
        if bound == 'mfma':
            e.update(peak=round(PEAK_BF16_MFMA / 3.0, 1), unit='TFLOP/s', frac=round(tf / (PEAK_BF16_MFMA / 3.0), 4))
        else:
            e.update(peak=HBM_PEAK, unit='GB/s', frac=round(gbs / HBM_PEAK, 4), frac_of_copy_peak=round(gbs / HBM_COPY_PEAK, 4))
        if note:
            e['note'] = note
        return e

    def bounds(*ts):
        out = []
        for t in ts:
            b = torch.zeros(64, device=dev)
            assert _lib.fn('dsnt_amax')(ptr(t), t.numel(), ptr(b), st) == 0
            out.append(b)
        return out

    res = []
    sc128, sh128 = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.1
    x128 = torch.randn(batch, H, H, 128, device=dev)
    g128 = torch.randn(batch, H, H, 128, device=dev) * 1e-3
    g256 = torch.randn(batch, H, H, 256, device=dev) * 1e-3
    ab = torch.full((64,), float(torch.relu(x128 * sc128 + sh128).max()) * 4.0, device=dev)
    gb128, gb256 = bounds(g128, g256)
    # -- weight gradients (hourglass.py:22-25 backward): 3x3 128->128 (halo kernel) and 1x1 128->256
    wg = _lib.fn('dsnt_conv_wgrad_f16x3')
    for k, cout, gy, gb in ((3, 128, g128, gb128), (1, 256, g256, gb256)):
        g = ConvGeom(batch, H, H, 128, H, H, cout, k, k, 1, k // 2, 1)
        ws = torch.empty(_lib.fn('dsnt_conv_wgrad_f16x3_ws_floats')(C.byref(g), 0), device=dev)
        us = timed(lambda: wg(ptr(x128), ptr(sc128), ptr(sh128), 1, ptr(gy), ptr(ws), None, None, 0, ptr(ab), ptr(gb),
                              C.byref(g), st))
        flops = 2.0 * M * k * k * 128 * cout
        nbytes = 4.0 * (M * 128 + M * cout + k * k * 128 * cout)
        res.append(entry('weight gradient %dx%d 128->%d @64x64 B=%d (fp16x3%s), slabs only' %
                         (k, k, cout, batch, ', halo kernel' if k == 3 else ''), us, flops, nbytes,
                         'mfma' if k == 3 else 'hbm',
                         'plus %.1f MB of split-M slabs written per launch (workspace, reduced once per parameter bucket)'
                         % (ws.numel() * 4 / 1e6)))
        del ws
    # -- 1x1 GEMM: the expanding conv3 of a Bottleneck, BN+ReLU prologue, residual add, statistics epilogue
    g = ConvGeom(batch, H, H, 128, H, H, 256, 1, 1, 1, 0, 1)
    w = torch.randn(256, 1, 1, 128, device=dev) * 0.05
    planes16 = torch.empty(2 * w.numel(), dtype=torch.float16, device=dev)
    wb, = bounds(w)
    assert _lib.fn('dsnt_split_f16x2')(ptr(w), ptr(planes16), w.numel(), w.numel(), ptr(wb), st) == 0
    y = torch.empty(batch, H, H, 256, device=dev)
    bias = torch.zeros(256, device=dev)
    if _lib.fn('dsnt_conv1x1_fwd_ok')(C.byref(g)):
        # (what the step launches since round 4: the LDS-staged streaming kernel of csrc/fwd1.hip, one statistics row per workgroup)
        stats = torch.empty(_lib.fn('dsnt_conv1x1_fwd_stats_rows')(C.byref(g), 0), 2, 256, device=dev)
        fwd1 = _lib.fn('dsnt_conv1x1_fwd_f16x3')
        us = timed(lambda: fwd1(ptr(x128), ptr(planes16), w.numel(), ptr(wb), ptr(ab), ptr(bias), ptr(y), ptr(sc128), ptr(sh128),
                                1, ptr(g256), ptr(stats), C.byref(g), None, st))
        kname = '1x1 convolution 128->256 @64x64 B=%d forward (fp16x3, fwd1_kernel: BN+ReLU prologue, residual add, statistics)'
    else:
        stats = torch.empty((M + 127) // 128, 2, 256, device=dev)
        fwd = _lib.fn('dsnt_conv_fwd_f16x3_ex')
        us = timed(lambda: fwd(ptr(x128), ptr(planes16), w.numel(), ptr(wb), ptr(ab), ptr(bias), ptr(y), ptr(sc128), ptr(sh128),
                               1, ptr(g256), None, ptr(stats), C.byref(g), None, None, st))
        kname = '1x1 GEMM 128->256 @64x64 B=%d (fp16x3, BN+ReLU prologue, residual add, statistics)'
    res.append(entry(kname % batch, us, 2.0 * M * 128 * 256, 4.0 * (M * 128 + 2 * M * 256), 'hbm'))
    # -- the whole backward of conv1 of a Bottleneck (256 -> 128) in one launch: data gradient + BatchNorm-backward sums +
    #    weight gradient + the folded BatchNorm backward of bn2 (csrc/bwd1.hip); every tensor once
    from dsnt._lib import BnBwdEpilogue, BnBwdApply
    g = ConvGeom(batch, H, H, 256, H, H, 128, 1, 1, 1, 0, 1)
    if _lib.fn('dsnt_conv1x1_bwd_ok')(C.byref(g)):
        x256 = torch.randn(batch, H, H, 256, device=dev)
        sc256, sh256 = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
        mu256, is256 = torch.randn(256, device=dev) * 0.1, torch.rand(256, device=dev) + 0.5
        wd = torch.randn(256, 128, device=dev) * 0.05
        wbd, = bounds(wd)
        pl = torch.empty(2 * wd.numel(), dtype=torch.float16, device=dev)
        assert _lib.fn('dsnt_split_f16x2')(ptr(wd), ptr(pl), wd.numel(), wd.numel(), ptr(wbd), st) == 0
        ab256 = torch.full((64,), float(torch.relu(x256 * sc256 + sh256).max()) * 4.0, device=dev)
        coef = torch.randn(2, 128, device=dev) * 1e-5
        gbd = torch.full((64,), float(g128.abs().max()) * 8.0, device=dev)
        xs = BnBwdEpilogue(ptr(x256), ptr(sc256), ptr(sh256), ptr(mu256), ptr(is256), 1)
        apn = BnBwdApply(ptr(x128), ptr(sc128), ptr(sh128), ptr(sc128), ptr(coef))
        nsp = _lib.fn('dsnt_conv1x1_bwd_splits')(C.byref(g), 0)
        ws = torch.empty(_lib.fn('dsnt_conv1x1_bwd_ws_floats')(C.byref(g), 0), device=dev)
        part, dzo = torch.empty(nsp, 2, 256, device=dev), torch.empty(batch, H, H, 256, device=dev)
        b1 = _lib.fn('dsnt_conv1x1_bwd_f16x3')
        us = timed(lambda: b1(C.byref(xs), ptr(g128), C.byref(apn), ptr(pl), wd.numel(), ptr(wbd), ptr(ab256), ptr(gbd),
                              ptr(dzo), ptr(part), ptr(ws), None, 0, C.byref(g), st))
        res.append(entry('one-pass backward of the 1x1 convolution 256->128 @64x64 B=%d (fp16x3: dX + BN-backward sums + dW + '
                         'folded BN backward of the layer behind)' % batch, us, 2 * 2.0 * M * 128 * 256,
                         4.0 * (2 * M * 128 + 2 * M * 256), 'hbm',
                         'replaces bn_act_bwd_apply + the 1x1 data gradient + the 1x1 weight gradient (737 MB in three launches); '
                         'plus %.1f MB of slabs (one per workgroup)' % (ws.numel() * 4 / 1e6)))
        del x256, ws, dzo
    # -- BatchNorm-backward apply (hourglass.py:36-43 backward): dx = f(dz, x), 12 B per element
    dz, dx = g128, torch.empty_like(g128)
    mean, invstd, coef = torch.zeros(128, device=dev), torch.ones(128, device=dev), torch.zeros(256, device=dev)
    amax = torch.zeros(64, device=dev)
    ap = _lib.fn('dsnt_bn_act_bwd_apply_amax')
    us = timed(lambda: ap(ptr(dz), ptr(x128), ptr(sc128), ptr(sh128), ptr(mean), ptr(invstd), ptr(coef), 0, ptr(dx), 0,
                          M, 128, ptr(amax), st))
    res.append(entry('bn_act_bwd_apply 128 channels @64x64 B=%d (with the fp16x3 bound of its output)' % batch, us,
                     8.0 * M * 128, 12.0 * M * 128, 'hbm'))
    return res


def _host_cpus():
    """(usable logical CPUs, physical cores among them, model string) of this process's CPU share."""
    usable = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else list(range(os.cpu_count() or 1))
    try:                                     # cgroup v2 quota (a GPU box hands out a share of the node)
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            usable = usable[:max(1, int(float(q) / float(per)))]
    except (OSError, ValueError):
        pass
    model, cores = 'unknown', set()
    try:
        cpu, phys, core = None, 0, None
        for line in open('/proc/cpuinfo'):
            k, _, v = line.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'processor':
                cpu = int(v)
            elif k == 'model name':
                model = v
            elif k == 'physical id':
                phys = int(v)
            elif k == 'core id':
                core = int(v)
            elif not k and cpu is not None:
                if cpu in usable:
                    cores.add((phys, core if core is not None else cpu))
                cpu, core = None, None
    except OSError:
        pass
    return len(usable), (len(cores) or len(usable)), model


def cpu_baseline(base, reg, batch=32, timed=5, warm=2, fallback_batch=8, fallback_above_s=60.0):
    """The CPU oracle on the same workload at a bounded batch (images/sec on the host cores).  BASELINE.md §3:
    physical-core count and CPU model stated, 2 warm-up + >= 5 timed steps, median; on more than 16 logical CPUs the
    thread count is swept first (an oversubscribed intra-op pool is several times slower)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    from dsnt_oracle import model as omodel
    from dsnt import synthetic
    logical, physical, cpu_model = _host_cpus()
    m = omodel.build_mpii_pose_model(base=base, output_strat='dsnt', reg=reg)
    synthetic.fill_state_dict(m, seed=0)
    m.train()
    opt = torch.optim.RMSprop(m.parameters(), lr=2.5e-4)
    asked = batch
    x, t, k = synthetic.batch(batch, size=256, seed=1)

    def step():
        t0 = time.perf_counter()
        out = m(x)
        loss = m.forward_loss(out, t, k)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return time.perf_counter() - t0

    saved = torch.get_num_threads()
    # A GPU box hands this process <= 16 logical CPUs: every one of them is used and there is nothing to sweep (the
    # round-2 sweep compared 8 with 16 threads on first-touch steps, i.e. it measured warm-up).  On a larger share the
    # thread count is swept with TWO warm-up steps per candidate before its timed one.
    sweep, cands = {}, [logical]
    if logical > 16:
        cands = sorted({n for n in (16, 32, 64, physical, logical) if 1 <= n <= logical})
        for n in cands:
            torch.set_num_threads(n)
            step()
            step()
            sweep[n] = step()
    best_n = min(sweep, key=sweep.get) if sweep else logical
    torch.set_num_threads(best_n)
    # the configuration's own batch ("bs 32 if memory/time permit else bs 8"): the first warm-up step decides
    first = step()
    fell_back = first > fallback_above_s and batch > fallback_batch
    if fell_back:
        batch = fallback_batch
        x, t, k = synthetic.batch(batch, size=256, seed=1)
        step()
    for _ in range(warm - 1):
        step()
    times = sorted(step() for _ in range(timed))
    torch.set_num_threads(saved)
    med = times[len(times) // 2]
    out = {'value': round(batch / med, 3), 'unit': 'images/sec', 'cores': best_n, 'kind': 'port',
           'physical_cores': physical, 'logical_cpus': logical, 'cpu_model': cpu_model,
           'spread_images_per_sec': [round(batch / times[-1], 3), round(batch / times[0], 3)],
           'batch': batch,
           'sample': '%s+dsnt reg=%s, batch %d%s, 256x256, RMSprop train step; %s; %d warm-up + %d timed steps at %d '
                     'threads (median); torch %s CPU ops'
                     % (base, reg, batch, (' (the configuration\'s batch %d took %.0f s per step: fell back)' % (asked, first))
                        if fell_back else '', ('thread sweep %s (2 warm-up + 1 timed each)' % cands) if sweep else
                        'all %d logical CPUs of this process (no sweep at <= 16)' % logical, warm, timed, best_n,
                        torch.__version__)}
    if sweep:
        out['thread_sweep_images_per_sec'] = {str(n): round(batch / v, 3) for n, v in sweep.items()}
    return out


def parity_vs_oracle(model, base, reg, batch=4):
    """max |dcoord| and PCKh@0.5 of the HIP path vs the CPU oracle: the weights the timed steps ended on, one fixed
    batch, train-mode forward (the path being timed), outside the timed region (SURVEY.md 8(d) "Metric")."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    from dsnt_oracle import model as omodel
    from dsnt_oracle.evaluator import PCKhEvaluator as OracleEval
    from dsnt import synthetic
    from dsnt.evaluator import PCKhEvaluator
    o = omodel.build_mpii_pose_model(base=base, output_strat='dsnt', reg=reg)
    o.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    o.train()
    size = 256
    x, t, k = synthetic.batch(batch, size=size, seed=5, mask_p=0.9)
    dev = next(model.parameters()).device
    with torch.no_grad():
        got = model.compute_coords(model(x.to(dev)))
        want = o.compute_coords(o(x))
    head, tm, tb = synthetic.pckh_inputs(batch)
    ev = PCKhEvaluator()
    ev.add_normalized(got.to(dev), t, k, head, tm, tb)
    oe = OracleEval()
    pred = torch.baddbmm(tb.double(), want.double(), tm.double())
    targ = torch.baddbmm(tb.double(), t.double(), tm.double())
    oe.add(pred, targ, k, head)
    a, b = ev.meters['all'].value()[0], oe.meters['all'].value()[0]
    return {'max_abs_dcoord': float((got - want).abs().max()), 'bar': 1e-4,
            'pckh_hip': round(a, 6), 'pckh_cpu_oracle': round(b, 6), 'pckh_equal': bool(a == b),
            'sample': 'batch %d, %dx%d, train-mode forward, weights after the timed steps' % (batch, size, size)}


def head_roofline(batch=1024, iters=10):
    """The DSNT head alone (softmax + expectation + Euclidean + JS, forward and backward) on a large batch:
    algorithmic HBM bytes = 4 passes x rows x 64 x 64 x 4 B (read logits, write heat-maps, read heat-maps, write
    d logits: SURVEY.md 8(d)) over the HIP-event time of the launches of one stack's head."""
    import dsnt.nn as dn
    dev = torch.device('cuda', torch.cuda.current_device())
    rows = batch * 16
    logits = torch.randn(batch, 16, 64, 64, device=dev) * 3
    target = torch.rand(batch, 16, 2, device=dev) * 2 - 1
    mask = torch.ones(batch, 16, device=dev)

    def once():
        lg = logits.requires_grad_()
        hm, coords = dn.head_forward(lg)
        loss = dn.head_loss(lg, hm.detach(), coords.detach(), target, mask, 'js', 2.0 / 64, 1.0)
        loss.backward()
        lg.grad = None
    for _ in range(2):
        once()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        once()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = 4.0 * rows * 64 * 64 * 4
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {'bound': 'hbm', 'kernels': 'head_fwd + head_loss_grad (+ loss reduce, gradient rescale) through dsnt.nn autograd', 'batch': batch,
            'achieved': round(gbs, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(gbs / 8000.0, 4),
            'frac_of_measured_copy_peak_6290': round(gbs / 6290.0, 4), 'algorithmic_bytes': nbytes,
            'ms_fwd_loss_bwd': round(ms, 4)}


def infer_bench(model, x, batch, world, rank, args):
    """Eval-mode forward throughput (replicas only across GPUs: no collective), with and without the reference's
    horizontal-flip test-time augmentation (inference.py:33-48: both orientations through the backbone, the flipped
    half mirrored back with left/right joints swapped, mean of the un-normalised heat-maps, then the DSNT head)."""
    from dsnt.inference import HFLIP_INDICES
    model.eval()
    idx = HFLIP_INDICES.to(x.device)

    def plain():
        with torch.no_grad():
            return model(x)[-1]

    def flipped():
        with torch.no_grad():
            hm = model.forward_part1(torch.cat([x, x.flip(-1)], 0))[-1]
            hm2 = hm[batch:].flip(-1).index_select(-3, idx)
            return model.forward_part2([(hm[:batch] + hm2) / 2])[-1]
    res = {}
    for name, fn in (('plain', plain), ('flip_tta', flipped)):
        for _ in range(max(args.warmup, 1)):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        res[name] = (time.perf_counter() - t0) / args.steps
    if rank == 0:
        out = {'metric': 'images/sec (inference, eval-mode forward, 256x256, 16 joints)', 'value': round(world * batch / res['plain'], 1),
               'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(1e3 * res['plain'], 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': 'hg2 + DSNT inference, batch %d per GPU, replicas' % batch, 'global_batch': world * batch,
                          'parallelism': 'replicas%d' % world},
               'flip_tta_images_per_sec': round(world * batch / res['flip_tta'], 1),
               'note': 'not the BASELINE metric (that is the train step: default workload)'}
        print(json.dumps(out), file=_JSON_OUT, flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


# The contract is ONE JSON line on stdout.  Libraries write to file descriptor 1 behind Python's back (RCCL prints a version
# banner when its first communicator comes up), so the process's stdout is pointed at stderr and the line goes to a private
# duplicate of the original descriptor.
_JSON_OUT = sys.stdout


def _claim_stdout():
    global _JSON_OUT
    sys.stdout.flush()
    _JSON_OUT = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)


def launcher_command(n_gpus, argv, port):
    """argv + environment of the child that runs this script as N ranks on one node: `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <the caller's own arguments>` — the command the driver
    uses for its scaling runs.  Pure (no GPU, no sockets): tests/test_host_logic_cpu.py checks it."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'ROLE_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    env['MASTER_ADDR'] = '127.0.0.1'
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '4')
    return cmd, env


def launch_ranks(n_gpus, argv):
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE): start the N ranks as a CHILD process — never exec: this process may not
    be replaced once anything in it could have touched the GPU, and it never does (no dsnt import, no torch.cuda call) — forward
    rank 0's one JSON line to the caller's stdout, everything else to stderr, and return the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd, env = launcher_command(n_gpus, argv, port)
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in child.stdout.splitlines() if ln.strip()]
    json_lines = [ln for ln in lines if ln.lstrip().startswith('{') and '"metric"' in ln]
    for ln in lines:
        if not json_lines or ln is not json_lines[-1]:
            print(ln, file=sys.stderr)
    if json_lines:
        print(json_lines[-1], file=_JSON_OUT, flush=True)
    elif child.returncode == 0:
        print('bench.py: the ranks ended without a JSON line', file=sys.stderr)
        return 1
    return child.returncode


def main():
    _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='hg2_js', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the workload\'s)')
    ap.add_argument('--global-batch', type=int, default=None,
                    help='strong scaling: total batch over all GPUs (per-GPU batch = this / N)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the parity / head / dominant-kernel legs (profiling)')
    ap.add_argument('--force-dp', action='store_true', help='wire the data-parallel hooks even at world size 1 (testing)')
    ap.add_argument('--baseline-ips', type=float, default=None,
                    help='images/sec of the 1-GPU run of the same scaling mode: the line then carries `efficiency` = '
                         'value / (N x this)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as the driver calls it: this process becomes the launcher of N ranks and nothing else
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d was started with WORLD_SIZE=%d: launch it as `python bench.py --gpus N` (it starts its '
                         'own ranks) or with torch.distributed.run --nproc-per-node N' % (args.gpus, world))
    # DSNT_BENCH_REHEARSAL=1 (a one-GPU box): the N > 1 code path with every rank on device 0 over gloo — RCCL refuses two ranks
    # on one device.  It rehearses the control flow of the multi-GPU line (barriers, max over ranks, the `dp` block), not its speed;
    # the line says so in `dp.rehearsal`.
    rehearsal = world > 1 and os.environ.get('DSNT_BENCH_REHEARSAL', '0') == '1'
    if rehearsal:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if local_rank >= torch.cuda.device_count():
        raise SystemExit('bench.py --gpus %d: rank %d needs cuda:%d but this node has %d device(s) (one process per GPU; '
                         'DSNT_BENCH_REHEARSAL=1 rehearses the control flow on one device over gloo)'
                         % (args.gpus, rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    from dsnt.model import build_mpii_pose_model
    from dsnt import synthetic, optim, parallel

    base, reg, batch, gflop = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit('--global-batch %d is not divisible by %d GPUs' % (args.global_batch, world))
        batch = args.global_batch // world
    model = build_mpii_pose_model(base=base, output_strat='dsnt', reg=reg)
    synthetic.fill_state_dict(model, seed=0)          # identical weights on every rank
    model.cuda().train()
    x, target, mask = synthetic.batch(batch, size=256, seed=1 + rank, mask_p=1.0)
    x, target, mask = x.to(dev), target.to(dev), mask.to(dev)

    if args.workload == 'hg2_infer':
        return infer_bench(model, x, batch, world, rank, args)
    runner = (model.hg if hasattr(model, 'hg') else model)._runner()
    runner.ensure(dev)
    from dsnt.guard import NanGuard
    guard = NanGuard(dev)                             # train.py:360-371 as a device-side flag, polled asynchronously
    opt = optim.RMSprop(model, lr=2.5e-4, guard=guard)   # train.py:88-99 defaults for rmsprop
    if world > 1 or args.force_dp:
        if args.force_dp and not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29511')
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        dp = parallel.DataParallel(model, opt)
        dp.reducer.time_waits = True           # HIP events around the wait for the collectives: `exposed_comm_ms`
    else:
        dp = None

    def step():
        out = model(x)
        loss = model.forward_loss(out, target, mask)
        guard.check(loss)
        opt.zero_grad()
        loss.backward()
        opt.step()
        guard.poll()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = float(loss.item())
    guard.sync()
    exposed = dp.reducer.exposed_comm_ms() if dp is not None else None

    out = None
    if rank == 0:
        ips = world * batch * args.steps / elapsed
        prog = [p for p in runner.programs.values() if p.training][0]
        out = {
            'metric': 'images/sec (train step, 256x256, 16 joints)',
            'value': round(ips, 2), 'unit': 'images/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'strong' if args.global_batch else 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'dtype_note': 'fp32 tensors and accumulation; large convolutions run as split-precision '
                                          'products on the 16-bit matrix cores with fp32-GEMM-grade error: two fp16 planes '
                                          'after a power-of-two scale + 3 MFMAs (fp16x3) where an operand bound exists, '
                                          'three bf16 planes + 6 MFMAs (bf16x6) elsewhere; small ones on fp32 MFMA',
            'data': 'synthetic',
            'config': {'workload': '%s + DSNT%s, 256x256 -> %s, batch %d per GPU, RMSprop lr 2.5e-4, '
                                   'train step fwd+loss+bwd%s+optim'
                                   % (base, '' if reg == 'none' else ' + %s reg' % reg.upper(),
                                      '8x8x16' if base.startswith('resnet') else '64x64x16', batch,
                                      '+RCCL all-reduce' if world > 1 else ''),
                       'global_batch': world * batch, 'parallelism': 'dp%d' % world,
                       'launches_fwd': prog.n_fwd, 'launches_bwd': prog.n_bwd},
            'final_loss': final_loss,
        }
        # the step against its two floors (per GPU): arithmetic at the split-precision MFMA peak the convolutions run
        # on (2500 TFLOP/s dense fp16 / 3 MFMAs per fp32 product), and the launch lists' algorithmic bytes (every
        # tensor a launch names is read or written once; weight-gradient slabs excluded; + 16 B/parameter for the
        # optimiser) at the measured 6.29 TB/s copy peak
        tape = prog.tape
        nparam = sum(p.numel() for p in model.parameters())
        step_bytes = tape.bytes_fwd + tape.bytes_bwd + 16 * nparam
        mfma_ms = batch * gflop * 1e9 / (PEAK_BF16_MFMA / 3.0 * 1e12) * 1e3
        hbm_ms = step_bytes / (HBM_COPY_PEAK * 1e9) * 1e3
        ms_step = 1e3 * elapsed / args.steps
        out['step_bounds'] = {'gflop_per_step': round(batch * gflop, 1), 'mfma_peak_tflops': round(PEAK_BF16_MFMA / 3.0, 1),
                              'mfma_ms': round(mfma_ms, 3),
                              # bytes the launch lists name (every pass the CURRENT design makes counts as algorithmic) ...
                              'algorithmic_gbytes_per_step': round(step_bytes / 1e9, 2),
                              'hbm_peak_gbs': HBM_COPY_PEAK, 'hbm_ms': round(hbm_ms, 3),
                              'frac_of_max_bound': round(max(mfma_ms, hbm_ms) / ms_step, 4),
                              'frac_of_sum_of_bounds': round((mfma_ms + hbm_ms) / ms_step, 4)}
        # ... and SURVEY.md 8(d)'s floor, which no design choice moves: 4 B x 3 (forward, data gradient, weight gradient) x
        # (conv-in + conv-out elements) with BN + ReLU folded into the operand loads and the adds into the epilogues
        elems = SURVEY_CONV_ELEMS.get(base)
        if elems:
            sv = 4.0 * 3.0 * elems * batch
            sv_ms = sv / (HBM_COPY_PEAK * 1e9) * 1e3
            out['step_bounds'].update(survey_algorithmic_gbytes=round(sv / 1e9, 2), survey_hbm_ms=round(sv_ms, 3),
                                      frac_of_survey_hbm_floor=round(sv_ms / ms_step, 4),
                                      frac_of_max_survey_bound=round(max(mfma_ms, sv_ms) / ms_step, 4))
        meas = measured_step_traffic(args.workload, batch)
        if meas:
            out['step_bounds'].update(measured_gbytes=meas['gbytes'], measured_gbytes_source=meas['source'],
                                      measured_hbm_ms=round(meas['gbytes'] / HBM_COPY_PEAK * 1e3, 3),
                                      sustained_tbs_on_measured_bytes=round(meas['gbytes'] / ms_step, 3))
        if dp is not None:
            # SURVEY.md 8(d) "DP all-reduce": the communication the step could not hide (device time the publishing stream
            # waited for the bucket all-reduces after the backward list was enqueued, mean per step, rank 0), RCCL's own
            # world size, and — against the 1-GPU throughput of the same mode — the scaling efficiency
            out['dp'] = {'rccl_world': dist.get_world_size() if dist.is_initialized() else 1,
                         'backend': dist.get_backend() if dist.is_initialized() else None,
                         'exposed_comm_ms': None if exposed is None else round(exposed, 4),
                         'gradient_mbytes': round(4e-6 * runner.arena.numel, 1), 'buckets': len(runner.arena.bucket_bounds),
                         # bytes of each bucket's all-reduce, in the order backward completes them (last stack first, stem last)
                         'bucket_allreduce_bytes': list(reversed(dp.reducer.bucket_bytes())),
                         'guard_flag_exchanged': True}
            if rehearsal:
                out['dp']['rehearsal'] = 'all %d ranks on one device over gloo: control flow only, not a scaling measurement' % world
            base_ips, base_src = args.baseline_ips, '--baseline-ips'
            if not base_ips and world > 1:
                base_ips, base_src = committed_baseline(args.workload, world * batch if args.global_batch else batch)
            if base_ips:
                out['dp']['baseline_ips'] = base_ips
                out['dp']['baseline_source'] = base_src
                # weak and strong alike: throughput per GPU kept.  Against a COMMITTED line (another box, +-3 %) the figure is named
                # for what it is; the driver computes the real efficiency from its own N = 1 run
                key = 'efficiency' if base_src == '--baseline-ips' else 'efficiency_vs_committed'
                out['dp'][key] = round(ips / (base_ips * world), 4)
        if not args.no_extras:
            out['roofline'] = dominant_kernel_roofline(batch)
            out['roofline']['by_time'] = family_rooflines(batch)
            out['head_roofline'] = head_roofline()
            if hasattr(model, 'hg'):
                out['parity'] = parity_vs_oracle(model, base, reg)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(base, reg)
            out['speedup_vs_cpu'] = round(out['value'] / out['cpu_baseline']['value'], 1)
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        # after the process group is gone: RCCL prints its version banner late, and the JSON line must be
        # the last line on stdout
        sys.stdout.flush()
        C.CDLL(None).fflush(None)       # RCCL's banner sits in the C stdio buffer until exit otherwise
        print(json.dumps(out), file=_JSON_OUT, flush=True)


if __name__ == '__main__':
    main()
