"""Build libdsnt_hip.so in-tree: hipcc --offload-arch=gfx950 over csrc/*.hip + api.cpp.

Cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
repo snapshot.  Usage: python dsnt-pose2d_amd/build.py [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libdsnt_hip.so')
SOURCES = ['api.cpp', 'conv.hip', 'wgrad3.hip', 'wgrad1.hip', 'gemm1.hip', 'bwd1.hip', 'fwd1.hip', 'stem4.hip', 'conv3s.hip', 'dgrad_up.hip', 'elementwise.hip', 'head.hip', 'heatmap.hip', 'debug.hip']
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wno-unused-value',
         '-Wno-unused-result']
if os.environ.get('DSNT_TIMELINE'):      # wave timeline stamps in the conv kernels (diagnostic scripts of rounds 1-3, removed in round 6: git history)
    FLAGS.append('-DDSNT_TIMELINE')
if os.environ.get('DSNT_TIMELINE') == '2':  # loader stamps split into wait / stage (same)
    FLAGS.append('-DDSNT_TIMELINE2')
# kernel experiments: extra compiler flags and another output name (load it with DSNT_HIP_LIB=<path>), e.g.
#   DSNT_CXXFLAGS=-DDSNT_WG6U_SCHED DSNT_LIB_NAME=libdsnt_exp.so python dsnt-pose2d_amd/build.py --force
FLAGS += os.environ.get('DSNT_CXXFLAGS', '').split()
if os.environ.get('DSNT_LIB_NAME'):
    LIB = os.path.join(CSRC, os.environ['DSNT_LIB_NAME'])


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    deps = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'bn_pro.h'), os.path.join(CSRC, 'conv_split.h'),
            os.path.join(CSRC, 'wgrad3.h'), os.path.join(CSRC, 'gemm1.h'), os.path.join(CSRC, 'conv3s.h'), os.path.join(CSRC, 'bwd1.h'), os.path.join(CSRC, 'fwd1.h'), os.path.join(CSRC, 'stem4.h'), os.path.join(CSRC, 'stage.h'), os.path.join(CSRC, 'ew_bodies.h'),
            os.path.join(HERE, '..', 'include', 'dsnt_hip.h'), os.path.join(HERE, '..', 'include', 'dsnt_hip_debug.h')]
    objs, jobs = [], []
    bdir = 'build' if not os.environ.get('DSNT_LIB_NAME') else 'build_' + os.path.splitext(os.environ['DSNT_LIB_NAME'])[0]
    os.makedirs(os.path.join(CSRC, bdir), exist_ok=True)
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, bdir, os.path.splitext(src)[0] + '.o')
        objs.append(o)
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            # No SLP vectorisation anywhere.  conv.hip: v_pk_*_f32 beside the MFMAs cost more issue time than they save.
            # Everywhere else: a pool kernel whose row sums the vectoriser had packed (v_pk_add_f32 with op_sel on a
            # v_mov_b64'd register pair) lost a term in about one launch in ten — only while other kernels shared its
            # CUs; the same source built without the vectoriser was bit-reproducible in 80 of 80 passes
            # (tools/determinism_fwd.py, DESIGN.md "round 2").
            extra = ['-fno-slp-vectorize'] if (src == 'conv.hip' or not os.environ.get('DSNT_SLP')) else []   # DSNT_SLP=1: A/B only
            if src == 'heatmap.hip':      # the reference's separately rounded fp32 coordinate steps: no FMA contraction
                extra = extra + ['-ffp-contract=off']
            cmd = [hipcc] + FLAGS + extra + (['-x', 'hip'] if src.endswith('.cpp') else []) + ['-c', s, '-o', o]
            jobs.append(cmd)
    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([hipcc, '-shared', '--offload-arch=gfx950', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
