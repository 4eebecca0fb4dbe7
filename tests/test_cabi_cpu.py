"""CPU checks of the C-ABI boundary: the library loads, exports every symbol include/dsnt_hip.h
declares, and its argument validation returns error codes without touching a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, 'include', 'dsnt_hip.h'),          # the product ABI
           os.path.join(ROOT, 'include', 'dsnt_hip_debug.h')]    # calibration / timeline diagnostics (tools/ only)


def _declared(headers=HEADERS):
    names = set()
    for h in headers:
        text = open(h).read()
        text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
        names |= set(re.findall(r'\b(dsnt_[a-z0-9_]+)\s*\(', text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from dsnt import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), 'libdsnt_hip.so does not export ' + n
    bound = set(_lib.SIGNATURES) | set(_lib.PLAIN)
    assert bound == set(names), (sorted(bound - set(names)), sorted(set(names) - bound))
    assert lib.dsnt_version() >= 100
    # the product header declares no diagnostics, and nothing in the product package binds one at import time
    assert not [n for n in _declared(HEADERS[:1]) if n.startswith('dsnt_debug')]


def test_argument_validation_without_gpu():
    from dsnt import _lib
    lib = _lib.load()
    g = _lib.ConvGeom(1, 8, 8, 6, 8, 8, 8, 1, 1, 1, 0, 1)       # Cin % 4 != 0
    rc = lib.dsnt_conv_fwd(None, None, None, None, None, None, 0, None, None, None, C.byref(g), None)
    assert rc == 2 and b'multiple of 4' in lib.dsnt_last_error()
    g = _lib.ConvGeom(1, 8, 8, 8, 7, 8, 8, 3, 3, 1, 1, 1)       # inconsistent output size
    rc = lib.dsnt_conv_fwd(None, None, None, None, None, None, 0, None, None, None, C.byref(g), None)
    assert rc == 1 and b'inconsistent' in lib.dsnt_last_error()
    g = _lib.ConvGeom(1, 8, 8, 8, 8, 8, 8, 3, 3, 1, 1, 1)
    rc = lib.dsnt_conv_fwd(None, None, None, None, None, None, 0, None, None, None, C.byref(g), None)
    assert rc == 3 and b'null' in lib.dsnt_last_error()
    assert lib.dsnt_head_fwd(None, None, None, 4, 8, 8, None) == 3
    assert lib.dsnt_reg_fwd(None, None, None, 4, 8, 8, 0.1, 9, None) == 3
    assert lib.dsnt_maxpool2_fwd(C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), 1, 7, 8, 4, None) == 1
    assert lib.dsnt_conv_wgrad_ws_floats(C.byref(g)) > 0
    assert lib.dsnt_conv_fwd_bm(C.byref(g)) in (32, 128)


def test_ops_refuse_cpu_tensors():
    import torch
    import dsnt.nn as dn
    from dsnt.model import build_mpii_pose_model
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        dn.dsnt(torch.zeros(1, 1, 4, 4))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        dn.euclidean_loss(torch.zeros(2, 3, 2), torch.zeros(2, 3, 2))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        dn.js_reg_loss(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 2), 0.1)
    m = build_mpii_pose_model(base='hg1', output_strat='dsnt')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.zeros(1, 3, 64, 64))


def test_missing_library_is_loud(monkeypatch):
    from dsnt import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libdsnt_hip.so')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load()


def test_launch_list_records_instead_of_launching():
    """dsnt_list_*: between begin and end the entry points validate as usual and launch nothing; errors do not record."""
    from dsnt import _lib
    lib = _lib.load()
    h = lib.dsnt_list_create()
    assert h and lib.dsnt_list_size(h) == 0 and lib.dsnt_list_segments(h) == 1
    assert lib.dsnt_list_end() != 0 and b'not recording' in lib.dsnt_last_error()
    assert lib.dsnt_list_begin(h) == 0
    assert lib.dsnt_list_begin(h) != 0                         # one recording per thread
    g = _lib.ConvGeom(1, 8, 8, 6, 8, 8, 8, 1, 1, 1, 0, 1)     # invalid geometry: rejected, nothing recorded
    assert lib.dsnt_conv_fwd(None, None, None, None, None, None, 0, None, None, None, C.byref(g), None) == 2
    assert lib.dsnt_list_size(h) == 0
    # a valid call records one launch without touching a device (pointers are only captured); stream = lane 1
    assert lib.dsnt_axpy(C.c_void_p(4096), C.c_void_p(8192), 1.0, 0, 1024, C.c_void_p(1)) == 0
    assert lib.dsnt_list_size(h) == 1
    assert lib.dsnt_list_mark(h) == 1 and lib.dsnt_list_segments(h) == 2
    assert lib.dsnt_list_end() == 0
    assert lib.dsnt_list_replay(h, 5, None, 3) != 0            # bad arguments are refused before anything is enqueued
    lib.dsnt_list_destroy(h)


def test_no_packed_fp32_instruction_with_crossed_halves_on_its_own_destination():
    """The one instruction form that loses results on gfx950 (profiles/r03_slp_packed_add_hazard.txt: a v_pk_*_f32 whose
    destination pair is also a source pair read with crossed halves — what hipcc's SLP vectoriser makes of adjacent scalar
    sums) must not appear in the shipped library: every translation unit is built with -fno-slp-vectorize (build.py),
    and this disassembles the built code objects to hold it to that."""
    import importlib.util
    import shutil
    if not shutil.which('/opt/rocm/lib/llvm/bin/llvm-objdump'):
        pytest.skip('llvm-objdump not available')
    spec = importlib.util.spec_from_file_location('check_isa', os.path.join(ROOT, 'tools', 'check_isa_packed_f32.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from dsnt import _lib
    total, bad = mod.scan(_lib.LIB_PATH)
    assert not bad, bad[:5]
