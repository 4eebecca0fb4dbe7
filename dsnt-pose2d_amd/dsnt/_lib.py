"""ctypes binding of libdsnt_hip.so (the C ABI declared in include/dsnt_hip.h).

There is no CPU fallback: if the library is missing or a tensor is not on a HIP device the
call raises.  Kernels are enqueued on PyTorch's current stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSNT_HIP_LIB points at another build of the same library (kernel experiments); default: the in-tree build
LIB_PATH = os.environ.get('DSNT_HIP_LIB') or os.path.join(os.path.dirname(_HERE), 'csrc', 'libdsnt_hip.so')

P = C.c_void_p
I = C.c_int
L = C.c_int64
F = C.c_float
D = C.c_double


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ('N', 'H', 'W', 'Cin', 'Ho', 'Wo', 'Cout', 'R', 'S', 'stride', 'pad', 'dil')]


GP = C.POINTER(ConvGeom)


class BnBwdEpilogue(C.Structure):
    _fields_ = [('x', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p), ('mean', C.c_void_p),
                ('invstd', C.c_void_p), ('relu', C.c_int)]


BP = C.POINTER(BnBwdEpilogue)


class BnPrologue(C.Structure):
    """dsnt_bn_prologue: a BatchNorm finalised in the prologue of the launch that consumes it."""
    _fields_ = [('partial', C.c_void_p), ('tiles', C.c_int), ('C', C.c_int), ('M', C.c_int64),
                ('gamma', C.c_void_p), ('beta', C.c_void_p), ('running_mean', C.c_void_p), ('running_var', C.c_void_p),
                ('momentum', C.c_float), ('eps', C.c_float),
                ('mean', C.c_void_p), ('invstd', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p)]


PP = C.POINTER(BnPrologue)


class BnTail(C.Structure):
    """dsnt_out_bounds: the fp16x3 operand bounds a launch leaves behind for the consumers of its output.  (The class keeps the
    name it had while the struct also described a BatchNorm finalisation in the producer's last workgroup.)"""
    _fields_ = [('amax', C.c_void_p), ('amax_bn', C.c_void_p), ('amax_scale', C.c_void_p), ('amax_shift', C.c_void_p),
                ('amax_relu', C.c_int), ('reserved', C.c_int)]


TP = C.POINTER(BnTail)


class BnBwdApply(C.Structure):
    """dsnt_bn_bwd_apply: the BatchNorm backward of a convolution's consumer, folded into dsnt_conv1x1_bwd_f16x3."""
    _fields_ = [('y', C.c_void_p), ('scale', C.c_void_p), ('mean', C.c_void_p), ('invstd', C.c_void_p),
                ('coef', C.c_void_p)]


AP = C.POINTER(BnBwdApply)

# name -> argtypes (the trailing `void* stream` included where the C signature has it)
SIGNATURES = {
    'dsnt_preact_fwd': [P, P, L, I, I, F, F, P],
    'dsnt_preact_bwd': [P, P, P, P, L, I, I, F, F, P],
    'dsnt_expect_fwd': [P, P, L, I, I, P],
    'dsnt_expect_bwd': [P, P, L, I, I, P],
    'dsnt_make_gauss': [P, P, L, I, I, F, P],
    'dsnt_make_gauss_bwd': [P, P, P, L, I, I, F, P],
    'dsnt_encode_heatmaps': [P, P, L, I, I, F, P],
    'dsnt_heatmap_mse_fwd': [P, P, P, L, I, I, F, P],
    'dsnt_heatmap_mse_bwd': [P, P, P, P, L, I, I, F, P],
    'dsnt_decode_heatmaps': [P, P, L, I, I, I, P],
    'dsnt_fc2_fwd': [P, P, P, P, L, I, P],
    'dsnt_fc2_bwd': [P, P, P, P, P, P, L, I, P],
    'dsnt_reg_fwd': [P, P, P, L, I, I, F, I, P],
    'dsnt_reg_bwd': [P, P, P, P, L, I, I, F, I, P],
    'dsnt_reg_bwd_mu': [P, P, P, P, L, I, I, F, I, P],
    'dsnt_euclid_fwd': [P, P, P, L, I, P],
    'dsnt_euclid_bwd': [P, P, P, P, P, L, I, P],
    'dsnt_masked_avg_fwd': [P, P, P, L, P],
    'dsnt_masked_avg_bwd': [P, P, P, P, L, P],
    'dsnt_head_fwd': [P, P, P, L, I, I, P],
    'dsnt_head_loss_rows': [P, P, P, P, P, L, I, I, F, I, P],
    'dsnt_head_bwd': [P, P, P, P, P, P, P, L, I, I, F, I, P],
    'dsnt_head_loss_grad': [P, P, P, P, P, P, P, P, L, I, I, F, I, F, P],
    'dsnt_mask_denom': [P, P, L, P],
    'dsnt_head_loss_reduce': [P, P, P, P, F, P, P, L, P],
    'dsnt_scale_by_scalar': [P, P, L, P],
    'dsnt_conv_fwd': [P, P, P, P, P, P, I, P, P, P, GP, P],
    'dsnt_conv_fwd_ex': [P, P, P, P, P, P, I, P, P, P, GP, BP, TP, P],
    'dsnt_conv_fwd_bf16x6_ex': [P, P, L, P, P, P, P, I, P, P, P, GP, BP, TP, P],
    'dsnt_conv_pack_dgrad': [P, P, I, I, I, I, P],
    'dsnt_conv_pack_dgrad_all': [P, I, P, P, P, L, P],
    'dsnt_conv_fwd_bf16x6': [P, P, L, P, P, P, P, I, P, P, P, GP, P],
    'dsnt_split_bf16x3': [P, P, L, P],
    'dsnt_conv_fwd_f16x3_ex': [P, P, L, P, P, P, P, P, P, I, P, P, P, GP, BP, TP, P],
    'dsnt_conv_fwd_f16x3_stream': [P, P, L, P, P, P, P, P, P, I, P, P, P, GP, BP, TP, P],
    'dsnt_amax': [P, L, P, P],
    'dsnt_split_f16x2': [P, P, L, L, P, P],
    'dsnt_f16_prep_weights': [P, I, I, P],
    'dsnt_f16_prep_bn_bounds': [P, I, P],
    'dsnt_conv_wgrad': [P, P, P, I, P, P, P, P, I, GP, P],
    'dsnt_conv_wgrad_bf16x6': [P, P, P, I, P, P, P, P, I, GP, P],
    'dsnt_conv_wgrad_f16x3': [P, P, P, I, P, P, P, P, I, P, P, GP, P],
    'dsnt_conv1x1_fwd_f16x3': [P, P, L, P, P, P, P, P, P, I, P, P, GP, TP, P],
    'dsnt_conv1x1_bwd_f16x3': [BP, P, AP, P, L, P, P, P, P, P, P, P, I, GP, P],
    'dsnt_conv_dgrad_f16x3_stream_apply': [P, AP, P, P, L, P, P, P, P, I, GP, BP, TP, P],
    'dsnt_wgrad_reduce_all': [P, I, I, P],
    'dsnt_conv_wgrad_group': [P, I, I, P],
    'dsnt_bn_stats': [P, P, L, I, P],
    'dsnt_bn_finalize': [P, I, L, I, P, P, P, P, F, F, I, P, P, P, P, P],
    'dsnt_bn_act_fwd': [P, P, P, I, P, L, I, P],
    'dsnt_bn_act_bwd_reduce': [P, P, P, P, P, P, I, P, L, I, P],
    'dsnt_bn_bwd_finalize': [P, I, L, I, P, P, I, P, P],
    'dsnt_bn_bwd_finalize_bound': [P, I, L, I, P, P, I, P, P, P, P, P],
    'dsnt_bn_eval_prep': [P, I, P],
    'dsnt_bn_act_bwd_apply': [P, P, P, P, P, P, P, I, P, I, L, I, P],
    'dsnt_bn_act_bwd_apply_amax': [P, P, P, P, P, P, P, I, P, I, L, I, P, P],
    'dsnt_bn_act_bwd_apply_pro': [P, P, P, P, P, P, P, I, P, P, I, P, I, P, I, L, I, P, P],
    'dsnt_stem4_fwd_f16x3': [P, P, L, P, P, P, P, P, GP, TP, P],
    'dsnt_bn_add_act_bwd_reduce': [P, P, P, P, P, I, P, P, L, I, P],
    'dsnt_bn_act_bwd_apply_base': [P, P, P, P, P, P, P, I, P, P, L, I, P, P],
    'dsnt_bn_act_bwd_apply_pro_base': [P, P, P, P, P, P, P, I, P, P, I, P, I, P, P, L, I, P, P],
    'dsnt_conv_fwd_pro': [P, P, P, P, PP, I, P, P, P, GP, TP, P],
    'dsnt_fill_zero': [P, L, P],
    'dsnt_axpy_amax': [P, P, F, I, L, P, P],
    'dsnt_maxpool2_bwd_amax': [P, P, P, I, I, I, I, I, P, P],
    'dsnt_maxpool2_bwd_add': [P, P, P, I, P, I, I, I, I, P, P],
    'dsnt_upsample2_bwd_amax': [P, P, I, I, I, I, I, P, P],
    'dsnt_maxpool2_fwd': [P, P, P, I, I, I, I, P],
    'dsnt_maxpool2_bwd': [P, P, P, I, I, I, I, I, P],
    'dsnt_maxpool2_fwd_stats': [P, P, P, P, I, I, I, I, TP, P],
    'dsnt_upsample2_add_fwd_stats': [P, P, P, P, I, I, I, I, TP, P],
    'dsnt_bn_act_fwd_stats': [P, P, P, I, P, P, L, I, TP, P],
    'dsnt_s2d_input': [P, P, I, I, I, I, TP, P],
    'dsnt_s2d_weights': [P, P, I, I, P],
    'dsnt_s2d_weights_prep': [P, P, P, P, P, I, P],
    'dsnt_maxpool3s2_fwd': [P, P, P, I, I, I, I, P],
    'dsnt_maxpool3s2_bwd': [P, P, P, I, I, I, I, I, P],
    'dsnt_bn_add_act_fwd': [P, P, P, P, I, P, L, I, P],
    'dsnt_relu_bwd': [P, P, P, L, P],
    'dsnt_zero_insert': [P, P, I, I, I, I, I, I, I, P],
    'dsnt_conv_dgrad_strided': [P, P, P, P, P, GP, BP, TP, P],
    'dsnt_upsample2_add_fwd': [P, P, P, I, I, I, I, P],
    'dsnt_upsample2_bwd': [P, P, I, I, I, I, I, P],
    'dsnt_axpy': [P, P, F, I, L, P],
    'dsnt_nchw_to_nhwc': [P, P, I, I, I, I, P],
    'dsnt_nhwc_to_nchw': [P, P, I, I, I, I, P],
    'dsnt_rmsprop_step': [P, P, P, L, F, F, F, F, F, P],
    'dsnt_sgd_step': [P, P, P, L, F, F, F, F, I, P],
    'dsnt_rmsprop_step_guarded': [P, P, P, L, F, F, F, F, F, P, P],
    'dsnt_sgd_step_guarded': [P, P, P, L, F, F, F, F, I, P, P],
    'dsnt_nonfinite_flag': [P, L, P, I, P],
    'dsnt_pckh': [P, P, P, P, P, P, F, P, P, I, I, P],
    'dsnt_debug_mfma_peak': [P, I, I, I, I, P],
    'dsnt_debug_coexec': [P, I, I, I, P],
    'dsnt_debug_bf16_peak': [P, I, I, I, I, P],
    'dsnt_debug_starve': [P, P, I, I, I, I, P],
    'dsnt_debug_grid_barrier': [P, I, I, I, P, P],
    'dsnt_debug_grid_barrier2': [P, I, I, I, P, P],
    'dsnt_debug_empty': [I, I, P, P],
    'dsnt_debug_force_gemm6': [I],
}
# entry points without the status/stream convention
PLAIN = {
    'dsnt_version': (I, []),
    'dsnt_last_error': (C.c_char_p, []),
    'dsnt_conv_fwd_bm': (I, [GP]),
    'dsnt_list_create': (P, []),
    'dsnt_list_destroy': (None, [P]),
    'dsnt_list_begin': (I, [P]),
    'dsnt_list_end': (I, []),
    'dsnt_list_sync': (I, [P, I, I]),
    'dsnt_list_mark': (I, [P]),
    'dsnt_list_segments': (I, [P]),
    'dsnt_list_size': (I, [P]),
    'dsnt_list_replay': (I, [P, I, C.POINTER(C.c_void_p), I]),
    'dsnt_list_fuse_bytes': (C.c_int64, [P, I, I]),
    'dsnt_list_fuse': (I, [P, P, C.c_int64, I, I, I]),
    'dsnt_list_stages': (I, [P, C.POINTER(C.c_int)]),
    'dsnt_list_fuse_plan': (I, [P, I, I, C.POINTER(C.c_int)]),
    'dsnt_list_stage_errors': (I, [P]),
    'dsnt_conv_bf16x6_ok': (I, [GP]),
    'dsnt_conv_wgrad_bf16x6_ok': (I, [GP]),
    'dsnt_conv_wgrad_splits': (I, [GP]),
    'dsnt_conv_wgrad_halo_ok': (I, [GP]),
    'dsnt_conv_fwd_stream_ok': (I, [GP]),
    'dsnt_conv_fwd_stream_form': (I, [GP, I]),
    'dsnt_conv_dgrad_strided_ok': (I, [GP]),
    'dsnt_conv_dgrad_strided_tiles': (I, [GP]),
    'dsnt_conv_fwd_pro_ok': (I, [GP, I, I]),
    'dsnt_conv_wgrad_f16x3_splits': (I, [GP, I]),
    'dsnt_conv_wgrad_f16x3_ws_floats': (L, [GP, I]),
    'dsnt_conv_wgrad_ws_floats': (L, [GP]),
    'dsnt_conv1x1_fwd_ok': (I, [GP]),
    'dsnt_conv1x1_fwd_stats_rows': (I, [GP, I]),
    'dsnt_stem4_fwd_ok': (I, [GP]),
    'dsnt_stem4_fwd_stats_rows': (I, [GP]),
    'dsnt_conv1x1_bwd_ok': (I, [GP]),
    'dsnt_conv1x1_bwd_splits': (I, [GP, I]),
    'dsnt_conv1x1_bwd_ws_floats': (L, [GP, I]),
    'dsnt_conv_wgrad_desc_bytes': (I, []),
    'dsnt_conv_wgrad_desc': (I, [P, P, P, I, P, P, GP, P]),
    'dsnt_conv_wgrad_desc_f16x3': (I, [P, P, P, I, P, P, P, P, GP, P]),
    'dsnt_debug_set_timeline': (I, [P, I]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libdsnt_hip.so not found at %s — build it with `python dsnt-pose2d_amd/build.py` '
                '(there is no CPU fallback for the dsnt hot path)' % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = I
        for name, (res, argtypes) in PLAIN.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = res
        _lib = lib
    return _lib


def fn(name):
    return getattr(load(), name)


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Invoke a status-returning entry point on the current stream; raise on error."""
    lib = load()
    rc = getattr(lib, name)(*args, stream_ptr())
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (name, rc, lib.dsnt_last_error().decode()))


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses anything not resident on a GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('dsnt: tensor is on %s; the dsnt hot path runs on the HIP device only '
                           '(no CPU fallback)' % t.device)
    if not t.is_contiguous():
        raise RuntimeError('dsnt: tensor must be contiguous')
    return t.data_ptr()


def f32(t):
    if t.dtype != torch.float32:
        raise RuntimeError('dsnt: expected float32, got %s (the HIP path computes in fp32)' % t.dtype)
    return t
