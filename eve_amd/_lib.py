"""ctypes binding of libeve_hip.so (include/eve_hip.h).  No fallback: if the library is missing the
product path raises -- there is no CPU implementation of the hot path in this package."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_longlong, c_uint, c_void_p

from .build import LIB_PATH


class EveLibraryError(RuntimeError):
    pass


class ConvDesc(Structure):
    _fields_ = [(n, c_int) for n in ('dtype', 'N', 'IH', 'IW', 'Cin', 'OH', 'OW', 'Cout',
                                     'KH', 'KW', 'stride', 'pad')]


class DispatchConfig(Structure):
    """include/eve_hip.h eve_dispatch_config: the kernel-selection table, resolved once when the library is loaded."""
    _fields_ = [(n, c_int) for n in (
        'struct_bytes', 'conv_impl_v1', 'conv_tile_big', 'conv_halo', 'conv_ws64', 'conv_wg8', 'conv_wg8_min_tiles',
        'conv_wg8_s2_min_tiles', 'halo_persist', 'wgrad_target_wgs', 'wgrad_min_rows', 'wgrad_halo', 'wgrad_wg8',
        'wg64_th', 'wg64_nreg', 'wg64_fixed', 'in_split', 'in_min_threads', 'in_stats_one_pass', 'stem_split',
        'in_trunk_kernels', 'stem_fused_wgrad', 'stem_fwd_pairs', 'conv1x1_stream', 'conv3x3_stream', 'in_big_planes', 'cgru_seq_max_b',
        'cgru_scan', 'small_linear', 'tail_loss_node', 'bucket_elems', 'gate_wait_polls')] + [('wgrad_halo_min_m', c_longlong)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != 'struct_bytes'}


class WgradProblem(Structure):
    """include/eve_hip.h eve_wgrad_problem"""
    _fields_ = [('dY', c_void_p), ('Y', c_void_p), ('X', c_void_p), ('X2', c_void_p), ('dW', c_void_p), ('db', c_void_p),
                ('M', c_int), ('N', c_int), ('K', c_int), ('K1', c_int), ('K2', c_int), ('act', c_int), ('rows_per_split', c_int),
                ('ldY', c_int), ('ldX', c_int), ('ldW', c_int), ('x_shift_T', c_int), ('reserved', c_int)]


WGRAD_BATCH_MAX = 12

P = c_void_p
I = c_int
L = c_longlong
F = c_float

# name -> argtypes; every function returns int (0 = ok) except the two noted below
class PackItem(Structure):
    _fields_ = [('w_ohwi', c_void_p), ('dst_ohwi', c_void_p), ('dst_ihwo', c_void_p),
                ('Cout', c_int), ('taps', c_int), ('Cin', c_int), ('src_Cout', c_int), ('src_Cin', c_int)]


class VecTerm(Structure):
    """include/eve_hip.h eve_vec_term"""
    _fields_ = [('pred', c_void_p), ('tgt', c_void_p), ('valid', c_void_p), ('dpred', c_void_p), ('D', c_int), ('kind', c_int)]


VEC_TERMS_MAX = 32
PACK_BATCH_MAX = 48
ABI_VERSION = 10         # include/eve_hip.h EVE_ABI_VERSION

SIGNATURES = {
    'eve_conv2d_fwd': [POINTER(ConvDesc), P, P, P, I, P, I, P, P],
    'eve_conv2d_fwd_stats': [POINTER(ConvDesc), P, P, P, I, P, P, F, POINTER(c_int), P],
    'eve_conv2d_dgrad': [POINTER(ConvDesc), P, P, P, P, ctypes.c_ulonglong, P],
    'eve_conv2d_dgrad_acc': [POINTER(ConvDesc), P, P, P, P],
    'eve_conv2d_wgrad': [POINTER(ConvDesc), P, P, P, I, P, P, ctypes.c_ulonglong, P],
    'eve_conv2d_wgrad_bias': [POINTER(ConvDesc), P, P, P, P, P, ctypes.c_ulonglong, P],
    'eve_get_dispatch_config': [POINTER(DispatchConfig)],
    'eve_get_default_dispatch_config': [POINTER(DispatchConfig)],
    'eve_set_dispatch_config': [POINTER(DispatchConfig)],
    'eve_stem_pack_input': [I, I, I, I, I, P, P, P],
    'eve_frames_u8_to_nchw': [L, I, I, I, P, F, F, I, P, P],
    'eve_frames_u8_to_stem': [I, L, I, I, I, P, F, F, P, P],
    'eve_stem7x7s2_fwd': [I, I, I, I, P, P, P, P],
    'eve_stem_fwd_fused': [I, I, I, I, P, P, F, P, P, P, P],
    'eve_stem_wgrad': [I, I, I, I, P, P, P, P],
    'eve_stem_bwd_dx': [I, I, I, I, P, P, P, P, P, P, P, P, P],
    'eve_stem_bwd_wgrad': [I, I, I, I, P, P, P, P, P, P, P, P, P, ctypes.c_ulonglong, P],
    'eve_stem_bwd_wgrad_workspace': [I, I, I],
    'eve_bias_grad': [I, L, I, P, P, P],
    'eve_cgru_scan_fwd': [I, I, I, P, P, P, P, P, P, P, P, P, P, P, P],
    'eve_cgru_scan_bwd': [I, I, I, P, P, P, P, P, P, P, P, P, P, P, P],
    'eve_vector_terms': [POINTER(VecTerm), I, I, I, P, P],
    'eve_gate_signal': [P, P],
    'eve_gate_wait': [P, c_uint, P, P, P, c_uint, P],
    'eve_crnn_scan_fwd': [I, I, P, P, P, P, P, P, P],
    'eve_crnn_scan_bwd': [I, I, P, P, P, P, P, P, P],
    'eve_clstm_scan_fwd': [I, I, P, P, P, P, P, P, P, P],
    'eve_rnn_scan_fwd': [I, I, I, P, P, P, P, P, P],
    'eve_rnn_scan_bwd': [I, I, I, P, P, P, P, P, P],
    'eve_lstm_scan_fwd': [I, I, I, P, P, P, P, P, P, P, P, P],
    'eve_lstm_scan_bwd': [I, I, I, P, P, P, P, P, P, P, P, P, P, P],
    'eve_pack_weights_batch': [I, I, P, P],
    'eve_eye_losses': [I, I, P, P, P, P, P, P, F, F, P, P, P, P],
    'eve_gaze_to_pog': [L, P, P, P, P, P, P, P, F, F, P, P, P, P, P],
    'eve_gaze_to_pog_bwd': [L, P, P, P, P, P, P],
    'eve_combined_gaze': [L, P, P, P, P, P, P],
    'eve_make_heatmaps': [L, I, I, P, P, F, F, F, P, P],
    'eve_make_heatmaps_bwd': [L, I, I, P, F, F, F, P, P, P],
    'eve_soft_argmax_fwd': [L, I, I, P, F, F, P, P, P],
    'eve_soft_argmax_bwd': [L, I, I, P, P, P, F, F, P, P],
    'eve_linear_fwd': [I, I, I, P, P, P, I, P, P],
    'eve_linear_dgrad': [I, I, I, P, P, I, P, P, P],
    'eve_linear_wgrad': [I, I, I, P, P, I, P, P, P, P],
    'eve_linear_fwd_ex': [I, I, I, P, I, P, P, I, I, P, I, P],
    'eve_linear_dgrad_ex': [I, I, I, P, I, P, I, P, P, I, I, P],
    'eve_tail_head_pose': [I, P, P, P, I, I, P],
    'eve_tail_outputs_fwd': [I, P, P, P, P, P],
    'eve_tail_outputs_bwd': [I, P, P, P, P, P, F, F, P, P, P],
    'eve_linear_wgrad_batch': [POINTER(WgradProblem), I, P],
    'eve_instnorm_stats': [I, I, I, I, P, F, P, P],
    'eve_instnorm_act_fwd': [I, I, I, I, P, P, P, P, P, I, P, P],
    'eve_instnorm_act_bwd': [I, I, I, I, P, P, P, P, P, P, I, P, P, P, P, P],
    'eve_instnorm_act2_fwd': [I, I, I, I, P, P, P, P, P, P, I, P, P, I, I, P, P, P],
    'eve_instnorm_act2_bwd': [I, I, I, I, P, P, I, P, P, P, P, P, P, I, P, P, P, I, P, P, P, P],
    'eve_instnorm_fwd_fused': [I, I, I, I, P, P, P, P, I, F, P, P, P, P],
    'eve_instnorm_bwd_fused': [I, I, I, I, P, P, P, P, P, P, P, I, P, P, P, P, P, P],
    'eve_sum_rows': [I, I, P, P, P],
    'eve_sum_rows_pairs': [I, I, P, P, P, P],
    'eve_act_bwd': [I, L, P, P, I, P, P],
    'eve_add': [I, L, P, P, P, P],
    'eve_maxpool3x3s2_fwd': [I, I, I, I, I, P, P, P, P],
    'eve_maxpool3x3s2_bwd': [I, I, I, I, I, P, P, P, P],
    'eve_in_relu_maxpool_fwd': [I, I, I, I, I, P, P, P, P, P],
    'eve_in_relu_maxpool_bwd': [I, I, I, I, I, P, P, P, P, P, P, P],
    'eve_avgpool_fwd': [I, I, I, I, P, P, P],
    'eve_avgpool_bwd': [I, I, I, I, P, P, P],
    'eve_avgpool_fwd_f32': [I, I, I, I, P, P, P],
    'eve_avgpool_bwd_f32': [I, I, I, I, P, P, P],
    'eve_adaptive_maxpool_fwd': [I, I, I, I, I, I, I, P, P, P, P],
    'eve_adaptive_maxpool_bwd': [I, I, I, I, I, I, I, P, P, P, P, P],
    'eve_bilinear_fwd': [I, I, I, I, I, I, I, P, P, P],
    'eve_bilinear_bwd': [I, I, I, I, I, I, I, P, P, P],
    'eve_nchw_to_nhwc': [I, I, I, I, I, I, P, P, P],
    'eve_nhwc_to_nchw': [I, I, I, I, I, I, P, P, P],
    'eve_cast': [I, I, L, P, P, P],
    'eve_pack_weights': [I, I, I, I, P, P, P, P],
    'eve_gru_scan_fwd': [I, I, I, P, P, P, P, P, P, P, P],
    'eve_gru_scan_bwd': [I, I, I, P, P, P, P, P, P, P, P, P, P],
    'eve_cgru_gates1': [I, L, I, P, P, P, P, P],
    'eve_cgru_gates2': [I, L, I, P, P, P, P, P, P],
    'eve_cgru_gates2_bwd': [I, L, I, P, P, P, P, P, P, P, P],
    'eve_cgru_gates1_bwd': [I, L, I, P, P, P, P, P, P, P],
    'eve_clstm_gates_fwd': [I, L, I, P, P, P, P, P],
    'eve_heatmap_head_fwd': [I, L, I, P, P, P],
    'eve_heatmap_head_bwd': [I, L, I, P, P, P, P],
    'eve_heatmap_loss_fwd': [I, I, I, I, P, P, P, P, P, P, P],
    'eve_heatmap_loss_bwd': [I, I, I, P, P, P, P, P, P],
    'eve_sumsq': [L, P, P, P, P],
    'eve_adam_step': [L, P, P, P, P, P, F, F, F, F, F, F, F, I, P, I, P, P, P],
}
EXPORTS = sorted(list(SIGNATURES) + ['eve_abi_version', 'eve_last_error', 'eve_last_kernel'])

_lib = None


def load(path=None):
    """Load (once) and type the library.  Raises EveLibraryError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get('EVE_HIP_LIB', LIB_PATH)
    if not os.path.isfile(path):
        raise EveLibraryError(
            'libeve_hip.so not found at %s -- build it with `python -m eve_amd.build` '
            '(hipcc, gfx950).  eve_amd has no CPU fallback for the hot path.' % path)
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.eve_stem_bwd_wgrad_workspace.restype = ctypes.c_ulonglong
    lib.eve_last_kernel.argtypes = []
    lib.eve_last_kernel.restype = ctypes.c_char_p
    lib.eve_abi_version.argtypes = []
    lib.eve_abi_version.restype = c_int
    lib.eve_last_error.argtypes = []
    lib.eve_last_error.restype = c_char_p
    if lib.eve_abi_version() != ABI_VERSION:
        raise EveLibraryError('libeve_hip.so ABI version %d, expected %d (rebuild: python -m eve_amd.build)' % (
            lib.eve_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(status, lib=None):
    if status != 0:
        lib = lib or load()
        raise RuntimeError('libeve_hip: ' + lib.eve_last_error().decode('utf-8', 'replace'))
