"""ctypes binding of ``libunibev_hip.so`` (C ABI declared in ``include/unibev_hip.h``).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
The product path never routes through a CPU implementation.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# UBV_LIB_PATH: load another build of the same library (kernel A/B studies); the default is the in-tree build
LIB_PATH = os.environ.get('UBV_LIB_PATH') or os.path.join(_HERE, 'libunibev_hip.so')

_lib = None

UBV_F32, UBV_F16, UBV_BF16 = 0, 1, 2

# name -> (restype, argtypes); mirrors include/unibev_hip.h one to one
_P = c_void_p
SIGNATURES = {
    'ubv_version': (c_int, []),
    'ubv_last_error': (c_char_p, []),
    'ubv_arch': (c_char_p, []),
    'ubv_debug_fill_lds': (c_int, [ctypes.c_uint32, _P]),
    'ubv_debug_aggressor': (c_int, [c_int, c_int, c_int, c_int, _P, c_int64, _P, _P]),
    'ubv_debug_ws_timing': (c_int, [_P]),
    'ubv_debug_set_wgrad_ws': (c_int, [c_int, c_int]),
    'ubv_profile_enable': (c_int, [c_int]),
    'ubv_profile_read': (c_int64, [c_char_p, c_int64]),
    'ubv_ms_deform_attn_forward': (c_int, [_P, _P, _P, _P, _P, _P] + [c_int] * 9 + [_P]),
    'ubv_ms_deform_attn_backward': (c_int, [_P] * 9 + [c_int] * 9 + [_P]),
    'ubv_ms_deform_attn_backward_workspace': (c_int64, [c_int] * 8),
    'ubv_ms_deform_attn_backward_planned': (c_int, [_P] * 9 + [c_int] * 10 + [_P, c_int64, _P]),
    'ubv_ms_deform_attn_grid_supported': (c_int, [c_int] * 10),
    'ubv_ms_deform_attn_forward_grid': (c_int, [_P] * 6 + [c_int] * 12 + [_P]),
    'ubv_ms_deform_attn_backward_grid_workspace': (c_int64, [c_int] * 10),
    'ubv_ms_deform_attn_backward_grid': (c_int, [_P] * 7 + [c_int] * 12 + [_P, c_int64, _P]),
    'ubv_bev_lift_forward_workspace': (c_int64, [c_int] * 8),
    'ubv_bev_lift_forward': (c_int, [_P, _P, c_int64, _P, c_int64, c_int, _P, _P, _P, _P]
                             + [c_int] * 12 + [_P, c_int64, _P]),
    'ubv_bev_lift_backward': (c_int, [_P, _P, c_int64, _P, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P,
                                      _P, c_int64, _P, c_int64] + [c_int] * 13 + [_P, _P, c_int64, _P]),
    'ubv_visible_lists_elems': (c_int64, [c_int, c_int]),
    'ubv_compact_visible': (c_int, [_P, c_int, c_int, _P, _P]),
    'ubv_compact_visible_grid': (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    'ubv_bev_lift_backward_workspace': (c_int64, [c_int] * 11),
    'ubv_bev_lift_supported': (c_int, [c_int] * 4),
    'ubv_point_sampling': (c_int, [_P, _P, _P, _P, ctypes.POINTER(c_float), c_float, c_float,
                                   _P, _P, _P, _P] + [c_int] * 5 + [_P]),
    'ubv_flatten_embed_forward': (c_int, [_P, _P, c_int, _P, _P] + [c_int] * 4 + [_P]),
    'ubv_flatten_embed_backward': (c_int, [_P, _P, _P] + [c_int] * 4 + [_P]),
    'ubv_bev_fuse_forward': (c_int, [_P] * 7 + [c_int] * 5 + [_P]),
    'ubv_bev_fuse_backward': (c_int, [_P] * 11 + [c_int] * 5 + [_P]),
    'ubv_add_dropout_layernorm_forward': (c_int, [_P] * 7 + [c_int64, c_int64, c_int, c_float, c_float, c_uint64,
                                                          _P, c_int, c_int, _P]),
    'ubv_add_dropout_layernorm_backward': (c_int, [_P] * 11 + [c_int64, c_int64, c_int, c_float, c_uint64,
                                                            _P, c_int, c_int, _P, _P]),
    'ubv_add_dropout_layernorm_backward_workspace': (c_int64, [c_int]),
    'ubv_relu_dropout_forward': (c_int, [_P, _P, c_int64, c_float, c_uint64, _P, c_int, _P]),
    'ubv_relu_dropout_backward': (c_int, [_P, _P, _P, c_int64, c_float, c_int, _P]),
    'ubv_linear_workspace': (c_int64, []),
    'ubv_linear_forward': (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_int, _P, c_int64, _P]),
    'ubv_gemm_nt': (c_int, [_P, c_int64, _P, _P, c_int64, _P, _P, _P, c_int64, c_int64, c_int, c_int, c_int, _P]),
    'ubv_gemm_nt_dual': (c_int, [_P, c_int64, _P, c_int64, c_int, _P, _P, c_int64, _P, _P, _P, c_int64, c_int64, _P, c_int64,
                                 _P, c_int64, c_int, c_int64, c_int, c_int, _P]),
    'ubv_gemm_wgrad_dual': (c_int, [_P, _P, c_int, _P, _P, _P, c_int64, c_int, c_int, c_int, _P]),
    'ubv_gemm_nt_rowbias': (c_int, [_P, c_int64, _P, _P, c_int64, _P, _P, c_int64, c_int64, _P, c_int64, c_int64, c_int,
                                    c_int, c_int, _P]),
    'ubv_gemm_nt_act': (c_int, [_P, c_int64, _P, _P, c_int64, _P, _P, c_int64, c_int64, c_int, c_int, c_int,
                                c_int, _P, c_float, c_uint64, _P, _P]),
    'ubv_grid_mask': (c_int, [_P, _P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'ubv_dcn_im2col': (c_int, [_P, _P, _P, _P] + [c_int] * 16 + [_P]),
    'ubv_dcn_col2im': (c_int, [_P] * 7 + [c_int] * 16 + [_P]),
    'ubv_split_weights_batched': (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'ubv_sumsq_workspace': (c_int64, []),
    'ubv_sumsq_f32': (c_int, [_P, c_int64, _P, _P, _P]),
    'ubv_adamw_flat': (c_int, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, _P, _P, c_float, _P]),
    'ubv_adamw_flat_max_groups': (c_int, []),
    'ubv_adamw_flat_groups': (c_int, [_P, _P, _P, _P, c_int64, c_int, _P, _P, _P, c_float, c_float, c_float, _P, _P,
                                      c_float, _P]),
    'ubv_spconv_table_slots': (c_int64, [c_int64]),
    'ubv_spconv_wgrad_splits': (c_int, [c_int64, c_int]),
    'ubv_spconv_wgrad': (c_int, [_P, _P, _P, c_int64, c_int64, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'ubv_rows_bn_partial_elems': (c_int64, [c_int]),
    'ubv_rows_bn_forward': (c_int, [_P] * 10 + [c_int64, c_int, c_float, c_float, c_int, c_int, c_int, _P]),
    'ubv_rows_bn_backward': (c_int, [_P] * 12 + [c_int64, c_int, c_int, c_int, _P]),
    'ubv_spconv_wgrad_pairs': (c_int, [_P, _P, _P, _P, _P, c_int64, c_int64, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'ubv_spconv_hash_build': (c_int, [_P, c_int64, c_int, c_int, c_int, _P, _P, c_int64, _P]),
    'ubv_spconv_neighbors': (c_int, [_P, c_int64, c_int, _P, _P, _P, _P, _P, c_int, _P, _P, c_int64, _P, c_int64, _P]),
    'ubv_spconv_sites_words': (c_int64, [c_int, _P]),
    'ubv_spconv_output_sites': (c_int, [_P, _P, c_int64, c_int, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, c_int64, _P, _P]),
    'ubv_spconv_pairs_chunks': (c_int64, [c_int64]),
    'ubv_spconv_pairs': (c_int, [_P, c_int64, c_int64, c_int, _P, _P, _P, _P, _P]),
    'ubv_spconv_weight_operand': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    'ubv_spconv_gather_mma': (c_int, [_P, _P, c_int64, c_int64, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'ubv_gemm_wgrad_splits': (c_int, [c_int64, c_int, c_int]),
    'ubv_gemm_wgrad': (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, _P]),
    'ubv_split_weight': (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P]),
    'ubv_linear_grad_reduce': (c_int, [_P, c_int64, c_int, _P, _P, c_int, c_int64, _P, c_int, _P]),
    'ubv_add2_f32': (c_int, [_P, _P, _P, c_int64, _P]),
    'ubv_slice_sum_f32': (c_int, [_P, c_int, c_int64, c_int, _P, c_int64, _P]),
    'ubv_hard_voxelize_workspace': (c_int64, [c_int, c_int, c_int]),
    'ubv_hard_voxelize_batch_workspace': (c_int64, [c_int, c_int, c_int, c_int]),
    'ubv_hard_voxelize_batch': (c_int, [_P, _P, c_int, _P, _P, _P, _P, _P, c_int64, c_int, ctypes.POINTER(c_float),
                                        ctypes.POINTER(c_float), c_int, c_int, _P]),
    'ubv_hard_voxelize_batch_vfe': (c_int, [_P, _P, c_int, _P, _P, _P, _P, _P, _P, c_int64, c_int, ctypes.POINTER(c_float),
                                            ctypes.POINTER(c_float), c_int, c_int, _P]),
    'ubv_hard_voxelize': (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, c_int,
                                  ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_int, c_int,
                                  _P]),
    'ubv_dynamic_voxelize': (c_int, [_P, _P, c_int, c_int, ctypes.POINTER(c_float),
                                     ctypes.POINTER(c_float), _P]),
    'ubv_dynamic_scatter_workspace': (c_int64, [c_int]),
    'ubv_dynamic_point_to_voxel_forward': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P,
                                                   _P, c_int64, _P]),
    'ubv_voxel_mean': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    'ubv_sparse_to_dense': (c_int, [_P, _P, _P, c_int, _P] + [c_int] * 5 + [_P]),
}


class UniBEVHipError(RuntimeError):
    pass


_CALL_HOOK = [None]          # debug.ForeignKernelLog: called with the entry point's name before every call


class _Hooked:
    """The handle with a hook in front of every entry point (debug mode only: ``set_call_hook``)."""

    def __init__(self, handle):
        self._h = handle

    def __getattr__(self, name):
        fn = getattr(self._h, name)
        hook = _CALL_HOOK[0]
        if hook is None or not name.startswith('ubv_'):
            return fn

        def call(*a):
            hook(name)
            return fn(*a)
        return call


def set_call_hook(fn):
    """Install (None: remove) a function called with the name of every C-ABI entry point about to be called."""
    prev, _CALL_HOOK[0] = _CALL_HOOK[0], fn
    return prev


def lib():
    """Load (once) and return the ctypes handle; raise loudly if it is absent."""
    global _lib
    if _lib is not None and _CALL_HOOK[0] is not None:
        return _Hooked(_lib)
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UniBEVHipError(
                f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; '
                f'g.build()"` (or `make -C unibev_amd/csrc`). There is no CPU fallback.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().ubv_last_error().decode(errors='replace')
        raise UniBEVHipError(f'{what} failed (status {status}): {msg}')


def float_array(values):
    return (c_float * len(values))(*[float(v) for v in values])
