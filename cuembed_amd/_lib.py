"""ctypes binding of libcuembed_amd.so (the C ABI declared in include/cuembed_amd.h).

There is deliberately no fallback: if the HIP library is missing or cannot be
loaded, importing the symbol table raises.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# CUEMBED_AMD_LIB: load another build of the same sources (tuning experiments: tools/); never a fallback
LIB_PATH = os.environ.get("CUEMBED_AMD_LIB") or os.path.join(_PKG, "lib", "libcuembed_amd.so")

_lib = None

_VP = ctypes.c_void_p
_I = ctypes.c_int


class CuembedLibraryError(RuntimeError):
    pass


def _declare(L):
    L.cuembed_embedding_forward.restype = None
    L.cuembed_embedding_forward.argtypes = [_VP, _I, _I, _VP, _I, _VP, _I, _VP, _I, _I, _I, _I, _VP, _VP]
    L.cuembed_embedding_forward_with_options.restype = None
    L.cuembed_embedding_forward_with_options.argtypes = [_VP, _I, _I, _VP, _I, _VP, _I, _VP, _I, _I, _I, _I, _VP, _I, _I,
                                                         _VP]
    L.cuembed_embedding_forward_ordered.restype = None
    L.cuembed_embedding_forward_ordered.argtypes = [_VP, _I, _I, _VP, _I, _VP, _I, _VP, _I, _I, _I, _I, _VP, _I, _I, _VP,
                                                    _VP]
    L.cuembed_embedding_forward_device_hints.restype = None
    L.cuembed_embedding_forward_device_hints.argtypes = [_VP, _I, _I, _VP, _I, _VP, _I, _VP, _I, _I, _I, _I, _VP, _I, _I,
                                                         _VP, _VP, _VP]
    L.cuembed_decide_row_loads.restype = None
    L.cuembed_decide_row_loads.argtypes = [_VP, _I, ctypes.c_int64, ctypes.c_int64, _VP, ctypes.c_uint, _VP]
    L.cuembed_bag_order_by_length.restype = None
    L.cuembed_bag_order_by_length.argtypes = [_VP, _I, _I, _I, _VP, _VP, ctypes.POINTER(ctypes.c_size_t), _VP]
    _L = ctypes.c_int64
    L.cuembed_exchange_pack_rows.restype = None
    L.cuembed_exchange_pack_rows.argtypes = [_VP, _I, _VP, _I, _L, _I, _VP, _VP, _I, _L, _L, _L, _VP, _VP, _VP, _VP, _VP]
    L.cuembed_exchange_finish_piece.restype = None
    L.cuembed_exchange_finish_piece.argtypes = [_VP, _VP, _L, _L, _L, _L, _L, _VP, _VP, _I, _I, _VP, _VP, _VP, _VP]
    L.cuembed_set_forward_row_load_policy.restype = None
    L.cuembed_set_forward_row_load_policy.argtypes = [_I]
    L.cuembed_get_forward_row_load_policy.restype = _I
    L.cuembed_get_forward_row_load_policy.argtypes = []
    L.cuembed_set_forward_wide_load.restype = None
    L.cuembed_set_forward_wide_load.argtypes = [_I]
    L.cuembed_embedding_backward.restype = None
    L.cuembed_embedding_backward.argtypes = [_VP, _I, _I, _I, _I, _VP, _VP, _VP, _I, _VP, _I, _VP, _VP, _VP]
    L.cuembed_embedding_backward_reference_sums.restype = None
    L.cuembed_embedding_backward_reference_sums.argtypes = [_VP, _I, _I, _I, _I, _VP, _VP, _VP, _I, _VP, _I, _VP, _VP, _VP]
    L.cuembed_set_backward_tuning.restype = None
    L.cuembed_set_backward_tuning.argtypes = [_I, _I]
    L.cuembed_get_backward_tuning.restype = None
    L.cuembed_get_backward_tuning.argtypes = [ctypes.POINTER(_I)]
    L.cuembed_transpose.restype = None
    L.cuembed_transpose.argtypes = [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP,
                                    ctypes.POINTER(ctypes.c_size_t), _VP]
    L.cuembed_transpose_bounded.restype = None
    L.cuembed_transpose_bounded.argtypes = [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP,
                                            ctypes.POINTER(ctypes.c_size_t), _I, _VP]
    L.cuembed_transpose_hinted.restype = None
    L.cuembed_transpose_hinted.argtypes = [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP,
                                           ctypes.POINTER(ctypes.c_size_t), _I, _I, _VP]
    L.cuembed_transpose_fixed_hotness.restype = None
    L.cuembed_transpose_fixed_hotness.argtypes = [_VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _VP,
                                                  ctypes.POINTER(ctypes.c_size_t), _I, _VP]
    L.cuembed_transpose_sample_blocks.restype = None
    L.cuembed_transpose_sample_blocks.argtypes = [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP,
                                                  ctypes.POINTER(ctypes.c_size_t), _I, _I, _I, _VP]
    L.cuembed_transpose_fixed_hotness_sample_blocks.restype = None
    L.cuembed_transpose_fixed_hotness_sample_blocks.argtypes = [_VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _VP,
                                                                ctypes.POINTER(ctypes.c_size_t), _I, _I, _VP]
    L.cuembed_transpose_remapped.restype = None
    L.cuembed_transpose_remapped.argtypes = [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP,
                                             ctypes.POINTER(ctypes.c_size_t), _I, _I, _I, _VP]
    L.cuembed_transpose_fixed_hotness_remapped.restype = None
    L.cuembed_transpose_fixed_hotness_remapped.argtypes = [_VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP,
                                                           ctypes.POINTER(ctypes.c_size_t), _I, _I, _VP]
    L.cuembed_transpose_sample_block_length.restype = ctypes.c_int64
    L.cuembed_transpose_sample_block_length.argtypes = [ctypes.c_int64, _I]
    L.cuembed_recommended_sample_blocks.restype = _I
    L.cuembed_recommended_sample_blocks.argtypes = [_I, _I, _I, ctypes.c_int64]
    L.cuembed_translate_indices_for_row_cache.restype = None
    L.cuembed_translate_indices_for_row_cache.argtypes = [_VP, _I, ctypes.c_int64, _VP, ctypes.c_int64, ctypes.c_int64,
                                                          _VP, _VP]
    L.cuembed_compute_compressed_grad_indices.restype = None
    L.cuembed_compute_compressed_grad_indices.argtypes = [_VP, _I, _I, _VP, _VP,
                                                          ctypes.POINTER(ctypes.c_size_t), _VP]
    L.cuembed_compute_compressed_grad_indices_blocked.restype = None
    L.cuembed_compute_compressed_grad_indices_blocked.argtypes = [_VP, _I, _I, _I, _VP, _VP, _VP, _VP,
                                                                  ctypes.POINTER(ctypes.c_size_t), _VP]
    L.cuembed_embedding_backward_blocked.restype = None
    L.cuembed_embedding_backward_blocked.argtypes = [_VP, _I, _I, _I, _I, _VP, _VP, _VP, _I, _VP, _I, _VP, _VP, _I, _VP,
                                                     _VP]
    L.cuembed_embedding_backward_bounded.restype = None
    L.cuembed_embedding_backward_bounded.argtypes = [_VP, _I, _I, _I, _I, _VP, _VP, _VP, _I, _VP, _I, _VP, _VP, _I, _VP,
                                                     _I, _VP, _I, _VP]
    L.cuembed_extract_row_ids_from_fixed.restype = None
    L.cuembed_extract_row_ids_from_fixed.argtypes = [_I, _I, _I, _VP, _VP]
    L.cuembed_extract_row_ids_from_csr.restype = None
    L.cuembed_extract_row_ids_from_csr.argtypes = [_VP, _I, _I, _I, _VP, _VP]
    L.cuembed_extract_row_ids_for_concat.restype = None
    L.cuembed_extract_row_ids_for_concat.argtypes = [_I, _I, _VP, _VP]
    L.cuembed_forward_launch_shape.restype = None
    L.cuembed_forward_launch_shape.argtypes = [_I, _I, _I, _I, _I, _I, _I, _I, ctypes.POINTER(_I)]
    L.cuembed_embedding_weight_grad.restype = None
    L.cuembed_embedding_weight_grad.argtypes = [_VP, _I, _I, _VP, _I, _VP, _I, _VP, _I, _I, _VP, _VP]
    L.cuembed_set_forward_reduction_order.restype = None
    L.cuembed_set_forward_reduction_order.argtypes = [_I]
    L.cuembed_get_forward_reduction_order.restype = _I
    L.cuembed_get_forward_reduction_order.argtypes = []
    L.cuembed_device_shape.restype = None
    L.cuembed_device_shape.argtypes = [ctypes.POINTER(_I)]
    L.cuembed_backward_launch_shape.restype = None
    L.cuembed_backward_launch_shape.argtypes = [_I, _I, _I, ctypes.c_int64, _I, _I, _I, ctypes.POINTER(_I)]
    L.cuembed_recommended_sample_blocks_on.restype = _I
    L.cuembed_recommended_sample_blocks_on.argtypes = [_I, _I, _I, ctypes.c_int64, _I, _I, ctypes.c_int64]
    L.cuembed_peek_last_error.restype = _I
    L.cuembed_peek_last_error.argtypes = []
    L.cuembed_version.restype = ctypes.c_char_p
    L.cuembed_version.argtypes = []


def lib():
    """The loaded shared library; raises CuembedLibraryError when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CuembedLibraryError(
                "cuembed_amd: %s not found. Build it with `python -m cuembed_amd.build` "
                "(needs hipcc); there is no CPU or PyTorch fallback." % LIB_PATH)
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # missing ROCm runtime etc.
            raise CuembedLibraryError("cuembed_amd: cannot load %s: %s" % (LIB_PATH, e))
        _declare(L)
        _lib = L
    return _lib


def version():
    return lib().cuembed_version().decode()
