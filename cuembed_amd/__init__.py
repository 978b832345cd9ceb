"""cuembed_amd -- MI355X-native (gfx950 / CDNA4) embedding lookup.

The gather-reduce hot path of NVIDIA/cuEmbed (EmbeddingForward, EmbeddingBackward,
Transpose and the index helpers) as hand-written HIP kernels behind
 * a header-only C++ host API   (cuembed_amd/csrc/cuembed/include/*.hpp),
 * a C ABI shared library       (include/cuembed_amd.h, cuembed_amd/lib/libcuembed_amd.so),
 * this Python host layer       (cuembed_amd.ops; the torch op surface of the reference's examples/pytorch is
                                 cuembed_amd.cuembed_pyt over cuembed_amd/lib/libcuembed_pyt.so).
"""
from . import _lib  # noqa: F401
from .ops import (CONCAT, MEAN, SUM, ComputeCompressedGradIndices, EmbeddingBackward,  # noqa: F401
                  EmbeddingForward, ExtractRowIdsForConcat, ExtractRowIdsFromCSR,
                  ExtractRowIdsFromFixed, Transpose, compressed_grad_workspace_bytes,
                  compute_compressed_grad_indices, compute_compressed_grad_indices_blocked,
                  compressed_grad_blocked_workspace_bytes, SHARED_ROW_BIT, embedding_backward, embedding_forward,
                  get_backward_tuning, set_backward_tuning, recommended_sample_blocks, transpose_sample_block_length,
                  embedding_weight_grad, bag_order_by_length, capacity_overflowed, decide_row_loads,
                  new_row_loads_decision,
                  extract_row_ids_for_concat, extract_row_ids_from_csr,
                  extract_row_ids_from_fixed, forward_launch_shape, backward_launch_shape, device_shape, get_forward_reduction_order,
                  set_forward_reduction_order, set_forward_row_load_policy, get_forward_row_load_policy, set_forward_wide_load, transpose,
                  transpose_fixed_hotness,
                  transpose_workspace_bytes)

__version__ = "0.1.0"
