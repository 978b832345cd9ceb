"""ctypes front-end of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/cuembed_oracle.cpp for the contract and the parity
status).  Arrays are numpy; fp16 tables are numpy.float16 (bit-identical to
IEEE binary16 storage).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CUEMBED_ORACLE_LIB: load another build of the same source (the ASan/UBSan one, `make -C oracle asan`)
_LIB_PATH = os.environ.get("CUEMBED_ORACLE_LIB") or os.path.join(_HERE, "libcuembed_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libref_datagen.so")

SUM, MEAN, CONCAT = 0, 1, 2
MODES = {"sum": SUM, "mean": MEAN, "concat": CONCAT}


def build(ref=True):
    """Compile the oracle (and, when /root/reference exists, oracle/_ref)."""
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    if ref and os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build(ref=False)
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_fnv1a64.restype = ctypes.c_uint64
        L.oracle_fnv1a64.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        L.oracle_generate_indices.restype = ctypes.c_int64
        L.oracle_h2f.restype = ctypes.c_float
        L.oracle_h2f.argtypes = [ctypes.c_uint16]
        L.oracle_f2h.restype = ctypes.c_uint16
        L.oracle_f2h.argtypes = [ctypes.c_float]
        _lib = L
    return _lib


def ref_lib():
    """The reference's own datagen (oracle/_ref); None when not built."""
    global _ref
    if _ref is None and os.path.exists(_REF_PATH):
        _ref = ctypes.CDLL(_REF_PATH)
    return _ref


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _etype(a):
    if a.dtype == np.float32:
        return 0
    if a.dtype == np.float16:
        return 1
    if a.dtype == np.uint16:      # bfloat16 bit patterns (numpy has no bf16 dtype)
        return 2
    raise TypeError("element type must be float32, float16 or uint16 (bf16 bits), got %s" % a.dtype)


def to_bf16_bits(a):
    """float32 array -> bfloat16 bit patterns (uint16), round-to-nearest-even."""
    x = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((x + 0x7fff + ((x >> 16) & 1)) >> 16).astype(np.uint16)
    return r.reshape(np.shape(a))


def from_bf16_bits(b):
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(np.shape(b))


def _itype(a):
    if a.dtype == np.int32:
        return 0
    if a.dtype == np.int64:
        return 1
    raise TypeError("index type must be int32 or int64, got %s" % a.dtype)


def _c(a):
    return None if a is None else np.ascontiguousarray(a)


def fnv1a64(a):
    a = np.ascontiguousarray(a)
    return int(lib().oracle_fnv1a64(_p(a), ctypes.c_uint64(a.nbytes)))


def embedding_forward(params, indices, offsets=None, weights=None, batch_size=None,
                      num_hots=0, mode="sum", fp16_math=False, threads=1):
    """EmbeddingForwardCpu restated (embedding_lookup_cpu.hpp:35-94)."""
    params, indices, offsets, weights = _c(params), _c(indices), _c(offsets), _c(weights)
    m = MODES[mode] if isinstance(mode, str) else int(mode)
    W = params.shape[-1]
    if offsets is not None:
        B = offsets.shape[0] - 1 if batch_size is None else batch_size
        nnz = int(offsets[B])
    else:
        B = indices.size // num_hots if batch_size is None else batch_size
        nnz = B * num_hots
    out_rows = nnz if m == CONCAT else B
    ret = np.zeros((out_rows, W), dtype=params.dtype)
    if weights is not None and weights.dtype != params.dtype:
        raise TypeError("weights dtype must equal the table dtype")
    rc = lib().oracle_embedding_forward(
        _p(params), _etype(params), ctypes.c_int(W), ctypes.c_int(B), ctypes.c_int(num_hots),
        _p(indices), _itype(indices), _p(offsets),
        0 if offsets is None else _itype(offsets), _p(weights), _p(ret),
        ctypes.c_int(m), ctypes.c_int(1 if fp16_math else 0), ctypes.c_int(threads))
    if rc != 0:
        raise ValueError("contract violation (reference CHECK would fail)")
    return ret


def embedding_backward(grad_y, embed_width, num_grad_rows, t_indices, t_sample_ids,
                       t_remapped=None, t_weights=None, skip_grad_init=False,
                       grad_embedding=None):
    """EmbeddingBackwardCpu restated (embedding_lookup_cpu.hpp:96-144).

    Returns (grad_embedding, inverse_mapping or None)."""
    grad_y, t_indices, t_sample_ids = _c(grad_y), _c(t_indices), _c(t_sample_ids)
    t_remapped, t_weights = _c(t_remapped), _c(t_weights)
    nnz = t_indices.shape[0]
    if grad_embedding is None:
        grad_embedding = np.zeros((num_grad_rows, embed_width), dtype=grad_y.dtype)
    inv = None
    if t_remapped is not None:
        inv = np.zeros((num_grad_rows,), dtype=t_indices.dtype)
    rc = lib().oracle_embedding_backward(
        _p(grad_y), _etype(grad_y), ctypes.c_int(embed_width),
        ctypes.c_int64(num_grad_rows), ctypes.c_int64(nnz), _p(t_indices),
        _p(t_sample_ids), _p(t_remapped), _itype(t_indices), _p(t_weights),
        ctypes.c_int(1 if skip_grad_init else 0), _p(grad_embedding), _p(inv))
    assert rc == 0
    return grad_embedding, inv


def transpose(rows, cols, weights=None, stable=True):
    """Transpose: sort (sample id[, weight]) by lookup index.

    stable=True  -> device contract (index_transforms.cuh:95-137, stable radix)
    stable=False -> CPU reference order (index_transforms_cpu.hpp:86-125)."""
    rows, cols, weights = _c(rows), _c(cols), _c(weights)
    nnz = rows.shape[0]
    t_rows = np.empty_like(cols)
    t_cols = np.empty_like(rows)
    t_w = None if weights is None else np.empty_like(weights)
    rc = lib().oracle_transpose(
        _p(rows), _p(cols), _p(weights), ctypes.c_int64(nnz), _itype(rows),
        0 if weights is None else _etype(weights), _p(t_rows), _p(t_cols), _p(t_w),
        ctypes.c_int(1 if stable else 0))
    assert rc == 0
    return t_rows, t_cols, t_w


def extract_row_ids_from_fixed(batch_size, num_hots, dtype=np.int32):
    out = np.empty((batch_size * num_hots,), dtype=dtype)
    lib().oracle_extract_row_ids_from_fixed(ctypes.c_int(batch_size), ctypes.c_int(num_hots),
                                            _itype(out), _p(out))
    return out


def extract_row_ids_from_csr(offsets, dtype=np.int32):
    offsets = _c(offsets)
    B = offsets.shape[0] - 1
    out = np.empty((int(offsets[B] - offsets[0]),), dtype=dtype)
    lib().oracle_extract_row_ids_from_csr(_p(offsets), _itype(offsets), ctypes.c_int(B),
                                          _itype(out), _p(out))
    return out


def extract_row_ids_for_concat(nnz, dtype=np.int32):
    out = np.empty((nnz,), dtype=dtype)
    lib().oracle_extract_row_ids_for_concat(ctypes.c_int64(nnz), _itype(out), _p(out))
    return out


def compute_compressed_grad_indices(indices):
    indices = _c(indices)
    out = np.empty_like(indices)
    lib().oracle_compute_compressed_grad_indices(_p(indices), ctypes.c_int64(indices.shape[0]),
                                                 _itype(indices), _p(out))
    return out


def allocate_forward(num_categories, embed_width, batch_size, hotness, alpha=0.0,
                     is_csr=False, elem=np.float32, index=np.int32, shuffle=True,
                     permute=True):
    """AllocateForward's RNG recipe (embedding_allocation.cu:96-169).

    Returns dict(table, offsets, indices, weights)."""
    table = np.empty((num_categories, embed_width), dtype=elem)
    offsets = np.empty((batch_size + 1,), dtype=np.int32)
    indices = np.empty((batch_size * hotness,), dtype=index)
    weights = np.empty((batch_size * hotness,), dtype=elem)
    nnz = ctypes.c_int64(0)
    rc = lib().oracle_allocate_forward(
        ctypes.c_int64(num_categories), ctypes.c_int(embed_width), ctypes.c_int(batch_size),
        ctypes.c_int(hotness), ctypes.c_double(alpha), ctypes.c_int(1 if is_csr else 0),
        ctypes.c_int(1 if shuffle else 0), ctypes.c_int(1 if permute else 0),
        _etype(table), _itype(indices), _p(table), _p(offsets), _p(indices), _p(weights),
        ctypes.byref(nnz))
    assert rc == 0
    n = nnz.value
    return dict(table=table, offsets=offsets, indices=indices[:n].copy(),
                weights=weights[:n].copy())


def generate_indices(num_categories, batch_size, hotness, alpha=0.0, index=np.int32,
                     shuffle=True, permute=True, offsets=None):
    cap = batch_size * hotness
    out = np.empty((cap,), dtype=index)
    off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.int32)
    n = lib().oracle_generate_indices(
        ctypes.c_int64(num_categories), ctypes.c_int(batch_size), ctypes.c_int(hotness),
        ctypes.c_double(alpha), ctypes.c_int(1 if shuffle else 0),
        ctypes.c_int(1 if permute else 0), _itype(out), _p(off), _p(out))
    return out[:n].copy()


def psx_samples(num_categories_arg, hot, alpha, n_samples, index=np.int32, shuffle=True,
                permute=True, use_reference=False):
    """n_samples consecutive getCategoryIndices() of one generator.

    use_reference=True runs the REFERENCE's datagen.cpp (oracle/_ref)."""
    out = np.empty((n_samples, hot), dtype=index)
    L = ref_lib() if use_reference else lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libref_datagen.so is not built")
    fn = L.ref_psx_samples if use_reference else L.oracle_psx_samples
    fn(ctypes.c_int64(num_categories_arg), ctypes.c_int(hot), ctypes.c_double(alpha),
       ctypes.c_int(1 if shuffle else 0), ctypes.c_int(1 if permute else 0),
       ctypes.c_int(n_samples), _itype(out), _p(out))
    return out


def allocate_grad_y(count, elem=np.float32):
    out = np.empty((count,), dtype=elem)
    lib().oracle_allocate_grad_y(ctypes.c_int64(count), _etype(out), _p(out))
    return out


def max_threads():
    return int(lib().oracle_max_threads())
