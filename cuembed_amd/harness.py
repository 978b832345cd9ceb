"""Synthetic workloads of the benchmark harness (host side).

Python front-end of cuembed_amd/lib/libcuembed_harness.so, the counterpart of the reference's
utils/src/embedding_allocation.cu + datagen.cpp: power-law lookup indices, the AllocateForward /
AllocateBackward RNG recipe.  Returns numpy arrays; callers move them to the GPU."""
import ctypes
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# CUEMBED_HARNESS_LIB: load another build of the same source (the ASan/UBSan build of the host sanitizer target)
_PATH = os.environ.get("CUEMBED_HARNESS_LIB") or os.path.join(_PKG, "lib", "libcuembed_harness.so")
_h = None


def _lib():
    global _h
    if _h is None:
        if not os.path.exists(_PATH):
            raise RuntimeError("cuembed_amd: %s not found; run `python -m cuembed_amd.build`" % _PATH)
        L = ctypes.CDLL(_PATH)
        L.cuembed_harness_generate_indices.restype = ctypes.c_int64
        L.cuembed_harness_allocate_forward.restype = ctypes.c_int64
        L.cuembed_harness_allocate_grad_y.restype = None
        _h = L
    return _h


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _check_alpha(alpha, hotness):
    """alpha = 1 is the one exponent the reference's inverse-CDF recipe cannot do (datagen.cpp:39-50: x = (u * span +
    1) ^ (1 / (1 - alpha)) with span = 0: every draw is id 1, and a sample of more than one DISTINCT id never fills --
    the reference's generator does not return either).  Say so instead of hanging."""
    if float(alpha) == 1.0 and hotness > 1:
        raise ValueError("alpha = 1 is singular in the reference's power-law recipe (every draw is the same id); "
                         "use e.g. 0.99 or 1.01")


def generate_indices(num_categories, batch_size, hotness, alpha=0.0, index=np.int32, shuffle=True,
                     permute=True, offsets=None):
    """batch_size samples x hotness distinct ids in [0, num_categories), power-law(alpha)."""
    _check_alpha(alpha, hotness)
    out = np.empty((batch_size * hotness,), dtype=index)
    off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.int32)
    n = _lib().cuembed_harness_generate_indices(
        ctypes.c_int64(num_categories), ctypes.c_int(batch_size), ctypes.c_int(hotness),
        ctypes.c_double(alpha), ctypes.c_int(int(shuffle)), ctypes.c_int(int(permute)),
        ctypes.c_int(1 if np.dtype(index) == np.int64 else 0), _p(off), _p(out))
    return out[:n]


def allocate_forward(num_categories, embed_width, batch_size, hotness, alpha=0.0, is_csr=False,
                     elem=np.float32, index=np.int32, shuffle=True, permute=True, with_table=True,
                     consume_table_draws=True):
    """The reference's forward workload (embedding_allocation.cu:96-169).  with_table=False skips
    the (slow, host-side) table fill but still consumes its draws so that offsets and weights
    are the reference's; use it when the table is filled on the GPU.  consume_table_draws=False
    additionally skips those draws (fast for 10M-row tables; offsets/weights then differ from the
    reference stream but have the same distribution)."""
    _check_alpha(alpha, hotness)
    table = np.empty((num_categories, embed_width), dtype=elem) if with_table else None
    offsets = np.empty((batch_size + 1,), dtype=np.int32)
    indices = np.empty((batch_size * hotness,), dtype=index)
    weights = np.empty((batch_size * hotness,), dtype=elem)
    n = _lib().cuembed_harness_allocate_forward(
        ctypes.c_int64(num_categories), ctypes.c_int(embed_width), ctypes.c_int(batch_size),
        ctypes.c_int(hotness), ctypes.c_double(alpha), ctypes.c_int(int(is_csr)),
        ctypes.c_int(int(shuffle)), ctypes.c_int(int(permute)),
        ctypes.c_int(1 if np.dtype(elem) == np.float16 else 0),
        ctypes.c_int(1 if np.dtype(index) == np.int64 else 0), _p(table), ctypes.c_int(1 if consume_table_draws else 0),
        _p(offsets), _p(indices), _p(weights))
    return dict(table=table, offsets=offsets, indices=indices[:n], weights=weights[:n])


def allocate_grad_y(count, elem=np.float32):
    out = np.empty((count,), dtype=elem)
    _lib().cuembed_harness_allocate_grad_y(ctypes.c_int64(count),
                                           ctypes.c_int(1 if np.dtype(elem) == np.float16 else 0), _p(out))
    return out
