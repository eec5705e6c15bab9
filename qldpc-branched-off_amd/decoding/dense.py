"""Dense-matrix entry points of the reference (src/decoding/dense.py): same names, arguments and return tuples.

The reference's dense min-sum path is bit-identical to its sparse path, so both entry points run the
same GPU decoder on the CSR structure of ``H != 0``; only ``alpha_estimation`` needs the single check pass.
"""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import check, f64, i8, lib, ptr
from .kernels import minsum_core


def performMinSum_Symmetric(H, syndrome, initialBelief, maxIter=50, alpha=1.0, alpha_mode="dynamical", damping=1.0, clip_llr=20.0,
                            alpha_estimation=False):
    """reference dense.py:5-73."""
    mode, aval, seq = _lib.alpha_args(alpha_mode, alpha)
    H = np.asarray(H, dtype=np.float64)
    syndrome = np.asarray(syndrome, dtype=np.int8)
    initialBelief = np.asarray(initialBelief, dtype=np.float64)
    m, n = H.shape
    if alpha_estimation:                                   # dense.py:54-56: unscaled iteration-0 messages
        if maxIter < 1:
            return np.zeros(n, dtype=np.int8), False, None, -1
        current_alpha = {_lib.ALPHA_DYNAMIC: 0.5, _lib.ALPHA_CONST: aval}.get(mode, float(seq[0]))
        mask = H != 0
        ssign = (1 - 2 * syndrome).astype(np.float64).reshape(-1, 1)
        R = minsum_core(H, np.where(mask, initialBelief, 0.0), ssign, mask, current_alpha)
        scale = current_alpha if current_alpha != 0 else 1.0
        return np.zeros(n, dtype=np.int8), False, R / scale, 0
    indptr, indices, _ = _lib.canonical_csr(H)
    g = _lib.graph_for(indptr, indices, n)
    err, conv, llr, it = _lib.minsum_decode_batch(g, syndrome.reshape(1, -1), initialBelief, maxIter, alpha_mode, alpha, damping, clip_llr)
    return err[0], bool(conv[0]), llr[0], int(it[0])


def performBeliefPropagationFast(H, syndrome, initialBelief, maxIter=50):
    """reference dense.py:75-96 (sum-product, no clip/damping)."""
    H = np.asarray(H, dtype=np.float64)
    syndrome = i8(syndrome).reshape(1, -1)
    prior = f64(initialBelief)
    m, n = H.shape
    indptr, indices, _ = _lib.canonical_csr(H)
    g = _lib.graph_for(indptr, indices, n)
    err, llr = np.zeros((1, n), np.int8), np.zeros((1, n))
    conv, it = np.zeros(1, np.uint8), np.zeros(1, np.int32)
    check(lib().qldpc_bp_decode_batch(g.handle, C.c_int64(1), ptr(syndrome, C.c_int8), ptr(prior, C.c_double), C.c_int(int(maxIter)),
                                      ptr(err, C.c_int8), ptr(llr, C.c_double), ptr(conv, C.c_uint8), ptr(it, C.c_int32)))
    return err[0], bool(conv[0]), llr[0], int(it[0])
