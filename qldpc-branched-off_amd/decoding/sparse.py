"""``performMinSum_Symmetric_Sparse`` with the reference's signature (src/decoding/sparse.py:5-54), plus a batched form."""
import numpy as np

from .. import _lib


def performMinSum_Symmetric_Sparse(H_csr, syndrome, initialBelief, maxIter=100, alpha=1.0, alpha_mode="dynamical", damping=1.0,
                                   clip_llr=20.0):
    """Normalised min-sum decode of one syndrome on the GPU.

    Returns ``(candidateError int8[n], converged bool, values float64[n], final_iter int)`` exactly as the reference
    (final_iter = maxIter-1 when not converged).  Raises ValueError for the same argument errors (sparse.py:24,29,39).
    """
    _lib.alpha_args(alpha_mode, alpha)                      # argument errors first, as in the reference
    syndrome = np.asarray(syndrome, dtype=np.int8)
    initialBelief = np.asarray(initialBelief, dtype=np.float64)
    indptr, indices, shape = _lib.canonical_csr(H_csr)
    g = _lib.graph_for(indptr, indices, shape[1])
    err, conv, llr, it = _lib.minsum_decode_batch(g, syndrome.reshape(1, -1), initialBelief, maxIter, alpha_mode, alpha, damping, clip_llr)
    return err[0], bool(conv[0]), llr[0], int(it[0])


def performMinSum_Symmetric_Sparse_batch(H_csr, syndromes, initialBelief, maxIter=100, alpha=1.0, alpha_mode="dynamical",
                                         damping=1.0, clip_llr=20.0, flags=0):
    """Batched form (not in the reference): syndromes[B, m] -> (errors int8[B,n], converged bool[B], values f64[B,n], final_iter int32[B])."""
    _lib.alpha_args(alpha_mode, alpha)
    indptr, indices, shape = _lib.canonical_csr(H_csr)
    g = _lib.graph_for(indptr, indices, shape[1])
    err, conv, llr, it = _lib.minsum_decode_batch(g, np.asarray(syndromes, dtype=np.int8), initialBelief, maxIter, alpha_mode, alpha,
                                                  damping, clip_llr, flags)
    return err, conv.astype(bool), llr, it
