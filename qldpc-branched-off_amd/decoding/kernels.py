"""Kernel-level entry points with the names and signatures of the reference's ``src/decoding/kernels.py``.

Every function forwards to a HIP kernel through the C ABI (include/qldpc_hip.h); nothing is computed in Python.
In-place semantics of the reference (gf2_elimination* mutate A and b) are preserved.
"""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import check, f64, i8, i32, lib, ptr, u8


def _graph(H_indices, H_indptr, n):
    return _lib.graph_for(H_indptr, H_indices, n)


def packed_words(n):
    return ((n + 7) // 8 + 7) // 8


def _pack_rows_uint64(A):
    """Rows of a 0/1 matrix as little-endian uint64 words, padded to 8 bytes (reference kernels.py:36-46)."""
    A = np.asarray(A)
    m, n = A.shape
    nw = packed_words(n)
    out = np.zeros((m, nw * 8), dtype=np.uint8)
    bits = np.packbits(A.astype(np.uint8, copy=False), axis=1, bitorder="little")
    out[:, :bits.shape[1]] = bits
    return out.view(np.uint64), n


def gf2_elimination_packed_core(A_packed, b, n):
    """reference kernels.py:48-96.  A_packed uint64[m, nwords] and b are reduced IN PLACE."""
    if A_packed.dtype != np.uint64 or not A_packed.flags.c_contiguous:
        raise ValueError("A_packed must be a C-contiguous uint64 array")
    m, nw = A_packed.shape
    b8 = u8(np.asarray(b) & 1).copy()
    maxp = max(min(m, n), 1)
    pr, pc, npv = np.zeros(maxp, np.int64), np.zeros(maxp, np.int64), np.zeros(1, np.int32)
    check(lib().qldpc_gf2_eliminate_packed(C.c_int64(1), C.c_int(m), C.c_int(n), C.c_int(nw), ptr(A_packed, C.c_uint64),
                                           ptr(b8, C.c_uint8), ptr(pr, C.c_int64), ptr(pc, C.c_int64), ptr(npv, C.c_int32)))
    b[...] = b8.astype(b.dtype)
    k = int(npv[0])
    return A_packed, b, pr[:k].copy(), pc[:k].copy()


def gf2_elimination_packed(A, b):
    """reference kernels.py:98-106: pack with NumPy, eliminate on the GPU; b is reduced in place."""
    A_packed, n = _pack_rows_uint64(A)
    return gf2_elimination_packed_core(A_packed, b, n)


def gf2_elimination(A, b):
    """reference kernels.py:5-34: Gauss-Jordan over GF(2); A (any integer dtype, 0/1) and b are reduced IN PLACE."""
    m, n = A.shape
    A8 = u8(np.asarray(A) & 1).copy()
    b8 = u8(np.asarray(b) & 1).copy()
    maxp = max(min(m, n), 1)
    pr, pc, npv = np.zeros(maxp, np.int64), np.zeros(maxp, np.int64), np.zeros(1, np.int32)
    check(lib().qldpc_gf2_eliminate(C.c_int64(1), C.c_int(m), C.c_int(n), ptr(A8, C.c_uint8), ptr(b8, C.c_uint8), ptr(pr, C.c_int64),
                                    ptr(pc, C.c_int64), ptr(npv, C.c_int32)))
    A[...] = A8.astype(A.dtype)
    b[...] = b8.astype(b.dtype)
    k = int(npv[0])
    return A, b, pr[:k].copy(), pc[:k].copy()


def minsum_core_sparse(H_data, H_indices, H_indptr, Q_flat, syndrome_sign, alpha, m, n):
    """reference kernels.py:138-169 -> (R_flat, R_sum)."""
    g = _graph(H_indices, H_indptr, n)
    Q, ss = f64(Q_flat), f64(syndrome_sign).reshape(-1)
    R, Rs = np.zeros(g.nnz), np.zeros(n)
    check(lib().qldpc_minsum_check_pass(g.handle, C.c_int64(1), ptr(Q, C.c_double), ptr(ss, C.c_double), C.c_double(float(alpha)),
                                        ptr(R, C.c_double), ptr(Rs, C.c_double)))
    return R, Rs


def _dense_pass(fn, Q, syndrome_sign, mask, param):
    mask = np.asarray(mask, dtype=bool)
    m, n = mask.shape
    indptr, indices, _ = _lib.canonical_csr(mask)
    g = _lib.graph_for(indptr, indices, n)
    Qf = f64(np.asarray(Q, dtype=np.float64)[mask])            # row-major boolean gather == CSR edge order
    ss = f64(syndrome_sign).reshape(-1)
    R, Rs = np.zeros(g.nnz), np.zeros(n)
    check(fn(g.handle, C.c_int64(1), ptr(Qf, C.c_double), ptr(ss, C.c_double), C.c_double(float(param)), ptr(R, C.c_double),
             ptr(Rs, C.c_double)))
    out = np.zeros((m, n), dtype=np.float64)
    out[mask] = R
    return out


def minsum_core(H, Q, syndrome_sign, mask, alpha):
    """reference kernels.py:108-136: dense-mask twin of minsum_core_sparse -> R[m, n]."""
    return _dense_pass(lib().qldpc_minsum_check_pass, Q, syndrome_sign, mask, alpha)


def bp_core(H, Q, syndrome_sign, mask, clip_val):
    """reference kernels.py:171-193: tanh-product check update -> R[m, n]."""
    return _dense_pass(lib().qldpc_bp_check_pass, Q, syndrome_sign, mask, clip_val)


def syndrome_check(H_data, H_indices, H_indptr, candidate, m):
    """reference kernels.py:222-231: s = H e over GF(2) -> int8[m]."""
    cand = i8(candidate)
    g = _graph(H_indices, H_indptr, cand.size)
    out = np.zeros(g.m, np.int8)
    check(lib().qldpc_gf2_spmv_batch(g.handle, C.c_int64(1), ptr(cand, C.c_int8), ptr(out, C.c_int8)))
    return out


def _decode_raw(H_indices, H_indptr, syndrome, initialBelief, maxIter, mode, aval, seq, damping, clip_llr):
    prior = f64(initialBelief)
    g = _graph(H_indices, H_indptr, prior.size)
    s = i8(syndrome).reshape(1, -1)
    seq = f64(seq)
    err, llr = np.zeros((1, g.n), np.int8), np.zeros((1, g.n))
    conv, it = np.zeros(1, np.uint8), np.zeros(1, np.int32)
    check(lib().qldpc_minsum_decode_batch(g.handle, C.c_int64(1), ptr(s, C.c_int8), ptr(prior, C.c_double), C.c_int(int(maxIter)),
                                          C.c_int(mode), C.c_double(float(aval)), ptr(seq, C.c_double), C.c_int(seq.size),
                                          C.c_double(float(damping)), C.c_double(float(clip_llr)), C.c_int(0), ptr(err, C.c_int8),
                                          ptr(llr, C.c_double), ptr(conv, C.c_uint8), ptr(it, C.c_int32)))
    return err[0], bool(conv[0]), llr[0], int(it[0])


def minsum_decoder_full(H_indices, H_indptr, syndrome, initialBelief, maxIter, use_dynamic_alpha, alpha_val, damping, clip_llr):
    """reference kernels.py:234-366 -> (candidateError int8[n], converged, values f64[n], final_iter)."""
    mode = _lib.ALPHA_DYNAMIC if use_dynamic_alpha else _lib.ALPHA_CONST
    return _decode_raw(H_indices, H_indptr, syndrome, initialBelief, maxIter, mode, alpha_val, np.zeros(1), damping, clip_llr)


def minsum_decoder_full_autoregressive(H_indices, H_indptr, syndrome, initialBelief, maxIter, alpha_seq, alpha_len, damping, clip_llr):
    """reference kernels.py:369-485."""
    seq = np.asarray(alpha_seq, dtype=np.float64)[:alpha_len]
    return _decode_raw(H_indices, H_indptr, syndrome, initialBelief, maxIter, _lib.ALPHA_SEQ, 0.0, seq, damping, clip_llr)
