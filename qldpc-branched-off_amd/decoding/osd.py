"""``performOSD_enhanced`` (reference src/decoding/osd.py:5-77).  OSD-0 runs on the GPU (qldpc_osd0_batch)."""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import check, f64, i8, i32, lib, ptr


def performOSD_enhanced(H, syndrome, llr, hard, order=0, max_combinations=None, ordering=None):
    """OSD-0 post-processing -> int64[n] solution, as osd.py:5-29.

    ``ordering`` (extension) pins the elimination order; default is ascending |llr| with ties by ascending index
    (the reference's np.argsort default kind leaves the tie order implementation-defined).

    ``order > 0``: the reference returns the OSD-0 solution whenever it reproduces the syndrome (osd.py:27-29) and only
    otherwise enters the combination sweep (osd.py:31-75).  OSD-0 always reproduces a CONSISTENT syndrome (one in the
    column space of H -- every syndrome the Monte-Carlo engine produces), so that case is exact here.  For an
    inconsistent syndrome with order > 0 the sweep is not implemented yet and NotImplementedError is raised.
    """
    indptr, indices, shape = _lib.canonical_csr(H)
    n = shape[1]
    g = _lib.graph_for(indptr, indices, n)
    s = i8(syndrome).reshape(1, -1)
    l = f64(llr).reshape(1, -1)
    h = i8(hard).reshape(1, -1)
    sol = np.zeros((1, n), np.int8)
    op = None
    if ordering is not None:
        ordering = i32(ordering).reshape(1, -1)
        op = ptr(ordering, C.c_int32)
    check(lib().qldpc_osd0_batch(g.handle, C.c_int64(1), ptr(s, C.c_int8), ptr(l, C.c_double), ptr(h, C.c_int8), op, ptr(sol, C.c_int8)))
    if order != 0:
        chk = np.zeros((1, g.m), np.int8)
        check(lib().qldpc_gf2_spmv_batch(g.handle, C.c_int64(1), ptr(sol, C.c_int8), ptr(chk, C.c_int8)))       # osd.py:27
        if not np.array_equal(chk[0], s[0] & 1):
            raise NotImplementedError("OSD-w sweep (order > 0) for a syndrome outside the column space of H is a 'next' row (SURVEY 8f-1)")
    return sol[0].astype(np.int64)
