"""``performOSD_enhanced`` (reference src/decoding/osd.py:5-77) on the GPU: qldpc_osd0_batch / qldpc_osdw_batch."""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import check, f64, i8, i32, lib, ptr


def performOSD_enhanced(H, syndrome, llr, hard, order=0, max_combinations=None, ordering=None, flags=0):
    """Ordered-statistics post-processing -> int64[n] solution.

    order == 0: OSD-0 (osd.py:5-29).  order > 0: the reference returns the OSD-0 solution whenever it reproduces the syndrome
    (osd.py:27-29) -- always the case for a syndrome in the column space of H, i.e. every syndrome the Monte-Carlo engine
    produces -- and otherwise scores the flip sets of weight <= order over the order+10 least reliable non-pivot positions
    (osd.py:31-75); both branches run on the device.

    The elimination order is the reference's own expression, ``np.argsort(np.abs(llr))`` (osd.py:11-12), evaluated on the host with the
    caller's NumPy: its default sort kind leaves the order of equal keys implementation-defined, so taking it from the same NumPy call is
    what makes this single-shot wrapper a drop-in (the batched device entry points, qldpc_osd0_batch without an ordering, use the stable
    rule: ascending |llr|, ties by ascending index).  ``ordering`` (extension) pins another order; ``flags`` (extension, order == 0 only)
    selects an OSD-0 kernel variant (QLDPC_FLAG_OSD_*; identical results).
    """
    if order < 0:
        order = 0          # the reference's sweep loops are empty for a negative order: it returns the OSD-0 solution
    indptr, indices, shape = _lib.canonical_csr(H)
    n = shape[1]
    g = _lib.graph_for(indptr, indices, n)
    s = i8(syndrome).reshape(1, -1)
    l = f64(llr).reshape(1, -1)
    h = i8(hard).reshape(1, -1)
    sol = np.zeros((1, n), np.int8)
    if ordering is None:
        ordering = np.argsort(np.abs(l[0]))                        # osd.py:11-12, literally
    ordering = i32(ordering).reshape(1, -1)
    op = ptr(ordering, C.c_int32)
    if order == 0:
        check(lib().qldpc_osd0_batch(g.handle, C.c_int64(1), ptr(s, C.c_int8), ptr(l, C.c_double), ptr(h, C.c_int8), op, int(flags), ptr(sol, C.c_int8)))
    else:
        check(lib().qldpc_osdw_batch(g.handle, C.c_int64(1), ptr(s, C.c_int8), ptr(l, C.c_double), ptr(h, C.c_int8), op, C.c_int(int(order)),
                                     C.c_int64(int(max_combinations or 0)), ptr(sol, C.c_int8)))
    return sol[0].astype(np.int64)
