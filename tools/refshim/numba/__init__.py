"""Container-only stand-in so the reference's *Python source* runs under CPython.

numba is not installed in this image (and cannot be).  The reference decorates its
kernels with ``@njit``; numba's contract for ``@njit`` is "same result as the Python
function" (modulo fastmath), so an identity decorator executes the reference's own
source unchanged with strict IEEE-754 semantics.  Used ONLY by
tests/golden/make_golden.py inside the build container; never shipped to the GPU box
as part of any product or test path.
"""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


jit = njit
prange = range
