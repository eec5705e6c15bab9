"""ctypes binding of csrc/libqldpc_hip.so (C ABI: include/qldpc_hip.h).  Fails loudly; no CPU fallback."""
import ctypes as C
import hashlib
import os
import threading
from collections import OrderedDict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# the product library.  select_build() switches THIS module to another build of the same ABI before the library is first used:
#   "experiments"  libqldpc_hip_experiments.so: plus the measured-and-rejected kernels (the parity tests load it through tests/conftest.py)
#   "timers"       libqldpc_hip_timers.so: clock reads inside the kernels (tools/ diagnostics; make -C csrc timers)
SO_PATH = os.path.join(_HERE, "csrc", "libqldpc_hip.so")
BUILD = "product"


def select_build(name):
    """Point this module at another build of the library ("product", "experiments", "timers").  Only before the first call into it."""
    global SO_PATH, BUILD
    if _lib is not None and name != BUILD:
        raise QldpcError(f"library already loaded ({BUILD}); select_build must come first")
    fn = {"product": "libqldpc_hip.so", "experiments": "libqldpc_hip_experiments.so", "timers": "libqldpc_hip_timers.so"}.get(name, name)   # or a file name in csrc/ (A/B builds of tools/)
    SO_PATH, BUILD = os.path.join(_HERE, "csrc", fn), name

ALPHA_CONST, ALPHA_DYNAMIC, ALPHA_SEQ = 0, 1, 2
FLAG_FIXED_ITERS, FLAG_KERNEL_STREAM, FLAG_KERNEL_RESIDENT, FLAG_KERNEL_GENERIC, FLAG_MC_UNFUSED = 0x1, 0x10, 0x20, 0x40, 0x80
TALLY_SLOTS = 16
TALLY = {"trials": 0, "z_err": 1, "x_err": 2, "total_err": 3, "bp_conv_z": 4, "bp_conv_x": 5, "osd_z": 6, "osd_x": 7,
         "iters_z": 8, "iters_x": 9, "zero_synd_z": 10, "zero_synd_x": 11, "unsat_z": 12, "unsat_x": 13}

FLAG_WG_EDGE_LANES, FLAG_OSD_LDS, FLAG_WG_IDXLOAD = 0x2, 0x20000, 0x40000
FLAG_OSD_REFORDER, FLAG_OSD_QUEUE, FLAG_WG_TABLES = 0x80000, 0x100000, 0x200000
FLAG_WG_VGLOBAL, FLAG_WG_GENERIC, FLAG_OSD_UG, FLAG_OSD_GLOBAL, FLAG_CLOCK_PROBE, FLAG_WG_ROWMAJOR = 0x100, 0x200, 0x400, 0x800, 0x4000, 0x8000
CIRCUIT_PHASES = ("sample", "bp_z", "osd_z", "bp_x", "osd_x", "judge")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "qldpc_hip.h")


class QldpcError(RuntimeError):
    pass


class CircuitDesc(C.Structure):
    """qldpc_circuit_desc (include/qldpc_hip.h)."""
    _fields_ = [("base_len", C.c_int64), ("suffix_len", C.c_int64),
                ("base_ops", C.POINTER(C.c_int32)), ("base_q1", C.POINTER(C.c_int32)), ("base_q2", C.POINTER(C.c_int32)),
                ("suffix_ops", C.POINTER(C.c_int32)), ("suffix_q1", C.POINTER(C.c_int32)), ("suffix_q2", C.POINTER(C.c_int32)),
                ("total_qubits", C.c_int32), ("num_x_checks", C.c_int32), ("num_z_checks", C.c_int32), ("n_data", C.c_int32),
                ("k", C.c_int32), ("reserved", C.c_int32),
                ("x_syn_positions", C.POINTER(C.c_int32)), ("x_syn_ptrs", C.POINTER(C.c_int32)),
                ("z_syn_positions", C.POINTER(C.c_int32)), ("z_syn_ptrs", C.POINTER(C.c_int32)),
                ("data_qubit_indices", C.POINTER(C.c_int32)), ("Lx", C.POINTER(C.c_uint8)), ("Lz", C.POINTER(C.c_uint8))]


_SCALARS = {"int": C.c_int, "int32_t": C.c_int32, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "uint32_t": C.c_uint32, "double": C.c_double,
            "int8_t": C.c_int8, "uint8_t": C.c_uint8, "uint16_t": C.c_uint16}


def _ctype_of(decl):
    """One C parameter declaration of the header -> ctypes type (opaque handles and void* -> c_void_p)."""
    decl = decl.replace("const", " ").strip()
    stars = decl.count("*") + decl.count("[")
    pname = decl.replace("*", " ").split("[")[0].split()[-1]
    if stars == 1 and pname.startswith("d_"):
        return C.c_void_p              # device pointer (hipMalloc / torch data_ptr()): passed as an address
    base = decl.replace("*", " ").split("[")[0].split()
    base = base[0] if len(base) == 1 else (base[0] if base[0] in _SCALARS or base[0].startswith("qldpc_") or base[0] in ("void", "char") else base[-2])
    if stars == 0:
        return _SCALARS[base]
    if base == "char" and stars == 1:
        return C.c_char_p              # NUL-terminated name
    if base == "qldpc_circuit_desc":
        t = CircuitDesc
    elif base in ("void", "char") or base.startswith("qldpc_"):
        t = None                       # opaque: void*
    else:
        t = _SCALARS[base]
    if t is None:
        return C.c_void_p if stars == 1 else C.POINTER(C.c_void_p)
    for _ in range(stars):
        t = C.POINTER(t)
    return t


def parse_header(path=HEADER_PATH):
    """{name: (restype, [argtypes])} for every function include/qldpc_hip.h declares: the binding is derived from the C ABI itself,
    so a mistyped or missing argument raises ctypes.ArgumentError instead of corrupting memory."""
    import re
    with open(path) as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = {}
    for ret, name, args in re.findall(r"\b(const\s+char\s*\*|int|void)\s*(qldpc_\w+)\s*\(([^;{}]*?)\)\s*;", text):
        ret = ret.replace(" ", "")
        restype = C.c_char_p if ret.startswith("constchar") else (C.c_int if ret == "int" else None)
        args = args.strip()
        argtypes = [] if args in ("", "void") else [_ctype_of(a) for a in args.split(",")]
        out[name] = (restype, argtypes)
    return out


_SIGNATURES = None


def signatures():
    global _SIGNATURES
    if _SIGNATURES is None:
        _SIGNATURES = parse_header()
    return _SIGNATURES


def exports():
    """Names of every function the C ABI declares."""
    return sorted(signatures())


_lib = None
_lock = threading.Lock()


def lib():
    """The loaded C-ABI library.  Raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(SO_PATH):
                    raise QldpcError(f"HIP extension missing: {SO_PATH} (build it with `make -C {os.path.dirname(SO_PATH)}` "
                                     "or __graft_entry__.build()); there is no CPU fallback")
                # Code-capacity plans keep up to eight small pieces in flight on streams of their own; the HIP runtime multiplexes streams onto
                # GPU_MAX_HW_QUEUES hardware queues (default 4), which is what bounds them: [[72,12,6]] at batch 4096 runs 1.0e8 shots/s on 4 queues,
                # 1.4e8 on 16 (profiles/r04f_other_configs.txt).  Read when the HIP runtime initialises, so it only takes effect if nothing in this process
                # has touched the GPU yet; a value the caller exported wins.
                os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
                L = C.CDLL(SO_PATH)
                for name, (restype, argtypes) in signatures().items():
                    fn = getattr(L, name)          # AttributeError here = the library lacks a declared symbol
                    fn.restype = restype
                    fn.argtypes = argtypes
                _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().qldpc_last_error()
        raise QldpcError(f"libqldpc_hip error {rc}: {msg.decode() if msg else '?'}")


def device_count():
    return int(lib().qldpc_device_count())


def set_option(name, value):
    """Process-wide kernel-selection switch (include/qldpc_hip.h: qldpc_set_option); results never depend on it."""
    check(lib().qldpc_set_option(name.encode(), int(value)))


class Stream:
    """A HIP stream of `device` owned by this object (qldpc_stream_*); `.ptr` goes where the ABI takes a `stream`."""

    def __init__(self, device=0):
        require_device()
        self.device = int(device)
        self._h = C.c_void_p()
        check(lib().qldpc_stream_create(self.device, C.byref(self._h)))

    @property
    def ptr(self):
        return self._h.value or 0

    def synchronize(self):
        check(lib().qldpc_stream_sync(self.device, self._h))

    def close(self):
        if self._h is not None and self._h.value:
            lib().qldpc_stream_destroy(self.device, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def require_device():
    if device_count() <= 0:
        raise QldpcError("no HIP device visible: libqldpc_hip runs on MI355X (gfx950) only and has no CPU fallback")


def ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i8(a):
    return np.ascontiguousarray(a, dtype=np.int8)


def u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def alpha_args(alpha_mode, alpha):
    """Alpha-mode rules of the reference wrappers (src/decoding/sparse.py:18-29,36-39; dense.py:19-33)."""
    if alpha_mode is None:
        mode = ALPHA_DYNAMIC if alpha == 0 else ALPHA_CONST
    elif alpha_mode == "dynamical":
        mode = ALPHA_DYNAMIC
    elif alpha_mode == "alvarado":
        if alpha <= 0:
            raise ValueError("alpha must be > 0 when alpha_mode='alvarado'")
        mode = ALPHA_CONST
    elif alpha_mode == "alvarado-autoregressive":
        mode = ALPHA_SEQ
    else:
        raise ValueError(f"Unsupported alpha_mode: {alpha_mode}")
    if mode == ALPHA_SEQ:
        seq = np.asarray(alpha, dtype=np.float64)
        if seq.ndim != 1 or seq.size == 0:
            raise ValueError("alpha must be a non-empty 1D sequence for alvarado-autoregressive")
        return mode, 0.0, np.ascontiguousarray(seq)
    return mode, float(alpha), np.zeros(1)


class Graph:
    """Owning wrapper of a qldpc_graph handle (Tanner graph of a parity-check matrix in canonical CSR)."""

    def __init__(self, indptr, indices, n, device=0):
        self.indptr, self.indices = i32(indptr), i32(indices)
        self.m, self.n, self.nnz = self.indptr.size - 1, int(n), int(self.indices.size)
        self.device = device
        self._h = C.c_void_p()
        require_device()
        check(lib().qldpc_graph_create(C.c_int(self.m), C.c_int(self.n), ptr(self.indptr, C.c_int32), ptr(self.indices, C.c_int32),
                                       C.c_int(device), C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                lib().qldpc_graph_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


_graph_cache = OrderedDict()
_GRAPH_CACHE_MAX = 16


def canonical_csr(H):
    """scipy sparse / dense array -> (indptr int32, indices int32 sorted per row, shape)."""
    import scipy.sparse as sp
    if sp.issparse(H):
        c = H.tocsr()
        if not c.has_sorted_indices:
            c = c.sorted_indices()
        if (c.data == 0).any():
            c = c.copy()
            c.eliminate_zeros()
        return c.indptr.astype(np.int32), c.indices.astype(np.int32), c.shape
    H = np.asarray(H)
    m, n = H.shape
    rows, cols = np.nonzero(H)
    indptr = np.zeros(m + 1, np.int32)
    np.add.at(indptr, rows + 1, 1)
    return np.cumsum(indptr, dtype=np.int64).astype(np.int32), cols.astype(np.int32), (m, n)


def graph_for(indptr, indices, n, device=0):
    """Cached Graph for a CSR structure (the reference passes the same H on every trial)."""
    indptr, indices = i32(indptr), i32(indices)
    key = (device, int(n), hashlib.blake2b(indptr.tobytes() + indices.tobytes(), digest_size=16).digest())
    g = _graph_cache.get(key)
    if g is None:
        g = Graph(indptr, indices, n, device)
        _graph_cache[key] = g
        while len(_graph_cache) > _GRAPH_CACHE_MAX:
            _graph_cache.popitem(last=False)
    else:
        _graph_cache.move_to_end(key)
    return g


def minsum_decode_batch(graph, syndromes, prior, max_iter, alpha_mode, alpha, damping=1.0, clip_llr=20.0, flags=0, want_llr=True):
    """qldpc_minsum_decode_batch on host arrays -> (err int8[B,n], conv uint8[B], llr f64[B,n], iters int32[B]).
    want_llr=False leaves the posteriors on the device (llr is returned as None): 9x fewer result bytes over PCIe."""
    mode, aval, seq = alpha_args(alpha_mode, alpha)
    syndromes = i8(syndromes).reshape(-1, graph.m) if graph.m else np.zeros((np.asarray(syndromes).shape[0], 0), np.int8)
    B = syndromes.shape[0]
    prior = f64(prior)
    if prior.size != graph.n:
        raise ValueError(f"initialBelief has {prior.size} entries, H has {graph.n} columns")
    err = np.zeros((B, graph.n), np.int8)
    llr = np.zeros((B, graph.n), np.float64) if want_llr else None
    conv = np.zeros(B, np.uint8)
    iters = np.zeros(B, np.int32)
    check(lib().qldpc_minsum_decode_batch(graph.handle, C.c_int64(B), ptr(syndromes, C.c_int8), ptr(prior, C.c_double),
                                          C.c_int(int(max_iter)), C.c_int(mode), C.c_double(aval), ptr(seq, C.c_double),
                                          C.c_int(seq.size), C.c_double(float(damping)), C.c_double(float(clip_llr)), C.c_int(flags),
                                          ptr(err, C.c_int8), ptr(llr, C.c_double) if want_llr else None, ptr(conv, C.c_uint8),
                                          ptr(iters, C.c_int32)))
    return err, conv, llr, iters


def osd0_batch(graph, syndromes, llr, hard, ordering=None, flags=0):
    """qldpc_osd0_batch on host arrays: OSD-0 solutions int8[B, n] (performOSD_enhanced with order = 0, osd.py:5-29)."""
    syndromes = i8(syndromes).reshape(-1, graph.m) if graph.m else np.zeros((np.asarray(hard).reshape(-1, graph.n).shape[0], 0), np.int8)
    B = syndromes.shape[0]
    llr, hard = f64(llr).reshape(B, graph.n), i8(hard).reshape(B, graph.n)
    sol = np.zeros((B, graph.n), np.int8)
    op = None
    if ordering is not None:
        ordering = i32(ordering).reshape(B, graph.n)
        op = ptr(ordering, C.c_int32)
    check(lib().qldpc_osd0_batch(graph.handle, B, ptr(syndromes, C.c_int8), ptr(llr, C.c_double), ptr(hard, C.c_int8), op, int(flags), ptr(sol, C.c_int8)))
    return sol


def osdw_batch(graph, syndromes, llr, hard, order, max_combinations=None, ordering=None):
    """qldpc_osdw_batch on host arrays: performOSD_enhanced(order, max_combinations) (osd.py:5-77) for B shots -> int8[B, n]."""
    syndromes = i8(syndromes).reshape(-1, graph.m)
    B = syndromes.shape[0]
    llr, hard = f64(llr).reshape(B, graph.n), i8(hard).reshape(B, graph.n)
    sol = np.zeros((B, graph.n), np.int8)
    op = None
    if ordering is not None:
        ordering = i32(ordering).reshape(B, graph.n)
        op = ptr(ordering, C.c_int32)
    check(lib().qldpc_osdw_batch(graph.handle, B, ptr(syndromes, C.c_int8), ptr(llr, C.c_double), ptr(hard, C.c_int8), op, int(order),
                                 int(max_combinations or 0), ptr(sol, C.c_int8)))
    return sol


def gf2_spmv_batch(graph, vectors):
    """s = H e over GF(2) for B vectors (qldpc_gf2_spmv_batch, kernels.py:222-231): int8[B, n] -> int8[B, m]."""
    vectors = i8(vectors).reshape(-1, graph.n)
    out = np.zeros((vectors.shape[0], graph.m), np.int8)
    check(lib().qldpc_gf2_spmv_batch(graph.handle, vectors.shape[0], ptr(vectors, C.c_int8), ptr(out, C.c_int8)))
    return out


def osd_timers(reset=True):
    """Phase counters of the OSD-0 kernels [0..15] and of the workgroup BP kernel [16..31] (diagnostic build only, see csrc/osd_common.h) -> uint64[32]."""
    out = np.zeros(32, np.uint64)
    check(lib().qldpc_osd_timers_read(ptr(out, C.c_uint64), int(reset)))
    return out


class Comm:
    """RCCL communicator of the tally all-reduce (qldpc_comm_*): `Comm.init_all(ndev)` for one process driving ndev GPUs,
    `Comm.init_rank(nranks, rank, id, device)` for one process per GPU (id = Comm.unique_id() of rank 0, handed over by the launcher)."""

    def __init__(self, handle):
        self._h = handle
        nr, nl = C.c_int(0), C.c_int(0)
        check(lib().qldpc_comm_size(self._h, C.byref(nr), C.byref(nl)))
        self.nranks, self.nlocal = nr.value, nl.value

    @staticmethod
    def unique_id():
        buf = np.zeros(128, np.uint8)
        check(lib().qldpc_comm_unique_id(ptr(buf, C.c_uint8)))
        return buf.tobytes()

    @classmethod
    def init_all(cls, ndev, devices=None):
        require_device()
        h = C.c_void_p()
        dv = None if devices is None else ptr(np.ascontiguousarray(devices, np.int32), C.c_int)
        check(lib().qldpc_comm_init_all(int(ndev), dv, C.byref(h)))
        return cls(h)

    @classmethod
    def init_rank(cls, nranks, rank, uid, device):
        require_device()
        h = C.c_void_p()
        buf = np.frombuffer(bytes(uid), np.uint8).copy()
        check(lib().qldpc_comm_init_rank(int(nranks), int(rank), ptr(buf, C.c_uint8), int(device), C.byref(h)))
        return cls(h)

    def allreduce(self, tallies):
        """int64[nlocal, 16] (or [16] when nlocal == 1) -> the sum over all ranks, same shape."""
        t = np.ascontiguousarray(tallies, np.int64).reshape(self.nlocal, TALLY_SLOTS).copy()
        check(lib().qldpc_tally_allreduce(self._h, ptr(t, C.c_int64)))
        return t.reshape(np.shape(tallies))

    def close(self):
        if self._h is not None and self._h.value:
            lib().qldpc_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cc_sample_decode_tally(graph, L, p, seed, shot_begin, count, max_iter=50, alpha=1.0, alpha_mode="dynamical", damping=1.0,
                           clip_llr=20.0, use_osd=True, flags=0):
    mode, aval, seq = alpha_args(alpha_mode, alpha)
    L = u8(L).reshape(-1, graph.n)
    tally = np.zeros(TALLY_SLOTS, np.int64)
    check(lib().qldpc_cc_sample_decode_tally(graph.handle, C.c_int(L.shape[0]), ptr(L, C.c_uint8), C.c_double(p), C.c_uint64(seed),
                                             C.c_int64(shot_begin), C.c_int64(count), C.c_int(max_iter), C.c_int(mode),
                                             C.c_double(aval), ptr(seq, C.c_double), C.c_int(seq.size), C.c_double(damping),
                                             C.c_double(clip_llr), C.c_int(int(use_osd)), C.c_int(flags), ptr(tally, C.c_int64)))
    return tally


class CodeCapacityPlan:
    """Asynchronous code-capacity Monte-Carlo plan (qldpc_cc_plan_*): device-resident sample -> decode -> tally."""

    def __init__(self, graph, L, p, max_iter=50, alpha=1.0, alpha_mode="dynamical", damping=1.0, clip_llr=20.0, use_osd=True,
                 flags=0, batch=1 << 18, min_launch=0):
        """batch: shots per piece, taken literally (the plan's device buffers hold `batch` shots per piece in flight).  min_launch (extension): let THIS plan
        cut its calls at a larger granule instead -- fewer, larger launches; results never depend on the cut."""
        mode, aval, seq = alpha_args(alpha_mode, alpha)
        L = u8(L).reshape(-1, graph.n)
        self.graph = graph
        self._h = C.c_void_p()
        check(lib().qldpc_cc_plan_create(graph.handle, C.c_int(L.shape[0]), ptr(L, C.c_uint8), C.c_double(p), C.c_int(max_iter),
                                         C.c_int(mode), C.c_double(aval), ptr(seq, C.c_double), C.c_int(seq.size), C.c_double(damping),
                                         C.c_double(clip_llr), C.c_int(int(use_osd)), C.c_int(flags), C.c_int64(batch), C.c_int64(min_launch), C.byref(self._h)))

    def run(self, seed, shot_begin, count, stream=0):
        check(lib().qldpc_cc_plan_run(self._h, C.c_uint64(seed), C.c_int64(shot_begin), C.c_int64(count), C.c_void_p(stream)))

    def read(self, stream=0, clear=False):
        tally = np.zeros(TALLY_SLOTS, np.int64)
        check(lib().qldpc_cc_plan_read(self._h, C.c_void_p(stream), C.c_int(int(clear)), ptr(tally, C.c_int64)))
        return tally

    def first_iteration_time(self):
        """ms spent in the bit-sliced first-iteration kernel since the last kernel_time() (call before it)"""
        ms = C.c_double()
        check(lib().qldpc_cc_plan_first_iteration_time(self._h, C.byref(ms)))
        return ms.value

    def kernel_time(self):
        ms, nl = C.c_double(0), C.c_int64(0)
        check(lib().qldpc_cc_plan_kernel_time(self._h, C.byref(ms), C.byref(nl)))
        return ms.value, nl.value

    def clock(self, stream=0):
        """Shader clock (MHz) held under the last fused decode launch (plan created with FLAG_CLOCK_PROBE)."""
        mhz = C.c_double(0)
        check(lib().qldpc_cc_plan_clock(self._h, C.c_void_p(stream), C.byref(mhz)))
        return float(mhz.value)

    def close(self):
        if self._h is not None and self._h.value:
            lib().qldpc_cc_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _attr(src, name):
    return src[name] if isinstance(src, dict) else getattr(src, name)


def logical_column_masks(logical_rows, n):
    """k x n logical rows of H*_full (dense 0/1 or (indptr, indices) CSR) -> uint64[n] with bit r = row r."""
    lm = np.zeros(n, np.uint64)
    if isinstance(logical_rows, tuple):
        ip, ix = logical_rows
        for r in range(len(ip) - 1):
            lm[np.asarray(ix[ip[r]:ip[r + 1]], dtype=np.int64)] |= np.uint64(1) << np.uint64(r)
    else:
        M = np.asarray(logical_rows)
        for r in range(M.shape[0]):
            lm[np.flatnonzero(M[r])] |= np.uint64(1) << np.uint64(r)
    return lm


def make_circuit_desc(compiled, Lx, Lz):
    """CompiledCircuit-like object (or dict of its arrays) -> (CircuitDesc, keep-alive dict)."""
    keep = {k: i32(_attr(compiled, k)) for k in ("base_ops", "base_q1", "base_q2", "suffix_ops", "suffix_q1", "suffix_q2", "x_syn_positions",
                                                 "x_syn_ptrs", "z_syn_positions", "z_syn_ptrs", "data_qubit_indices")}
    keep["Lx"], keep["Lz"] = u8(Lx), u8(Lz)
    d = CircuitDesc()
    d.base_len, d.suffix_len = keep["base_ops"].size, keep["suffix_ops"].size
    for k in ("base_ops", "base_q1", "base_q2", "suffix_ops", "suffix_q1", "suffix_q2", "x_syn_positions", "x_syn_ptrs", "z_syn_positions",
              "z_syn_ptrs", "data_qubit_indices"):
        setattr(d, k, ptr(keep[k], C.c_int32))
    d.total_qubits = int(_attr(compiled, "total_qubits"))
    d.num_x_checks, d.num_z_checks = keep["x_syn_ptrs"].size - 1, keep["z_syn_ptrs"].size - 1
    d.n_data, d.k = keep["data_qubit_indices"].size, keep["Lx"].shape[0]
    d.Lx, d.Lz = ptr(keep["Lx"], C.c_uint8), ptr(keep["Lz"], C.c_uint8)
    return d, keep


def circuit_fault_signatures(compiled, Lx, Lz, sector_is_x):
    """(ptr int32[2*L+1], idx uint16[...], logmask uint64[2*L]) of every single-qubit flip at every base-circuit location."""
    require_device()
    d, keep = make_circuit_desc(compiled, Lx, Lz)
    L = keep["base_ops"].size
    sp = np.zeros(2 * L + 1, np.int32)
    lm = np.zeros(2 * L, np.uint64)
    cap = max(1024, 64 * 2 * L)
    idx = np.zeros(cap, np.uint16)
    need = C.c_int64(0)
    check(lib().qldpc_circuit_fault_signatures(C.byref(d), C.c_int(int(sector_is_x)), ptr(sp, C.c_int32), ptr(idx, C.c_uint16), C.c_int64(cap),
                                               ptr(lm, C.c_uint64), C.byref(need)))
    return sp, idx[:need.value].copy(), lm


STATS_CHECK_MESSAGES, STATS_POSTERIOR = 0, 1


class MessageStats:
    """Device-resident samples of an estimator trial loop (qldpc_msgstats_*): check messages after `iters` decoder iterations
    (alpha.py:119-137, 206-255) or the decoder's final posteriors (scopt.py:80-134), split by the true error bit."""

    def __init__(self, graph, errors, prior, kind, iters, alpha_mode="dynamical", alpha=1.0, damping=1.0, clip_llr=20.0):
        errors = i8(errors).reshape(-1, graph.n)
        prior = f64(prior)
        if prior.size != graph.n:
            raise ValueError(f"prior has {prior.size} entries, the graph has {graph.n} columns")
        mode, aval, seq = alpha_args(alpha_mode, alpha)
        rng, fin = np.zeros(2), np.zeros(2, np.int64)
        self._h = C.c_void_p()
        check(lib().qldpc_msgstats_create(graph.handle, C.c_int64(errors.shape[0]), ptr(errors, C.c_int8), ptr(prior, C.c_double),
                                          C.c_int(kind), C.c_int(iters), C.c_int(mode), C.c_double(aval), ptr(seq, C.c_double),
                                          C.c_int(seq.size), C.c_double(damping), C.c_double(clip_llr), ptr(rng, C.c_double),
                                          ptr(fin, C.c_int64), C.byref(self._h)))
        self.range = (float(rng[0]), float(rng[1]))
        self.finite = (int(fin[0]), int(fin[1]))

    def histogram(self, edges):
        """Counts per bin for the two classes, np.histogram's bin rule -> (int64[bins], int64[bins])."""
        edges = f64(edges)
        bins = edges.size - 1
        h0, h1 = np.zeros(bins, np.int64), np.zeros(bins, np.int64)
        check(lib().qldpc_msgstats_histogram(self._h, ptr(edges, C.c_double), C.c_int(bins), ptr(h0, C.c_int64), ptr(h1, C.c_int64)))
        return h0, h1

    def close(self):
        if self._h is not None and self._h.value:
            lib().qldpc_msgstats_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CircuitPlan:
    """Circuit-level Monte-Carlo plan (qldpc_circuit_plan_*): Philox fault sampling through precomputed fault signatures,
    decode of both sectors, OSD-0, logical comparison and tally, all on the device."""

    def __init__(self, compiled, Lx, Lz, graph_z, graph_x, prior_z, prior_x, logmask_z, logmask_x, p, max_iter=50, alpha_z=1.0, alpha_x=1.0,
                 alpha_mode="dynamical", damping=1.0, clip_llr=20.0, use_osd=True, flags=0, batch=16384):
        mode, az, sz = alpha_args(alpha_mode, alpha_z)
        _, ax, sx = alpha_args(alpha_mode, alpha_x)
        d, keep = make_circuit_desc(compiled, Lx, Lz)
        pz, px = f64(prior_z), f64(prior_x)
        lz, lx = np.ascontiguousarray(logmask_z, np.uint64), np.ascontiguousarray(logmask_x, np.uint64)
        self.k, self.nsx, self.nsz = d.k, int(keep["x_syn_ptrs"][-1]), int(keep["z_syn_ptrs"][-1])
        self.graph_z, self.graph_x = graph_z, graph_x
        self.flags = flags
        self._h = C.c_void_p()
        check(lib().qldpc_circuit_plan_create(C.byref(d), graph_z.handle, graph_x.handle, ptr(pz, C.c_double), ptr(px, C.c_double),
                                              ptr(lz, C.c_uint64), ptr(lx, C.c_uint64), C.c_double(p), C.c_int(max_iter), C.c_int(mode),
                                              C.c_double(az), C.c_double(ax), ptr(sz, C.c_double), C.c_int(sz.size), ptr(sx, C.c_double),
                                              C.c_int(sx.size), C.c_double(damping), C.c_double(clip_llr), C.c_int(int(use_osd)), C.c_int(flags),
                                              C.c_int64(batch), C.byref(self._h)))

    def run(self, seed, trial_begin, count, stream=0):
        check(lib().qldpc_circuit_plan_run(self._h, C.c_uint64(seed), C.c_int64(trial_begin), C.c_int64(count), C.c_void_p(stream)))

    def run_outcomes(self, seed, trial_begin, count, stream=0):
        """run() + the per-trial verdicts in trial order: uint8[count], bit0 = z_err, bit1 = x_err."""
        out = np.zeros(max(int(count), 0), np.uint8)
        check(lib().qldpc_circuit_plan_run_outcomes(self._h, C.c_uint64(seed), C.c_int64(trial_begin), C.c_int64(count), C.c_void_p(stream),
                                                    ptr(out, C.c_uint8)))
        return out

    def read(self, stream=0, clear=False):
        tally = np.zeros(TALLY_SLOTS, np.int64)
        check(lib().qldpc_circuit_plan_read(self._h, C.c_void_p(stream), C.c_int(int(clear)), ptr(tally, C.c_int64)))
        return tally

    def phase_times(self):
        """({phase: ms summed over the batches since the last call}, batches): hipEvent brackets of sampler / BP / OSD-0 per sector / judge."""
        ms = (C.c_double * len(CIRCUIT_PHASES))()
        nb = C.c_int64(0)
        check(lib().qldpc_circuit_plan_phase_times(self._h, ms, C.byref(nb)))
        return {k: float(ms[i]) for i, k in enumerate(CIRCUIT_PHASES)}, int(nb.value)

    def clock(self, stream=0):
        """Shader clock (MHz) under the sector-Z decode and OSD-0 kernels of the last batch (plan created with FLAG_CLOCK_PROBE)."""
        mhz = (C.c_double * 2)()
        check(lib().qldpc_circuit_plan_clock(self._h, C.c_void_p(stream), mhz))
        return float(mhz[0]), float(mhz[1])

    def sample(self, seed, trial_begin, count):
        """Batched run_trial_fast -> (sparse_z int8[count, nsx], true_z int8[count, k], sparse_x, true_x)."""
        spz, spx = np.zeros((count, self.nsx), np.int8), np.zeros((count, self.nsz), np.int8)
        tz, tx = np.zeros((count, self.k), np.int8), np.zeros((count, self.k), np.int8)
        check(lib().qldpc_circuit_plan_sample(self._h, C.c_uint64(seed), C.c_int64(trial_begin), C.c_int64(count), ptr(spz, C.c_int8),
                                              ptr(tz, C.c_int8), ptr(spx, C.c_int8), ptr(tx, C.c_int8)))
        return spz, tz, spx, tx

    def close(self):
        if self._h is not None and self._h.value:
            lib().qldpc_circuit_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
