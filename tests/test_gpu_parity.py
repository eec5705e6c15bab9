"""GPU parity tests (-m gpu): the HIP path, called through the C ABI / the mirrored reference API, against the
golden vectors (reference source outputs) and against the oracle on seeded inputs.  Integer / GF(2) results
must be bit-exact; LLRs are held to the north-star tolerance 1e-5 (and are in fact bit-identical)."""
import os

import numpy as np
import pytest

try:            # this image carries two HIP runtimes (system ROCm and the one bundled with torch): import torch FIRST so that
    import torch  # noqa: F401   # libqldpc_hip.so binds to the already-loaded runtime and both share the device
except ImportError:
    torch = None

from conftest import ROOT, assert_llr_close, experiments_lib, package_of

pytestmark = pytest.mark.gpu
LLR_TOL = 1e-5


@pytest.fixture(scope="module")
def L():
    import qldpc_amd  # noqa: F401
    from qldpc_amd import _lib
    if not os.path.exists(_lib.SO_PATH):          # a fresh checkout: build the HIP library first (never fall back to anything else)
        import __graft_entry__
        __graft_entry__.build()
    _lib.require_device()
    return _lib


@pytest.fixture(scope="module")
def LX(L):
    """libqldpc_hip_experiments.so: the same ABI plus the measured-and-rejected kernels (make -C csrc experiments)"""
    X = experiments_lib()
    if not os.path.exists(X.SO_PATH):
        import __graft_entry__
        __graft_entry__.build()
    X.require_device()
    return X


# kernel variants that exist in the experiments build only (include/qldpc_hip.h); the product library refuses them
XFLAGS = 0x2 | 0x40000 | 0x100000      # WG_EDGE_LANES, WG_IDXLOAD, OSD_QUEUE


def for_build(L, variants, key=lambda v: v):
    """the kernel variants a build is asked for: the product library everything but the experiments, the experiments build only those"""
    x = L.BUILD == "experiments"
    return [v for v in variants if bool(key(v) & XFLAGS) == x]


@pytest.fixture(params=["product", "experiments"])
def Lb(request, L):
    return L if request.param == "product" else request.getfixturevalue("LX")


KERNELS = {"regular": 0x20, "generic": 0x20 | 0x40, "stream": 0x10}   # "regular" falls back to generic on irregular graphs
VARIANT_KW = {
    "const": dict(alpha=0.8, alpha_mode="alvarado"),
    "damp": dict(alpha=1.0, alpha_mode="dynamical", damping=0.7),
    "clip5": dict(alpha=1.0, alpha_mode="dynamical", clip_llr=5.0),
    "dampclip": dict(alpha=0.9, alpha_mode="alvarado", damping=0.5, clip_llr=6.0),
}


def check(out, g, key):
    err, conv, llr, it = out
    assert np.array_equal(err, g[key + "_err"]), key
    assert np.array_equal(conv, g[key + "_conv"]), key
    assert np.array_equal(it, g[key + "_iter"]), key
    assert_llr_close(llr, g[key + "_llr"], LLR_TOL)
    assert np.array_equal(llr, g[key + "_llr"], equal_nan=True), key + ": LLRs not bit-identical"


def decode(L, graph, synd, prior, max_iter, flags, alpha=1.0, alpha_mode="dynamical", damping=1.0, clip_llr=20.0):
    return L.minsum_decode_batch(graph, synd, prior, max_iter, alpha_mode, alpha, damping, clip_llr, flags)


@pytest.mark.parametrize("kern", ["regular", "generic", "stream"])
def test_steane_golden(L, golden, kern):
    g = golden("steane_minsum")
    graph = L.Graph(g["indptr"], g["indices"], 7)
    mi = int(g["max_iter"])
    modes = {"dyn": dict(alpha=1.0, alpha_mode="dynamical"), "const": dict(alpha=0.8, alpha_mode="alvarado"),
             "seq": dict(alpha=g["seq_alpha"], alpha_mode="alvarado-autoregressive"), "none0": dict(alpha=0, alpha_mode=None),
             "none1": dict(alpha=0.9, alpha_mode=None)}
    for tag, kw in modes.items():
        check(decode(L, graph, g["syndromes"], g["prior"], mi, KERNELS[kern], **kw), g, tag)
    check(decode(L, graph, g["syndromes"], g["prior2"], mi, KERNELS[kern]), g, "p2")


@pytest.mark.parametrize("kern", ["regular", "generic", "stream"])
@pytest.mark.parametrize("tag", ["bb72", "bb144", "bb288"])
def test_bb_golden(L, golden, tag, kern):
    g = golden(tag + "_minsum")
    fl = KERNELS[kern]
    for h in ("Hx", "Hz"):
        n = int(g[h + "_shape"][1])
        graph = L.Graph(g[h + "_indptr"], g[h + "_indices"], n)
        for p in ("p005", "p030", "p080"):
            base = f"{h}_{p}"
            synd, prior = g[base + "_syndromes"], g[base + "_prior"]
            for mi in (1, 5, 50):
                check(decode(L, graph, synd, prior, mi, fl), g, f"{base}_dyn_it{mi}")
            # fixed-work mode must give the same outputs (frozen at convergence)
            check(decode(L, graph, synd, prior, 50, fl | L.FLAG_FIXED_ITERS), g, f"{base}_dyn_it50")
            if h == "Hx":
                for v, kw in VARIANT_KW.items():
                    check(decode(L, graph, synd, prior, 30, fl, **kw), g, f"{base}_{v}")
                check(decode(L, graph, synd, prior, 30, fl, alpha=g["seq_alpha"], alpha_mode="alvarado-autoregressive"), g, f"{base}_seq")


@pytest.mark.parametrize("tag", ["circ72", "circ144"])
def test_circuit_level_golden(Lb, golden, oracle, tag):
    L = Lb
    from qldpc_amd.data import load_circuit_matrices
    import importlib
    performOSD_enhanced = importlib.import_module(package_of(L).__name__ + ".decoding.osd").performOSD_enhanced
    import scipy.sparse as sp
    g = golden(tag + "_decode")
    data = load_circuit_matrices(tag)
    for s in ("Z", "X"):
        ip, ix = data[f"Hdec{s}_indptr"], data[f"Hdec{s}_indices"]
        m, n = (int(x) for x in data[f"Hdec{s}_shape"])
        graph = L.Graph(ip, ix, n)
        # auto = workgroup-per-shot kernel, LDS-resident form (csrc/minsum_wg2.hip: the prior is known on the host); FLAG_WG_TABLES = the form with its
        # index tables in HBM / L2 (csrc/minsum_wg.hip, what the *_dev entry points run); streaming kernel forced
        for fl in for_build(L, (0, L.FLAG_FIXED_ITERS, L.FLAG_WG_TABLES, L.FLAG_WG_TABLES | L.FLAG_FIXED_ITERS, L.FLAG_KERNEL_STREAM)):
            check(decode(L, graph, g[f"{s}_syndromes"], g[f"llrs_{s}"], int(g["max_iter"]), fl), g, s)
        # the generic (any-input) workgroup kernel and the natural row / column order must agree with the lean, degree-sorted default
        for fl in for_build(L, (L.FLAG_WG_GENERIC, L.FLAG_WG_ROWMAJOR, L.FLAG_WG_GENERIC | L.FLAG_WG_ROWMAJOR, L.FLAG_WG_EDGE_LANES, L.FLAG_WG_EDGE_LANES | L.FLAG_FIXED_ITERS,
                                L.FLAG_WG_IDXLOAD, L.FLAG_WG_IDXLOAD | L.FLAG_FIXED_ITERS)):     # (default for m <= 1024: column indices resident in registers)
            check(decode(L, graph, g[f"{s}_syndromes"], g[f"llrs_{s}"], int(g["max_iter"]), fl), g, s)
        rng = np.random.default_rng(5)                                 # ragged random batch, both kernels, vs the oracle
        synd = (rng.random((37, m)) < 0.1).astype(np.int8)
        ref = oracle.minsum_decode_batch(ip, ix, n, synd, g[f"llrs_{s}"], max_iter=12, threads=0)
        for fl in (0, L.FLAG_KERNEL_STREAM):
            for a, b in zip(decode(L, graph, synd, g[f"llrs_{s}"], 12, fl), ref):
                assert np.array_equal(a, b, equal_nan=True)
        H = sp.csr_matrix((np.ones(ix.size, np.int8), ix, ip), shape=(m, n))
        for t, case in enumerate(g[f"{s}_osd_cases"]):
            sol = performOSD_enhanced(H, g[f"{s}_syndromes"][case], g[f"{s}_llr"][case], g[f"{s}_err"][case], order=0,
                                      ordering=g[f"{s}_osd_ordering"][t])
            assert np.array_equal(sol, g[f"{s}_osd_solution"][t])
            sol_w = performOSD_enhanced(H, g[f"{s}_syndromes"][case], g[f"{s}_llr"][case], g[f"{s}_err"][case], order=2,
                                        ordering=g[f"{s}_osd_ordering"][t])       # consistent syndrome: OSD-2 == OSD-0 (osd.py:27-29)
            assert np.array_equal(sol_w, sol)
            # the drop-in call (no ordering=): the wrapper evaluates the reference's own np.argsort(np.abs(llr)) (osd.py:11-12) on the host.  Every fixture holds
            # runs of equal keys, so the tie order is NumPy's choice on this machine -- the reference's answer must come out whichever it is
            sol_d = performOSD_enhanced(H, g[f"{s}_syndromes"][case], g[f"{s}_llr"][case], g[f"{s}_err"][case], order=0)
            assert np.array_equal(sol_d, g[f"{s}_osd_solution"][t]), (tag, s, t, "drop-in call")
            # the stable ordering (ascending |llr|, ties by index: what the batched device entry points use): identical to the oracle's with the same rule
            stable = np.argsort(np.abs(g[f"{s}_llr"][case]), kind="stable")
            sol2 = performOSD_enhanced(H, g[f"{s}_syndromes"][case], g[f"{s}_llr"][case], g[f"{s}_err"][case], order=0, ordering=stable)
            sol2b = L.osd0_batch(graph, g[f"{s}_syndromes"][case][None], g[f"{s}_llr"][case][None], g[f"{s}_err"][case][None])[0]       # batched entry point, no ordering
            assert np.array_equal(sol2, sol2b)
            try:        # the device's own sort: a short sorted head + the rest on demand, a head of 2000 columns, the whole order up front (option "osd_presort")
                for presort in (40, 2000, 0):
                    L.set_option("osd_presort", presort)
                    assert np.array_equal(L.osd0_batch(graph, g[f"{s}_syndromes"][case][None], g[f"{s}_llr"][case][None], g[f"{s}_err"][case][None])[0], sol2), (tag, s, t, presort)
            finally:
                L.set_option("osd_presort", -1)
            ref2 = oracle.osd0(ip, ix, n, g[f"{s}_syndromes"][case], g[f"{s}_llr"][case], g[f"{s}_err"][case])
            assert np.array_equal(sol2, ref2)
            assert np.array_equal(oracle.syndrome_check(ip, ix, sol2.astype(np.int8)), g[f"{s}_syndromes"][case])
            # the general global-memory OSD kernel (used when m > 1024) must agree with the LDS-resident one
            for kfl in for_build(L, (L.FLAG_OSD_GLOBAL, L.FLAG_OSD_REFORDER, L.FLAG_OSD_QUEUE)):   # ... and the other forms of the LDS kernel
                sol3 = performOSD_enhanced(H, g[f"{s}_syndromes"][case], g[f"{s}_llr"][case], g[f"{s}_err"][case], order=0, ordering=stable, flags=kfl)
                sol4 = performOSD_enhanced(H, g[f"{s}_syndromes"][case], g[f"{s}_llr"][case], g[f"{s}_err"][case], order=0,
                                           ordering=g[f"{s}_osd_ordering"][t], flags=kfl)
                assert np.array_equal(sol3, ref2) and np.array_equal(sol4, g[f"{s}_osd_solution"][t]), (tag, s, t, kfl)
        # OSD-0 on arbitrary (also inconsistent) inputs: random syndromes / llrs with heavy ties / hard decisions
        rng2 = np.random.default_rng(11)
        for trial in range(3):
            sy = (rng2.random(m) < 0.3).astype(np.int8)
            ll = np.round(rng2.normal(0, 2, n), 1 if trial else 0)          # many exact ties, zeros
            hd = (rng2.random(n) < 0.05).astype(np.int8)
            assert np.array_equal(performOSD_enhanced(H, sy, ll, hd, order=0, ordering=np.argsort(np.abs(ll), kind="stable")), oracle.osd0(ip, ix, n, sy, ll, hd)), (tag, s, trial)
            assert np.array_equal(L.osd0_batch(graph, sy[None], ll[None], hd[None])[0], oracle.osd0(ip, ix, n, sy, ll, hd)), (tag, s, trial)
            try:        # (an inconsistent right-hand side: the sweep runs through every chunk, i.e. past any sorted head)
                L.set_option("osd_presort", 1500)
                assert np.array_equal(L.osd0_batch(graph, sy[None], ll[None], hd[None])[0], oracle.osd0(ip, ix, n, sy, ll, hd)), (tag, s, trial, "head 1500")
            finally:
                L.set_option("osd_presort", -1)


@pytest.mark.parametrize("tag", ["circ72", "circ144"])
def test_workgroup_kernel_lds_resident_form(L, oracle, tag):
    """The two forms of the workgroup-per-shot decoder -- tables in LDS (csrc/minsum_wg2.hip: host-known prior, column slots by (degree, prior) class) and
    tables in HBM / L2 (csrc/minsum_wg.hip) -- against the oracle and each other on inputs that take different paths through the first one: the real
    priors (11 - 15 classes, a few chunks that mix classes), a two-valued prior scattered at random, a uniform prior, a prior of all-distinct values (not
    eligible: falls back to the table kernel whatever the flag), ragged batches, few and many iterations, the fixed-work mode, and both sectors (HdecX has
    degree-1 checks: +-inf messages and the NaN -> 0 rule of kernels.py:328)."""
    from qldpc_amd.data import load_circuit_matrices
    from qldpc_amd.simulation.engine import prior_llrs
    d = load_circuit_matrices(tag)
    rng = np.random.default_rng(23)
    for sct in "ZX":
        ip, ix = d[f"Hdec{sct}_indptr"], d[f"Hdec{sct}_indices"]
        m, n = (int(x) for x in d[f"Hdec{sct}_shape"])
        graph = L.Graph(ip, ix, n)
        real = prior_llrs(d[f"channel_probs{sct}"])
        priors = {"real": real, "two values": np.where(rng.random(n) < 0.3, 2.5, 6.25), "uniform": np.full(n, 4.0),
                  "all distinct": 3.0 + np.arange(n) * 1e-3, "with a negative class": np.where(rng.random(n) < 0.1, -1.5, real)}
        for name, pr in priors.items():
            for B, iters, p_err in ((5, 3, 0.004), (67, 17, 0.006)):
                E = (rng.random((B, n)) < p_err).astype(np.int8)
                synd = np.stack([oracle.syndrome_check(ip, ix, e) for e in E])
                synd[0] = rng.random(m) < 0.4                      # an unrealisable syndrome: never converges
                ref = oracle.minsum_decode_batch(ip, ix, n, synd, pr, max_iter=iters, threads=0)
                for fl in (0, L.FLAG_FIXED_ITERS, L.FLAG_WG_TABLES):
                    got = L.minsum_decode_batch(graph, synd, pr, iters, "dynamical", 1.0, flags=fl)
                    for x, y, what in zip(got, ref, ("err", "conv", "llr", "iter")):
                        assert np.array_equal(x, y, equal_nan=(what == "llr")), (tag, sct, name, B, iters, fl, what)
                got = L.minsum_decode_batch(graph, synd, pr, iters, "alvarado", 0.8, clip_llr=7.5)       # another alpha schedule and clip bound
                ref2 = oracle.minsum_decode_batch(ip, ix, n, synd, pr, max_iter=iters, alpha=0.8, alpha_mode="alvarado", clip_llr=7.5, threads=0)
                for x, y in zip(got, ref2):
                    assert np.array_equal(x, y, equal_nan=True), (tag, sct, name, "alvarado")


@pytest.mark.parametrize("kern", ["regular", "generic", "stream"])
def test_random_batch_vs_oracle(L, oracle, kern):
    """4096 seeded shots per point, hard regime included; ragged batch size (not a multiple of any tile)."""
    from qldpc_amd.data import load_code
    rng = np.random.default_rng(7)
    for tag, p, B in (("bb144", 0.06, 4099), ("bb72", 0.1, 1031), ("bb288", 0.04, 777), ("bb90", 0.05, 513), ("bb108", 0.05, 300)):
        c = load_code(tag)
        ip, ix, n = c["Hx_indptr"], c["Hx_indices"], c["n"]
        graph = L.Graph(ip, ix, n)
        errs = (rng.random((B, n)) < p).astype(np.int8)
        synd = (errs @ c["Hx"].T.astype(np.int64) % 2).astype(np.int8)
        prior = np.full(n, np.log((1 - p) / p))
        ref = oracle.minsum_decode_batch(ip, ix, n, synd, prior, max_iter=50, threads=0)
        for flags in (KERNELS[kern], KERNELS[kern] | L.FLAG_FIXED_ITERS):
            out = decode(L, graph, synd, prior, 50, flags)
            for a, b in zip(out, ref):
                assert np.array_equal(a, b, equal_nan=True)
        assert (ref[1] == 0).any() and (ref[3] > 3).any()


def test_edge_cases(L, oracle):
    # empty batch, B=1, max_iter=1, a graph with an empty row and an isolated column, prior beyond the clip
    ip = np.array([0, 2, 2, 5], np.int32)
    ix = np.array([0, 2, 1, 2, 4], np.int32)
    n = 6
    graph = L.Graph(ip, ix, n)
    prior = np.array([25.0, 0.5, -1.0, 3.0, 2.0, 1.0])
    out = decode(L, graph, np.zeros((0, 3), np.int8), prior, 10, 0)
    assert out[0].shape == (0, n)
    synds = np.array([[0, 0, 0], [1, 0, 1], [0, 1, 0], [1, 1, 1]], np.int8)
    for kern in KERNELS.values():
        for mi in (1, 2, 7):
            ref = oracle.minsum_decode_batch(ip, ix, n, synds, prior, max_iter=mi)
            out = decode(L, graph, synds, prior, mi, kern)
            for a, b in zip(out, ref):
                assert np.array_equal(a, b, equal_nan=True)
    with pytest.raises(L.QldpcError):
        L.Graph(np.array([0, 2], np.int32), np.array([1, 0], np.int32), 2)      # unsorted row


def test_reference_named_api(L, golden):
    import scipy.sparse as sp
    from qldpc_amd.decoding.sparse import performMinSum_Symmetric_Sparse
    from qldpc_amd.decoding.dense import performMinSum_Symmetric, performBeliefPropagationFast
    from qldpc_amd.decoding import kernels as K
    g = golden("core_passes")
    m, n = (int(x) for x in g["shape"])
    ip, ix = g["indptr"], g["indices"]
    H = np.zeros((m, n))
    for i in range(m):
        H[i, ix[ip[i]:ip[i + 1]]] = 1.0
    Hc = sp.csr_matrix(H)
    mask = H != 0
    for t, s in enumerate(g["bpdrv_syndromes"][:8]):
        for fn, Hin in ((performMinSum_Symmetric_Sparse, Hc), (performMinSum_Symmetric, H)):
            e, c, v, it = fn(Hin, s, g["bpdrv_prior"], maxIter=12)
            assert e.dtype == np.int8 and isinstance(c, bool) and isinstance(it, int)
            assert np.array_equal(e, g["dense_err"][t]) and c == bool(g["dense_conv"][t]) and it == g["dense_iter"][t]
            assert np.array_equal(v, g["dense_llr"][t], equal_nan=True)
        e, c, v, it = performBeliefPropagationFast(H, s, g["bpdrv_prior"], maxIter=12)
        assert np.array_equal(e, g["bpdrv_err"][t]) and c == bool(g["bpdrv_conv"][t]) and it == g["bpdrv_iter"][t]
        assert_llr_close(v, g["bpdrv_llr"][t], LLR_TOL)
    e, c, R, it = performMinSum_Symmetric(H, g["bpdrv_syndromes"][3], g["bpdrv_prior"], maxIter=12, alpha_estimation=True)
    assert it == 0 and c is False and np.array_equal(R, g["alphaest_R"], equal_nan=True)
    for t in range(g["Q"].shape[0]):
        R, Rs = K.minsum_core_sparse(np.ones(ix.size, np.int8), ix, ip, g["Q"][t], g["syndrome_sign"][t], float(g["alphas"][t]), m, n)
        assert np.array_equal(R, g["R_flat"][t], equal_nan=True) and np.array_equal(Rs, g["R_sum"][t], equal_nan=True)
        Qd = np.zeros((m, n)); Qd[mask] = g["Q"][t]
        Rd = K.minsum_core(H, Qd, g["syndrome_sign"][t].reshape(-1, 1), mask, float(g["alphas"][t]))
        assert np.array_equal(Rd, g["R_dense"][t], equal_nan=True)
    for t in range(g["bp_Q"].shape[0]):
        Qd = np.zeros((m, n)); Qd[mask] = g["bp_Q"][t]
        Rb = K.bp_core(H, Qd, g["syndrome_sign"][t].reshape(-1, 1), mask, 0.9999999)
        assert_llr_close(Rb, g["bp_R_dense"][t], LLR_TOL)
    for cnd, ref in zip(g["sc_candidates"], g["sc_syndromes"]):
        assert np.array_equal(K.syndrome_check(np.ones(ix.size, np.int8), ix, ip, cnd, m), ref)
    with pytest.raises(ValueError):
        performMinSum_Symmetric_Sparse(Hc, g["bpdrv_syndromes"][0], g["bpdrv_prior"], alpha_mode="alvarado", alpha=0.0)
    with pytest.raises(ValueError):
        performMinSum_Symmetric_Sparse(Hc, g["bpdrv_syndromes"][0], g["bpdrv_prior"], alpha_mode="alvarado-autoregressive", alpha=[])
    with pytest.raises(ValueError):
        performMinSum_Symmetric(H, g["bpdrv_syndromes"][0], g["bpdrv_prior"], alpha_mode="nope")


def test_bb_golden_256(L, golden):
    """>= 256 reference-decoded syndromes per (code, p) point: Hx of [[72,12,6]] / [[144,12,12]] / [[288,12,18]] at p = 0.005 / 0.02 /
    0.05, decoder defaults, all three resident / streaming kernels; bit-identical posteriors."""
    from qldpc_amd.data import load_code
    g = golden("bb_256")
    for tag in ("bb72", "bb144", "bb288"):
        c = load_code(tag)
        m, n = (int(x) for x in g[f"{tag}_shape"])
        graph = L.Graph(c["Hx_indptr"], c["Hx_indices"], n)
        for p in (0.005, 0.02, 0.05):
            k = f"{tag}_p{int(round(p * 1000)):03d}"
            errs = np.unpackbits(g[f"{k}_errors"], axis=1, bitorder="little")[:, :n].astype(np.int8)
            hard = np.unpackbits(g[f"{k}_hard"], axis=1, bitorder="little")[:, :n].astype(np.int8)
            synd = L.gf2_spmv_batch(graph, errs)                                     # a6 on the device: s = H e
            assert np.array_equal(synd, (errs.astype(np.int64) @ c["Hx"].T.astype(np.int64) % 2).astype(np.int8))
            prior = np.full(n, np.log((1 - p) / p))
            for fl in (0, L.FLAG_KERNEL_GENERIC, L.FLAG_KERNEL_STREAM, L.FLAG_FIXED_ITERS):
                err, conv, llr, it = L.minsum_decode_batch(graph, synd, prior, 50, "dynamical", 1.0, flags=fl)
                assert np.array_equal(err, hard) and np.array_equal(conv.astype(bool), g[f"{k}_conv"].astype(bool)), (k, fl)
                assert np.array_equal(it, g[f"{k}_iter"]) and np.array_equal(llr, g[f"{k}_llr"]), (k, fl)


def test_gf2_elimination_production_size(L, golden):
    """The 1008 x 8785 instance of the [[144,12,12]] x 12-cycle experiment, columns in the |llr| order of a non-converged decode, through
    gf2_elimination_packed (kernels.py:48-106) in place: reduced packed matrix, right-hand side and pivots equal the reference's."""
    from qldpc_amd.decoding import kernels as K
    from qldpc_amd.data import load_circuit_matrices
    g, gd = golden("gf2_big"), golden("circ144_decode")
    d = load_circuit_matrices("circ144")
    m, n = (int(x) for x in d["HdecZ_shape"])
    H = np.zeros((m, n), np.int64)
    ip, ix = d["HdecZ_indptr"], d["HdecZ_indices"]
    for i in range(m):
        H[i, ix[ip[i]:ip[i + 1]]] = 1
    case = int(g["case"])
    b0 = (gd["Z_syndromes"][case].astype(np.int64) + H @ gd["Z_err"][case].astype(np.int64)) % 2
    assert np.array_equal(b0, g["b"])                                                # osd.py:8-9 reproduced from the shipped data
    A = H[:, g["ordering"].astype(np.int64)].copy()
    b = b0.copy()
    Ap, bp, pr, pc = K.gf2_elimination_packed(A, b)
    assert bp is b                                                                   # in place, like the reference
    assert np.array_equal(pr, g["pivot_rows"]) and np.array_equal(pc, g["pivot_cols"]) and np.array_equal(b, g["b_red"])
    assert np.array_equal(np.asarray(Ap, np.uint64), g["A_packed_red"])


def test_gf2_elimination_golden(L, golden):
    from qldpc_amd.decoding import kernels as K
    g = golden("gf2_elimination")
    for tag in g["cases"]:
        A = g[f"{tag}_A"].astype(np.int64)
        b = g[f"{tag}_b"].astype(np.int64)
        A1, b1 = A.copy(), b.copy()
        Ar, br, pr, pc = K.gf2_elimination(A1, b1)
        assert Ar is A1 and br is b1                                   # in place, like the reference
        assert np.array_equal(A1, g[f"{tag}_A_red"]) and np.array_equal(b1, g[f"{tag}_b_red"])
        assert np.array_equal(pr, g[f"{tag}_pivot_rows"]) and np.array_equal(pc, g[f"{tag}_pivot_cols"])
        A2, b2 = A.copy(), b.copy()
        Ap, bp, pr2, pc2 = K.gf2_elimination_packed(A2, b2)
        assert np.array_equal(Ap, g[f"{tag}_A_packed_red"]) and np.array_equal(b2, g[f"{tag}_b_red"])
        assert np.array_equal(pr2, pr) and np.array_equal(pc2, pc)


@pytest.mark.parametrize("tag", ["circ72", "circ144"])
def test_noise_kernels_golden(L, golden, tag):
    from qldpc_amd.noise import kernels as NK
    g = golden(tag + "_noise")
    cap = int(g["max_circuit_size"])
    tq = int(g["total_qubits"])
    for t, p in enumerate(g["error_rates"]):
        oo, o1, o2 = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        Ln = NK.generate_noisy_circuit_jit(g["base_ops"], g["base_q1"], g["base_q2"], float(p), g["random_vals"][t],
                                           g["random_paulis"][t], g["random_two_qubit"][t], oo, o1, o2)
        assert Ln == g["noisy_len"][t]
        assert np.array_equal(oo[:Ln], g["noisy_ops"][t][:Ln]) and np.array_equal(o1[:Ln], g["noisy_q1"][t][:Ln])
        assert np.array_equal(o2[:Ln], g["noisy_q2"][t][:Ln])
        ops = np.concatenate([oo[:Ln], g["suffix_ops"]]); q1 = np.concatenate([o1[:Ln], g["suffix_q1"]])
        q2 = np.concatenate([o2[:Ln], g["suffix_q2"]])
        hz, sz, ncz, ecz = NK.simulate_circuit_Z_jit(ops, q1, q2, tq, g["x_check_indices"], g["x_check_ptrs"], int(g["max_syndromes_x"]))
        hx, sx, ncx, ecx = NK.simulate_circuit_X_jit(ops, q1, q2, tq, g["z_check_indices"], g["z_check_ptrs"], int(g["max_syndromes_z"]))
        assert np.array_equal(hz, g["hist_z"][t]) and np.array_equal(sz, g["state_z"][t])
        assert np.array_equal(hx, g["hist_x"][t]) and np.array_equal(sx, g["state_x"][t])
        assert [ncz, ecz, ncx, ecx] == g["counts"][t].tolist()
        spz = NK.sparsify_syndrome_jit(hz, ncz, g["x_syn_positions"], g["x_syn_ptrs"], int(g["num_x_checks"]))
        spx = NK.sparsify_syndrome_jit(hx, ncx, g["z_syn_positions"], g["z_syn_ptrs"], int(g["num_z_checks"]))
        assert np.array_equal(spz, g["sparse_z"][t]) and np.array_equal(spx, g["sparse_x"][t])


def test_code_capacity_tally_matches_oracle(L, oracle):
    """Same Philox streams on both sides -> identical tallies; and the result is independent of how shots are split."""
    from qldpc_amd.data import load_code
    for tag, p, N in (("bb144", 0.02, 30000), ("bb72", 0.05, 20000), ("steane", 0.01, 1000)):
        c = load_code(tag)
        ip, ix, n = c["Hx_indptr"], c["Hx_indices"], c["n"]
        graph = L.Graph(ip, ix, n)
        ref = oracle.cc_sample_decode_tally(ip, ix, n, c["Lx"], p, 20260206, 0, N, max_iter=50, threads=0)
        t1 = L.cc_sample_decode_tally(graph, c["Lx"], p, 20260206, 0, N, max_iter=50)
        assert np.array_equal(t1, ref), (tag, t1.tolist(), ref.tolist())
        ta = L.cc_sample_decode_tally(graph, c["Lx"], p, 20260206, 0, N // 3, max_iter=50)
        tb = L.cc_sample_decode_tally(graph, c["Lx"], p, 20260206, N // 3, N - N // 3, max_iter=50)
        assert np.array_equal(ta + tb, ref)
        for fl in (L.FLAG_FIXED_ITERS | L.FLAG_KERNEL_STREAM, L.FLAG_MC_UNFUSED, L.FLAG_FIXED_ITERS, L.FLAG_MC_UNFUSED | L.FLAG_KERNEL_GENERIC):
            t2 = L.cc_sample_decode_tally(graph, c["Lx"], p, 20260206, 0, N, max_iter=50, flags=fl)
            assert np.array_equal(t2, ref), (tag, fl, t2.tolist(), ref.tolist())
        # without OSD the BP failures are judged as they are (and counted unsatisfied)
        r0 = oracle.cc_sample_decode_tally(ip, ix, n, c["Lx"], p, 20260206, 0, N, max_iter=50, use_osd=False, threads=0)
        for fl in (0, L.FLAG_MC_UNFUSED):
            t0 = L.cc_sample_decode_tally(graph, c["Lx"], p, 20260206, 0, N, max_iter=50, use_osd=False, flags=fl)
            assert np.array_equal(t0, r0), (tag, fl, t0.tolist(), r0.tolist())
        assert ref[L.TALLY["unsat_z"]] == 0          # OSD-0 always reproduces a realisable syndrome
    assert ref[0] == 1000


def test_code_capacity_small_pieces_many_failures_per_piece(L, oracle):
    """Fresh plans whose every piece lists thousands of BP failures (p = 0.08, three iterations): the one-wave OSD-0 kernels then hand most records out
    through their ticket counter, from the FIRST launch of every lane on.  Round 4's soak found that counter zeroed in the middle of such a first launch
    (a bare hipMemset is queued on the null stream, which the lanes' non-blocking streams do not wait for): records solved and judged twice, a
    logical-error count a few hundred too high, only with the lanes on hardware queues of their own.  Several fresh plans, shots split over calls."""
    from qldpc_amd.data import load_code
    c = load_code("bb72")
    ip, ix, n = c["Hx_indptr"], c["Hx_indices"], int(c["n"])
    graph = L.Graph(ip, ix, n)
    seed = 2982864118342514790
    for flags in (L.FLAG_KERNEL_GENERIC, L.FLAG_FIXED_ITERS, 0):
        for batch, runs in ((4096, (81959,)), (4096, (100, 39900, 42000)), (40000, (120000,))):
            plan = L.CodeCapacityPlan(graph, c["Lx"], 0.08, max_iter=3, use_osd=True, flags=flags, batch=batch)
            want, begin = np.zeros(16, np.int64), 1000
            for count in runs:
                plan.run(seed, begin, count)
                want += oracle.cc_sample_decode_tally(ip, ix, n, c["Lx"], 0.08, seed, begin, count, max_iter=3, use_osd=True, threads=0)
                begin += count
            got = plan.read()
            plan.close()
            assert np.array_equal(got, want), (flags, batch, runs, got.tolist(), want.tolist())


@pytest.fixture
def options(L):
    """qldpc_set_option switches are process-wide: put the defaults back after a test that turns them"""
    yield L.set_option
    for name, v in (("mc_first_iteration", 1), ("mc_first_bits", 8), ("mc_tail_overlap", 1), ("osd_presort", -1)):
        L.set_option(name, v)


def test_first_iteration_pipeline_equals_full_decoder_and_oracle(L, oracle, options):
    """Reference semantics through the bit-sliced first iteration (csrc/mc_first.hip) + the full decoder on the shots it lists
    == the full decoder on every shot == the oracle: identical tallies for every code, error rate, iteration cap and ragged batch."""
    from qldpc_amd.data import load_code
    for tag in ("bb72", "bb144", "bb288", "bb90"):
        c = load_code(tag)
        ip, ix, n = c["Hx_indptr"], c["Hx_indices"], c["n"]
        graph = L.Graph(ip, ix, n)
        for p, mi, count, begin, osd in ((0.005, 50, 20011, 12345, True), (0.03, 50, 9000, 7, True), (0.08, 7, 5003, 0, True), (0.005, 1, 8000, 99, True),
                                         (0.02, 2, 6000, 5, False), (0.3, 30, 700, 1, True), (0.01, 50, 1, 3, True), (0.01, 50, 511, 3, True)):
            ref = oracle.cc_sample_decode_tally(ip, ix, n, c["Lx"], p, 4242, begin, count, max_iter=mi, use_osd=osd, threads=0)
            options("mc_first_iteration", 0)
            full = L.cc_sample_decode_tally(graph, c["Lx"], p, 4242, begin, count, max_iter=mi, use_osd=osd)
            assert np.array_equal(full, ref), (tag, p, mi, full.tolist(), ref.tolist())
            options("mc_first_iteration", 1)
            for bits in (8, 16, 32):
                options("mc_first_bits", bits)
                got = L.cc_sample_decode_tally(graph, c["Lx"], p, 4242, begin, count, max_iter=mi, use_osd=osd)
                assert np.array_equal(got, ref), (tag, p, mi, bits, got.tolist(), ref.tolist())
    # a plan over several batches, the last one ragged; and a code the shortcut does not apply to (Steane: irregular -> unfused pipeline)
    c = load_code("bb144")
    graph = L.Graph(c["Hx_indptr"], c["Hx_indices"], c["n"])
    ref = oracle.cc_sample_decode_tally(c["Hx_indptr"], c["Hx_indices"], c["n"], c["Lx"], 0.01, 5, 100, 10000, max_iter=50, threads=0)
    for granule in (0, 262144):        # the batch taken literally (three pieces on the plan's streams), and under a launch granule of this plan
        plan = L.CodeCapacityPlan(graph, c["Lx"], 0.01, max_iter=50, batch=4096, min_launch=granule)
        plan.run(5, 100, 10000)
        assert np.array_equal(plan.read(), ref)
        plan.close()
    c = load_code("steane")
    graph = L.Graph(c["Hx_indptr"], c["Hx_indices"], c["n"])
    ref = oracle.cc_sample_decode_tally(c["Hx_indptr"], c["Hx_indices"], c["n"], c["Lx"], 0.05, 5, 0, 3000, max_iter=50, threads=0)
    assert np.array_equal(L.cc_sample_decode_tally(graph, c["Lx"], 0.05, 5, 0, 3000, max_iter=50), ref)


def test_wave_private_kernel_matches_golden_and_team_kernel(LX, golden, oracle):
    L = LX
    options = L.set_option
    """csrc/minsum_wave.hip (option regular_kernel = 2): a team of lanes of one wave per shot, no workgroup barrier.  Bit-identical to the
    reference fixtures in decode mode (clean inputs; everything else must still take the team kernel) and identical tallies in the fused
    Monte-Carlo mode, fixed-work and early-exit, for every instantiated shape."""
    from qldpc_amd.data import load_code
    options("regular_kernel", 2)
    for tag in ("bb72", "bb144", "bb288"):
        g = golden(tag + "_minsum")
        n = int(g["Hx_shape"][1])
        graph = L.Graph(g["Hx_indptr"], g["Hx_indices"], n)
        for cpl in (0, 4, 5, 6, 9):
            options("wave_cpl", cpl)
            for p in ("p005", "p030", "p080"):
                base = f"Hx_{p}"
                synd, prior = g[base + "_syndromes"], g[base + "_prior"]
                for mi in (1, 5, 50):
                    check(decode(L, graph, synd, prior, mi, 0), g, f"{base}_dyn_it{mi}")
                check(decode(L, graph, synd, prior, 50, L.FLAG_FIXED_ITERS), g, f"{base}_dyn_it50")
                for v, kw in VARIANT_KW.items():          # damping / tight clips: not eligible, must fall through to the team kernel unchanged
                    check(decode(L, graph, synd, prior, 30, 0, **kw), g, f"{base}_{v}")
        options("wave_cpl", 0)
        c = load_code(tag)
        for p, N in ((0.005, 20000), (0.04, 9001)):
            ref = oracle.cc_sample_decode_tally(c["Hx_indptr"], c["Hx_indices"], n, c["Lx"], p, 77, 1000, N, max_iter=50, threads=0)
            for fl in (0, L.FLAG_FIXED_ITERS):
                for rst in (6, 7):
                    options("wave_rst", rst)
                    got = L.cc_sample_decode_tally(graph, c["Lx"], p, 77, 1000, N, max_iter=50, flags=fl)
                    assert np.array_equal(got, ref), (tag, p, fl, rst, got.tolist(), ref.tolist())
            options("wave_rst", 0)
    # a non-uniform prior below the clip (general iteration 0) and one above it (not eligible)
    c = load_code("bb144")
    graph = L.Graph(c["Hx_indptr"], c["Hx_indices"], c["n"])
    rng = np.random.default_rng(11)
    errs = (rng.random((500, c["n"])) < 0.05).astype(np.int8)
    synd = np.array([oracle.syndrome_check(c["Hx_indptr"], c["Hx_indices"], e) for e in errs])
    for scale in (1.0, 9.0):
        prior = np.full(c["n"], np.log(0.95 / 0.05)) * scale
        prior[::5] *= 0.8
        ref = oracle.minsum_decode_batch(c["Hx_indptr"], c["Hx_indices"], c["n"], synd, prior, max_iter=20)
        for fl in (0, L.FLAG_FIXED_ITERS):
            for a, b in zip(decode(L, graph, synd, prior, 20, fl), ref):
                assert np.array_equal(a, b, equal_nan=True)
    for name in ("regular_kernel", "wave_cpl", "wave_rst", "wave_grid"):
        options(name, 0)


def test_product_library_refuses_the_experiments(L, oracle):
    """Measured-and-rejected kernels live in libqldpc_hip_experiments.so only: the product library answers their flags and options with
    QLDPC_ERR_UNSUPPORTED instead of silently running something else."""
    from qldpc_amd.data import load_code, load_circuit_matrices
    d = load_circuit_matrices("circ72")
    n = int(d["HdecZ_shape"][1])
    graph = L.Graph(d["HdecZ_indptr"], d["HdecZ_indices"], n)
    synd = np.zeros((2, int(d["HdecZ_shape"][0])), np.int8)
    prior = np.full(n, 3.0)
    for fl in (L.FLAG_WG_EDGE_LANES, L.FLAG_WG_IDXLOAD):
        with pytest.raises(L.QldpcError, match="experiment"):
            L.minsum_decode_batch(graph, synd, prior, 5, "dynamical", 1.0, flags=fl)
    for fl in (L.FLAG_OSD_QUEUE, L.FLAG_OSD_QUEUE | L.FLAG_OSD_LDS):
        with pytest.raises(L.QldpcError, match="experiment"):
            L.osd0_batch(graph, synd, np.ones((2, n)), np.zeros((2, n), np.int8), flags=fl)
    with pytest.raises(L.QldpcError, match="experiment"):
        L.set_option("regular_kernel", 2)
    L.set_option("regular_kernel", 0)
    with pytest.raises(L.QldpcError, match="unknown option"):
        L.set_option("no_such_switch", 1)


def test_philox_known_answer(L, oracle):
    import ctypes as C
    for ctr, key in (([0, 0, 0, 0], [0, 0]), ([0xffffffff] * 4, [0xffffffff] * 2), ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])):
        c = np.array(ctr, np.uint32); k = np.array(key, np.uint32); o = np.zeros(4, np.uint32)
        L.lib().qldpc_philox4x32_10(L.ptr(c, C.c_uint32), L.ptr(k, C.c_uint32), L.ptr(o, C.c_uint32))
        assert np.array_equal(o, oracle.philox(c, k))


def _circuit_setup(L, oracle, tag, golden):
    from qldpc_amd.data import load_circuit_matrices
    g = golden(tag + "_noise")
    d = load_circuit_matrices(tag)
    circ = oracle.make_circuit(g, g["Lx"], g["Lz"])
    secs, graphs, priors, masks = [], [], [], []
    for s in "ZX":
        n = int(d[f"Hdec{s}_shape"][1])
        prior = oracle.prior_llrs(d[f"channel_probs{s}"])
        secs.append(oracle.make_sector(d[f"Hdec{s}_indptr"], d[f"Hdec{s}_indices"], n, prior, d[f"H{s}_logical_indptr"], d[f"H{s}_logical_indices"]))
        graphs.append(L.Graph(d[f"Hdec{s}_indptr"], d[f"Hdec{s}_indices"], n))
        priors.append(prior)
        masks.append(L.logical_column_masks((d[f"H{s}_logical_indptr"], d[f"H{s}_logical_indices"]), n))
    return g, circ, secs, graphs, priors, masks


def test_osd0_key_distributions_at_production_size(L, oracle):
    """The device's own |llr| order at 1008 x 8785 on key distributions chosen against the sorted head (sample-based bound, split, value buckets, the rest on demand):
    every key equal; two values; a distinct head below a 3000-long run of equal keys; a run of equal keys that STRADDLES the bound; nearly everything infinite;
    keys descending with the column index; a head squeezed against the bound (lopsided buckets).  Realisable and unrealisable syndromes; against the oracle."""
    from qldpc_amd.data import load_circuit_matrices
    d = load_circuit_matrices("circ144")
    rng = np.random.default_rng(7)
    for s in "ZX":
        ip, ix = d[f"Hdec{s}_indptr"], d[f"Hdec{s}_indices"]
        m, n = (int(v) for v in d[f"Hdec{s}_shape"])
        graph = L.Graph(ip, ix, n)
        cases = []
        cases.append(np.full(n, 2.5))
        two = np.where(rng.random(n) < 0.1, 0.75, 6.0); cases.append(two)
        a = np.full(n, 9.0); idx = rng.permutation(n); a[idx[:1300]] = rng.random(1300) * 3.0; a[idx[4300:]] = 9.0 + rng.random(n - 4300) * 5.0; cases.append(a)
        b = 20.0 + rng.random(n) * 5.0; idx = rng.permutation(n); b[idx[:600]] = rng.random(600); b[idx[600:2600]] = 4.0; cases.append(b)
        c = np.full(n, np.inf); c[rng.permutation(n)[:100]] = rng.random(100) * 7.0; cases.append(c)
        cases.append(np.arange(n, 0, -1) * 1e-3)
        e = 10.0 - rng.random(n) ** 8 * 1e-6; e[rng.permutation(n)[:50]] = rng.random(50); cases.append(e)
        llr = np.stack(cases) * np.where(rng.random((len(cases), n)) < 0.5, -1.0, 1.0)
        B = llr.shape[0]
        err = (rng.random((B, n)) < 0.01).astype(np.int8)
        synd = np.stack([oracle.syndrome_check(ip, ix, e_) for e_ in err])
        hard = (rng.random((B, n)) < 0.002).astype(np.int8)
        for unreal in (False, True):
            sy = synd.copy()
            if unreal:
                sy[:, :7] ^= 1
            want = np.stack([oracle.osd0(ip, ix, n, sy[i], llr[i], hard[i]) for i in range(B)])
            got = L.osd0_batch(graph, sy, llr, hard)
            assert np.array_equal(got, want), (s, unreal, np.flatnonzero((got != want).any(1)))


@pytest.mark.parametrize("tag,ntrial", [("circ72", 96), ("circ144", 24)])
def test_circuit_sampler_equals_literal_simulation(L, oracle, golden, tag, ntrial):
    """Signature-XOR sampler on the GPU == the oracle's literal noisy-circuit simulation on the same Philox draws (a10-a13)."""
    g, circ, secs, graphs, priors, masks = _circuit_setup(L, oracle, tag, golden)
    for p in (0.005, 0.05):
        plan = L.CircuitPlan(g, g["Lx"], g["Lz"], graphs[0], graphs[1], priors[0], priors[1], masks[0], masks[1], p, batch=64)
        spz, tz, spx, tx = plan.sample(4242, 1000, ntrial)          # more trials than one batch -> exercises chunking
        for t in range(ntrial):
            a, b, c, d = oracle.circuit_sample(circ, p, 4242, 1000 + t)
            assert np.array_equal(spz[t], a) and np.array_equal(tz[t], b) and np.array_equal(spx[t], c) and np.array_equal(tx[t], d), (tag, p, t)
        assert spz.any() and spx.any() and tz.any()
        plan.close()


@pytest.mark.parametrize("tag,ntrial", [("circ72", 200), ("circ144", 40)])
def test_circuit_level_tally_matches_oracle(L, oracle, golden, tag, ntrial):
    g, circ, secs, graphs, priors, masks = _circuit_setup(L, oracle, tag, golden)
    p = 0.005
    ref = oracle.circuit_sample_decode_tally(circ, secs[0], secs[1], p, 77, 0, ntrial, max_iter=50, threads=0)
    plan = L.CircuitPlan(g, g["Lx"], g["Lz"], graphs[0], graphs[1], priors[0], priors[1], masks[0], masks[1], p, max_iter=50, batch=64)
    plan.run(77, 0, ntrial // 2)
    plan.run(77, ntrial // 2, ntrial - ntrial // 2)              # split invariance
    t = plan.read()
    assert np.array_equal(t, ref), (t.tolist(), ref.tolist())
    assert ref[L.TALLY["osd_z"]] > 0 and ref[L.TALLY["unsat_z"]] == 0 and ref[L.TALLY["total_err"]] > 0
    plan.close()
    # without OSD-0 the BP failures are judged as they are: the judge's syndrome check (H @ det, from the ones of the correction) has something to find
    ref0 = oracle.circuit_sample_decode_tally(circ, secs[0], secs[1], p, 77, 0, ntrial, max_iter=50, use_osd=False, threads=0)
    plan0 = L.CircuitPlan(g, g["Lx"], g["Lz"], graphs[0], graphs[1], priors[0], priors[1], masks[0], masks[1], p, max_iter=50, use_osd=False, batch=64)
    plan0.run(77, 0, ntrial)
    t0 = plan0.read()
    plan0.close()
    assert np.array_equal(t0, ref0), (t0.tolist(), ref0.tolist())
    assert ref0[L.TALLY["unsat_z"]] > 0 and ref0[L.TALLY["unsat_x"]] > 0


def _two_alpha_tally(oracle, circ, secs, seed, count, max_iter, mode, alpha_z, alpha_x):
    """Oracle tally when the two sectors use different normalisation factors (the oracle entry point takes one): per-trial runs with
    each factor, Z slots from the first, X slots from the second, total = trials where either sector fails."""
    A = np.stack([oracle.circuit_sample_decode_tally(circ, secs[0], secs[1], 0.005, seed, i, 1, max_iter=max_iter, alpha=alpha_z, alpha_mode=mode, threads=1)
                  for i in range(count)])
    B = np.stack([oracle.circuit_sample_decode_tally(circ, secs[0], secs[1], 0.005, seed, i, 1, max_iter=max_iter, alpha=alpha_x, alpha_mode=mode, threads=1)
                  for i in range(count)])
    t = np.zeros(16, np.int64)
    t[0] = count
    for zslot in (1, 4, 6, 8, 10, 12):
        t[zslot] = A[:, zslot].sum()
        t[zslot + 1] = B[:, zslot + 1].sum()
    t[3] = int(np.count_nonzero((A[:, 1] != 0) | (B[:, 2] != 0)))
    return t


def test_run_simulation_mirror(L, oracle, golden):
    """run_simulation (engine.py:193-488) on [[72,12,6]] x 6 cycles against the oracle tally for the same trial stream."""
    from qldpc_amd.data import load_code, load_precomputed_matrices
    from qldpc_amd.simulation.engine import run_simulation
    c = load_code("bb72")
    g, circ, secs, graphs, priors, masks = _circuit_setup(L, oracle, "circ72", golden)
    bb = dict(ell=c["ell"], m=c["m_dim"], a_x_powers=c["a_x_powers"], a_y_powers=c["a_y_powers"], b_y_powers=c["b_y_powers"], b_x_powers=c["b_x_powers"])
    res = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=150, num_cycles=6, maxIter=50, osd_order=0,
                         precomputed_matrices=load_precomputed_matrices("circ72"), base_seed=31337, batch=64, **bb)
    ref = oracle.circuit_sample_decode_tally(circ, secs[0], secs[1], 0.005, 31337, 0, 150, max_iter=50, threads=0)
    assert np.array_equal(res["tally"], ref)
    assert res["num_trials"] == 150 and res["logical_errors"] == ref[3] and abs(res["logical_error_rate"] - ref[3] / 150) < 1e-12
    assert set(["logical_error_rate", "z_logical_error_rate", "x_logical_error_rate", "num_trials", "logical_errors"]) <= set(res)
    # without precomputed matrices the builder runs first and must lead to the identical tally
    res2 = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=150, num_cycles=6, maxIter=50, base_seed=31337, batch=64, **bb)
    assert np.array_equal(res2["tally"], ref)
    # osd_order=2 (main.py:44): the reference returns the OSD-0 solution whenever it satisfies the syndrome (osd.py:27-29) -> same tally
    res3 = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=150, num_cycles=6, maxIter=50, osd_order=2,
                          precomputed_matrices=load_precomputed_matrices("circ72"), base_seed=31337, batch=64, **bb)
    assert np.array_equal(res3["tally"], ref)
    # estimated normalisation factors (the mode main.py:48 selects) + SCOPT beta: the estimators draw from default_rng(base_seed), Z first
    from qldpc_amd.decoding.alpha import estimate_alpha_alvarado
    from qldpc_amd.decoding.scopt import estimate_scopt_beta
    res4 = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=100, num_cycles=6, maxIter=50, alpha_mode="alvarado", scopt=True,
                          alpha_estimation_trials=300, precomputed_matrices=load_precomputed_matrices("circ72"), base_seed=4242, batch=64, **bb)
    rng = np.random.default_rng(4242)
    az, r2z = estimate_alpha_alvarado(graphs[0], 0.005, trials=300, rng=rng, llrs=priors[0])
    ax, r2x = estimate_alpha_alvarado(graphs[1], 0.005, trials=300, rng=rng, llrs=priors[1])
    assert res4["alpha_r2_z"] == r2z and res4["alpha_r2_x"] == r2x and 0 < az < 1.5 and 0 < ax < 1.5
    bz, _ = estimate_scopt_beta(graphs[0], 0.005, trials=500, alpha=az, alpha_mode="alvarado", maxIter=50, rng=rng, llrs=priors[0])
    assert res4["beta_z"] == bz and set(["beta_x", "beta_r2_z", "beta_r2_x"]) <= set(res4)
    assert np.array_equal(res4["tally"], _two_alpha_tally(oracle, circ, secs, 4242, 100, 50, "alvarado", az, ax))
    res5 = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=60, num_cycles=6, maxIter=4, alpha_mode="alvarado-autoregressive",
                          alpha_estimation_trials=100, precomputed_matrices=load_precomputed_matrices("circ72"), base_seed=99, batch=64, **bb)
    assert res5["alpha_values_z"].shape == (4,) and res5["alpha_r2_values_x"].shape == (4,) and res5["num_trials"] == 60
    assert np.array_equal(res5["tally"], _two_alpha_tally(oracle, circ, secs, 99, 60, 4, "alvarado-autoregressive", res5["alpha_values_z"], res5["alpha_values_x"]))
    # in-order early stop (engine.py:441-464): the run ends AT the trial that brings the error count to the target
    per_trial = np.stack([oracle.circuit_sample_decode_tally(circ, secs[0], secs[1], 0.005, 31337, i, 1, max_iter=50, threads=1) for i in range(150)])
    bad = np.flatnonzero(per_trial[:, 3])
    assert bad.size >= 4
    for target in (1, 3, bad.size, bad.size + 5):
        r = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=150, num_cycles=6, maxIter=50,
                           precomputed_matrices=load_precomputed_matrices("circ72"), base_seed=31337, batch=32,
                           target_logical_errors=target, **bb)
        stop = int(bad[target - 1]) + 1 if target <= bad.size else 150
        assert r["num_trials"] == stop and r["logical_errors"] == int(per_trial[:stop, 3].sum())
        assert r["z_logical_error_rate"] == per_trial[:stop, 1].sum() / stop and r["x_logical_error_rate"] == per_trial[:stop, 2].sum() / stop


@pytest.mark.parametrize("tag,code,cycles", [("circ72", "bb72", 6), ("circ144", "bb144", 12), ("circ90", "bb90", 10), ("circ108", "bb108", 10),
                                             ("circ288", "bb288", 18)])
def test_builder_reproduces_reference_matrix_cache(L, tag, code, cycles):
    """build_decoding_matrices (builder.py:69-176) against the matrices the REFERENCE cached (matrix_cache/*.npz, re-packed as CSR):
    same columns in the same order, bit-identical summed probabilities, same logical rows."""
    from qldpc_amd.data import load_code, load_circuit_matrices
    from qldpc_amd.codes.bb_code import BBCodeCircuit
    from qldpc_amd.noise.builder import build_decoding_matrices
    c = load_code(code)
    d = load_circuit_matrices(tag)
    cb = BBCodeCircuit(c["Hx"], c["Hz"], num_cycles=cycles, ell=c["ell"], m=c["m_dim"], a_x_powers=c["a_x_powers"], a_y_powers=c["a_y_powers"],
                       b_y_powers=c["b_y_powers"], b_x_powers=c["b_x_powers"])
    M = build_decoding_matrices(cb, c["Lx"], c["Lz"], 0.005, verbose=False)
    k = int(c["Lx"].shape[0])
    assert M["k"] == k and M["num_cycles"] == cycles
    for s in "ZX":
        ip, ix, shape = L.canonical_csr(M[f"Hdec{s}"])
        assert tuple(shape) == tuple(int(x) for x in d[f"Hdec{s}_shape"])
        assert np.array_equal(ip, d[f"Hdec{s}_indptr"]) and np.array_equal(ix, d[f"Hdec{s}_indices"])
        assert np.array_equal(M[f"channel_probs{s}"], d[f"channel_probs{s}"])          # bitwise: same summation order
        flr = M[f"first_logical_row{s}"]
        lip, lix, _ = L.canonical_csr(M[f"H{s}_full"][flr:flr + k])
        assert np.array_equal(lip, d[f"H{s}_logical_indptr"]) and np.array_equal(lix, d[f"H{s}_logical_indices"])


def test_api_edge_cases(L, oracle):
    """Empty / degenerate inputs through every batched entry point; results against the oracle where one exists."""
    import ctypes as C
    from qldpc_amd.data import load_code
    c = load_code("bb72")
    ip, ix, n, m = c["Hx_indptr"], c["Hx_indices"], c["n"], c["m"]
    g = L.Graph(ip, ix, n)
    lib = L.lib()
    # B = 0 everywhere
    z8 = np.zeros(0, np.int8); zf = np.zeros(0)
    L.check(lib.qldpc_gf2_spmv_batch(g.handle, C.c_int64(0), L.ptr(z8, C.c_int8), L.ptr(z8, C.c_int8)))
    L.check(lib.qldpc_osd0_batch(g.handle, C.c_int64(0), L.ptr(z8, C.c_int8), L.ptr(zf, C.c_double), L.ptr(z8, C.c_int8), None, 0, L.ptr(z8, C.c_int8)))
    L.check(lib.qldpc_minsum_check_pass(g.handle, C.c_int64(0), L.ptr(zf, C.c_double), L.ptr(zf, C.c_double), C.c_double(1.0), L.ptr(zf, C.c_double), L.ptr(zf, C.c_double)))
    assert np.array_equal(L.cc_sample_decode_tally(g, c["Lx"], 0.01, 1, 0, 0), np.zeros(16, np.int64))
    # max_iter = 0: nothing is decoded; final_iter = -1, not converged (kernels.py:267-268 with an empty loop)
    synd = np.zeros((5, m), np.int8); synd[1, 3] = 1
    prior = np.full(n, 2.0)
    for fl in (0, L.FLAG_KERNEL_GENERIC | L.FLAG_KERNEL_RESIDENT, L.FLAG_KERNEL_STREAM):
        e, cv, v, it = L.minsum_decode_batch(g, synd, prior, 0, "dynamical", 1.0, flags=fl)
        assert not e.any() and not cv.any() and (it == -1).all() and not v.any()
    # negative, zero and mixed priors (not "clean": a -0.0 entry) must still match the oracle bit for bit on every kernel
    rng = np.random.default_rng(3)
    pri = rng.normal(0.5, 2.0, n); pri[5] = 0.0; pri[6] = -0.0; pri[7] = -3.0
    synd = (rng.random((65, m)) < 0.2).astype(np.int8)
    ref = oracle.minsum_decode_batch(ip, ix, n, synd, pri, max_iter=9, threads=0)
    for fl in (0, L.FLAG_KERNEL_GENERIC | L.FLAG_KERNEL_RESIDENT, L.FLAG_KERNEL_STREAM):
        for a, b in zip(L.minsum_decode_batch(g, synd, pri, 9, "dynamical", 1.0, flags=fl), ref):
            assert np.array_equal(a, b, equal_nan=True)
    # a graph without any check: every shot "converges" at iteration 0 with values = prior (kernels.py:319-364)
    g0 = L.Graph(np.zeros(1, np.int32), np.zeros(0, np.int32), 4)
    e, cv, v, it = L.minsum_decode_batch(g0, np.zeros((3, 0), np.int8), np.array([1.0, -2.0, 0.0, 3.0]), 5, "dynamical", 1.0)
    assert cv.all() and (it == 0).all() and np.array_equal(v, np.tile([1.0, -2.0, 0.0, 3.0], (3, 1))) and np.array_equal(e[0], [0, 1, 0, 0])


def test_device_pointer_entry_point(L, oracle):
    """qldpc_minsum_decode_batch_dev on torch CUDA tensors and a non-default stream (the _dev ABI never touches host memory)."""
    import ctypes as C
    import torch
    from qldpc_amd.data import load_code
    c = load_code("bb144")
    ip, ix, n, m = c["Hx_indptr"], c["Hx_indices"], c["n"], c["m"]
    g = L.Graph(ip, ix, n)
    rng = np.random.default_rng(9)
    B = 1000
    synd = (rng.random((B, m)) < 0.08).astype(np.int8)
    prior = np.full(n, np.log(0.97 / 0.03))
    ref = oracle.minsum_decode_batch(ip, ix, n, synd, prior, max_iter=20, threads=0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ds = torch.from_numpy(synd).cuda(non_blocking=False); dp = torch.from_numpy(prior).cuda()
        de = torch.empty((B, n), dtype=torch.int8, device="cuda"); dl = torch.empty((B, n), dtype=torch.float64, device="cuda")
        dc = torch.empty(B, dtype=torch.uint8, device="cuda"); di = torch.empty(B, dtype=torch.int32, device="cuda")
        st.synchronize()
        L.check(L.lib().qldpc_minsum_decode_batch_dev(g.handle, C.c_int64(B), C.c_void_p(ds.data_ptr()), C.c_void_p(dp.data_ptr()), C.c_int(20),
                                                      C.c_int(L.ALPHA_DYNAMIC), C.c_double(1.0), None, C.c_int(0), C.c_double(1.0), C.c_double(20.0),
                                                      C.c_int(0), C.c_void_p(de.data_ptr()), C.c_void_p(dl.data_ptr()), C.c_void_p(dc.data_ptr()),
                                                      C.c_void_p(di.data_ptr()), C.c_void_p(st.cuda_stream)))
        st.synchronize()
    for a, b in zip((de.cpu().numpy(), dc.cpu().numpy(), dl.cpu().numpy(), di.cpu().numpy()), ref):
        assert np.array_equal(a, b, equal_nan=True)
    # host-pointer form without the posteriors (out_llr == NULL): same decisions, 9x fewer result bytes over PCIe
    e2, c2, l2, i2 = L.minsum_decode_batch(g, synd, prior, 20, "dynamical", 1.0, want_llr=False)
    assert l2 is None and np.array_equal(e2, ref[0]) and np.array_equal(c2, ref[1]) and np.array_equal(i2, ref[3])


def test_estimators_golden(L, golden):
    """f4: estimate_alpha_alvarado / _autoregressive / estimate_scopt_beta on the GPU against the values the REFERENCE produced from
    the same error patterns (tests/golden/estimators.npz): identical histograms, factors within 1e-9 (scipy's fit on equal inputs)."""
    from conftest import ReplayRng, estimator_cases
    from qldpc_amd.decoding.alpha import estimate_alpha_alvarado, estimate_alpha_alvarado_autoregressive
    from qldpc_amd.decoding.scopt import estimate_scopt_beta
    from qldpc_amd.decoding._fit import class_densities
    from scipy.sparse import csr_matrix
    for c in estimator_cases(golden("estimators")):
        m = len(c["indptr"]) - 1
        H = csr_matrix((np.ones(len(c["indices"]), np.int8), c["indices"], c["indptr"]), shape=(m, c["n"]))
        g = L.graph_for(*L.canonical_csr(H)[:2], c["n"])
        if c["kind"] == "alvarado":
            a, r2 = estimate_alpha_alvarado(H, c["p"], trials=c["trials"], bins=c["bins"], rng=ReplayRng(c["errors"]), llrs=c["prior"])
            assert abs(a - c["out"]["alpha"]) <= 1e-9 and abs(r2 - c["out"]["r2"]) <= 1e-9, c["name"]
            st = L.MessageStats(g, c["errors"], c["prior"], L.STATS_CHECK_MESSAGES, 0)
            f0, f1, edges = class_densities(st, c["bins"], "alpha")
            st.close()
            assert np.array_equal(f0, c["hist"][0]) and np.array_equal(f1, c["hist"][1]) and np.array_equal(edges, c["edges"][0])
        elif c["kind"] == "autoregressive":
            av, rv = estimate_alpha_alvarado_autoregressive(H, c["p"], maxIter=c["iters"], trials=c["trials"], bins=c["bins"], damping=c["damping"],
                                                            clip_llr=c["clip"], rng=ReplayRng(c["errors"]), llrs=c["prior"])
            assert av.shape == c["out"]["alpha"].shape and np.max(np.abs(av - c["out"]["alpha"])) <= 1e-9, (c["name"], av, c["out"]["alpha"])
            assert np.max(np.abs(rv - c["out"]["r2"])) <= 1e-9
            k = c["iters"] - 1                  # the deepest fit, fed with the reference's own previous factors
            E = c["errors"][k * c["trials"]:(k + 1) * c["trials"]]
            st = L.MessageStats(g, E, c["prior"], L.STATS_CHECK_MESSAGES, k, alpha_mode="alvarado-autoregressive", alpha=c["out"]["alpha"][:k],
                                damping=c["damping"], clip_llr=c["clip"])
            f0, f1, edges = class_densities(st, c["bins"], "alpha")
            st.close()
            assert np.array_equal(f0, c["hist"][2 * k]) and np.array_equal(f1, c["hist"][2 * k + 1]) and np.array_equal(edges, c["edges"][k])
        else:
            b, r2 = estimate_scopt_beta(H, c["p"], trials=c["trials"], bins=c["bins"], alpha=c["alpha"], alpha_mode=c["alpha_mode"], maxIter=c["iters"],
                                        damping=c["damping"], clip_llr=c["clip"], rng=ReplayRng(c["errors"]), llrs=c["prior"])
            assert abs(b - c["out"]["beta"]) <= 1e-9 and abs(r2 - c["out"]["r2"]) <= 1e-9, c["name"]


def test_estimator_statistics_vs_oracle(L, oracle):
    """f4 at a larger size: range, finite counts and both histograms of the device trial loops equal the oracle's samples binned by numpy
    (bit-exact counts), for check messages after 0 / 3 iterations and for posteriors; includes a degree-1 check (infinite messages)."""
    from qldpc_amd.data import load_code
    rng = np.random.default_rng(77)
    c = load_code("bb144")
    ip, ix, n = c["Hx_indptr"], c["Hx_indices"], int(c["n"])
    g = L.graph_for(ip, ix, n)
    q = rng.uniform(0.005, 0.09, n)
    prior = np.log((1 - q) / q)
    E = (rng.random((3000, n)) < 0.04).astype(np.int8)
    cols = np.asarray(ix)

    def check(stats, samples, bits, bins):
        fin = np.isfinite(samples)
        assert stats.finite == (int(np.count_nonzero(fin & (bits == 0))), int(np.count_nonzero(fin & (bits == 1))))
        assert stats.range == (samples[fin].min(), samples[fin].max())
        edges = np.histogram_bin_edges(np.zeros(0), bins=bins, range=stats.range)
        h0, h1 = stats.histogram(edges)
        assert np.array_equal(h0, np.histogram(samples[fin & (bits == 0)], bins=bins, range=stats.range)[0])
        assert np.array_equal(h1, np.histogram(samples[fin & (bits == 1)], bins=bins, range=stats.range)[0])
        stats.close()

    for prev, damping, clip in (((), 1.0, 20.0), ((0.6, 0.75, 0.85), 0.8, 7.5)):
        R = oracle.alpha_messages(ip, ix, n, E, prior, alpha_prev=prev, damping=damping, clip_llr=clip)
        mode, alpha = ("alvarado-autoregressive", np.array(prev)) if prev else ("dynamical", 1.0)
        check(L.MessageStats(g, E, prior, L.STATS_CHECK_MESSAGES, len(prev), alpha_mode=mode, alpha=alpha, damping=damping, clip_llr=clip),
              R.ravel(), E[:, cols].ravel(), 64)
    for mode, alpha, iters in (("dynamical", 1.0, 30), ("alvarado", 0.75, 10), ("alvarado-autoregressive", np.array([0.6, 0.8]), 7)):
        V = oracle.scopt_values(ip, ix, n, E, prior, max_iter=iters, alpha=alpha, alpha_mode=mode)
        check(L.MessageStats(g, E, prior, L.STATS_POSTERIOR, iters, alpha_mode=mode, alpha=alpha), V.ravel(), E.ravel(), 50)
    # a graph with a degree-1 check: its message is +-inf (min over an empty set) and must be dropped like np.isfinite does
    ip2 = np.array([0, 1, 3, 6], np.int32); ix2 = np.array([0, 0, 1, 1, 2, 3], np.int32)
    g2 = L.Graph(ip2, ix2, 4)
    E2 = (rng.random((500, 4)) < 0.2).astype(np.int8)
    pr2 = np.array([1.5, 2.5, 0.5, 3.0])
    R2 = oracle.alpha_messages(ip2, ix2, 4, E2, pr2)
    assert np.isinf(R2).any()
    check(L.MessageStats(g2, E2, pr2, L.STATS_CHECK_MESSAGES, 0), R2.ravel(), E2[:, ix2].ravel(), 10)
    # argument errors mirror the reference
    with pytest.raises(ValueError):
        from qldpc_amd.decoding.alpha import estimate_alpha_alvarado
        estimate_alpha_alvarado(np.eye(3, dtype=np.int8), 0.7, llrs=np.ones(3))


def test_osd_order_w_golden(L, golden, oracle):
    """f1: performOSD_enhanced with order > 0 on the GPU against the solutions the REFERENCE returned (tests/golden/osdw.npz: syndromes
    OSD-0 cannot satisfy, so the flip-set sweep of osd.py:31-75 runs; 18 cases end on a swept candidate), then a batch against the oracle."""
    from conftest import osdw_cases
    from qldpc_amd.decoding.osd import performOSD_enhanced
    cases = osdw_cases(golden("osdw"))
    for c in cases:
        Hd = np.asarray(c["H"].todense(), dtype=np.float64)
        for ordering in (c["ordering"], None):
            sol = performOSD_enhanced(Hd, c["syndrome"], c["llr"], c["hard"], order=c["order"], max_combinations=c["maxc"], ordering=ordering)
            assert sol.dtype == np.int64 and np.array_equal(sol, c["solution"]), (c["name"], c["order"], c["maxc"])
    assert np.array_equal(performOSD_enhanced(Hd, c["syndrome"], c["llr"], c["hard"], order=-3), performOSD_enhanced(Hd, c["syndrome"], c["llr"], c["hard"]))
    # batch entry point vs the oracle: [[72,12,6]] Hx (rank 30 of 36 rows) with random syndromes, and a wide random matrix
    import ctypes as C
    rng = np.random.default_rng(2)
    from qldpc_amd.data import load_code
    code = load_code("bb72")
    wide = (rng.random((7, 11)) < 0.35).astype(np.int8); wide[3] = wide[1] ^ wide[2]; wide[6] = wide[0]
    total_changed = 0
    for ip, ix, n in ((code["Hx_indptr"], code["Hx_indices"], 72), L.canonical_csr(wide)[:2] + (11,)):
        g = L.graph_for(ip, ix, n)
        B, m = 48, len(ip) - 1
        synd = (rng.random((B, m)) < 0.5).astype(np.int8)
        synd[::7] = 0
        llr = rng.normal(0, 2.5, (B, n)); llr[3, :5] = 0.0; llr[4, 10] = np.inf
        hard = (rng.random((B, n)) < 0.15).astype(np.int8)
        for order, maxc in ((1, 0), (2, 0), (3, 50), (4, 0)):
            sol = np.zeros((B, n), np.int8)
            L.check(L.lib().qldpc_osdw_batch(g.handle, C.c_int64(B), L.ptr(synd, C.c_int8), L.ptr(llr, C.c_double), L.ptr(hard, C.c_int8), None,
                                             C.c_int(order), C.c_int64(maxc), L.ptr(sol, C.c_int8)))
            changed = 0
            for b in range(B):
                want = oracle.osdw(ip, ix, n, synd[b], llr[b], hard[b], order, maxc or None)
                assert np.array_equal(sol[b], want), (n, order, maxc, b)
                changed += not np.array_equal(want, oracle.osd0(ip, ix, n, synd[b], llr[b], hard[b]))
            total_changed += changed
    assert total_changed > 0                    # the sweep really replaced OSD-0 answers in the batch
    with pytest.raises(L.QldpcError):           # 106,761 flip sets per shot: refused without max_combinations
        performOSD_enhanced(np.asarray(code["Hx"], dtype=np.float64), np.ones(36, np.int8), rng.normal(0, 1, 72), np.zeros(72, np.int8), order=8)


def test_generic_fused_monte_carlo_equals_oracle(L, oracle, options):
    """The fused Monte-Carlo form of the irregular-degree resident kernel (csrc/minsum_resident.hip, MC = true: Steane = BASELINE config 1, or any
    small graph through FLAG_KERNEL_GENERIC) == the unfused sample / decode / judge launches == the oracle: identical tallies with and without
    OSD-0, under both iteration policies, over several pieces with a ragged tail."""
    from qldpc_amd.data import load_code
    for tag, flags in (("steane", 0), ("bb72", L.FLAG_KERNEL_GENERIC), ("bb144", L.FLAG_KERNEL_GENERIC), ("bb288", L.FLAG_KERNEL_GENERIC)):
        c = load_code(tag)
        ip, ix, n = c["Hx_indptr"], c["Hx_indices"], c["n"]
        graph = L.Graph(ip, ix, n)
        for p, mi, count, begin, osd in ((0.05, 50, 3000, 0, True), (0.02, 50, 5003, 77, False), (0.1, 3, 2500, 5, True), (0.3, 50, 700, 9, True)):
            ref = oracle.cc_sample_decode_tally(ip, ix, n, c["Lx"], p, 4242, begin, count, max_iter=mi, use_osd=osd, threads=0)
            for fl in (flags, flags | L.FLAG_FIXED_ITERS, flags | L.FLAG_MC_UNFUSED):
                plan = L.CodeCapacityPlan(graph, c["Lx"], p, max_iter=mi, use_osd=osd, flags=fl, batch=1024)
                plan.run(4242, begin, count)
                got = plan.read()
                plan.close()
                assert np.array_equal(got, ref), (tag, p, mi, osd, fl, got.tolist(), ref.tolist())


def test_full_size_properties(L, oracle, options):
    """BASELINE.json's own sizes, where the oracle would take minutes: size-independent properties of the Monte-Carlo tally.
      * additivity / order independence: the tally of [0, N) equals the sum over any split into sub-ranges, in any order, with any batch;
      * the fixed-work and the early-exit decoders (different kernels paths) agree on every counter;
      * every decode ends on a correction that reproduces its syndrome (unsat == 0 with OSD-0), zero-syndrome shots converge at once,
        counters are mutually consistent; an oracle-checked prefix anchors the stream itself."""
    from qldpc_amd.data import load_code
    T = L.TALLY
    # BASELINE configs 3, 2 (quoted at batch = 4096) and 4 (all three points of the p-sweep)
    for tag, p, N, batch in (("bb144", 0.005, 10_000_000, 1 << 20), ("bb72", 0.005, 1_000_000, 4096), ("bb288", 0.004, 10_000_000, 1 << 20),
                             ("bb288", 0.005, 10_000_000, 1 << 20), ("bb288", 0.006, 10_000_000, 1 << 20)):      # 1e7 shots per point, as BASELINE quotes config 4
        c = load_code(tag)
        graph = L.graph_for(c["Hx_indptr"], c["Hx_indices"], c["n"])
        plan = L.CodeCapacityPlan(graph, c["Lx"], p, max_iter=50, use_osd=True, batch=batch)      # config 2: its batch of 4096 taken literally, 245 pieces on the plan's eight streams
        plan.run(7, 0, N)
        whole = plan.read(clear=True)
        cuts = [0, 1, 4097, N // 3, N // 3 + 1_000_003 % N, N]
        cuts = sorted(set(min(x, N) for x in cuts))
        parts = np.zeros_like(whole)
        for a, b in reversed(list(zip(cuts[:-1], cuts[1:]))):          # sub-ranges in reverse order
            plan.run(7, a, b - a)
            parts += plan.read(clear=True)
        plan.close()
        assert np.array_equal(parts, whole), (tag, parts.tolist(), whole.tolist())
        fixed = L.CodeCapacityPlan(graph, c["Lx"], p, max_iter=50, use_osd=True, flags=L.FLAG_FIXED_ITERS, batch=1 << 19)
        fixed.run(7, 0, N)
        assert np.array_equal(fixed.read(), whole), tag
        fixed.close()
        assert whole[T["trials"]] == N and whole[T["unsat_z"]] == 0
        assert whole[T["bp_conv_z"]] + whole[T["osd_z"]] == N                      # every shot is either converged or post-processed
        assert whole[T["zero_synd_z"]] <= whole[T["bp_conv_z"]]                    # a zero syndrome converges (to the zero correction)
        assert N <= whole[T["iters_z"]] <= 50 * N and whole[T["z_err"]] == whole[T["total_err"]] <= whole[T["osd_z"]] + whole[T["bp_conv_z"]]
        expect0 = N * (1 - p) ** c["n"]                                            # P(no error at all) <= P(zero syndrome)
        assert whole[T["zero_synd_z"]] >= expect0 - 6 * np.sqrt(expect0)
        k = 50_000                                                                 # the same stream, prefix checked against the oracle
        ref = oracle.cc_sample_decode_tally(c["Hx_indptr"], c["Hx_indices"], c["n"], c["Lx"], p, 7, 0, k, max_iter=50, threads=0)
        assert np.array_equal(L.cc_sample_decode_tally(graph, c["Lx"], p, 7, 0, k, max_iter=50), ref)


def test_circuit_level_full_size_properties(L, oracle, golden):
    """Config 5 at a size the oracle cannot follow (100k trials, ~3 decodes each): additivity over splits and batch sizes, every OSD-0
    answer reproduces its syndrome, and the logical error rate sits inside the band of the reference's own results for this point
    (output/*/results.npz report 0.33-0.63 at p = 0.005)."""
    g, circ, secs, graphs, priors, masks = _circuit_setup(L, oracle, "circ144", golden)
    T = L.TALLY
    N = 100_000
    plan = L.CircuitPlan(g, g["Lx"], g["Lz"], graphs[0], graphs[1], priors[0], priors[1], masks[0], masks[1], 0.005, batch=16384)
    plan.run(11, 0, N)
    whole = plan.read(clear=True)
    plan.close()
    small = L.CircuitPlan(g, g["Lx"], g["Lz"], graphs[0], graphs[1], priors[0], priors[1], masks[0], masks[1], 0.005, batch=3000)
    parts = np.zeros_like(whole)
    for a, b in ((60_001, N), (0, 777), (777, 60_001)):
        small.run(11, a, b - a)
        parts += small.read(clear=True)
    outcomes = small.run_outcomes(11, 0, 5000)
    head = small.read(clear=True)
    small.close()
    assert np.array_equal(parts, whole), (parts.tolist(), whole.tolist())
    assert whole[T["trials"]] == N and whole[T["unsat_z"]] == 0 and whole[T["unsat_x"]] == 0
    assert whole[T["bp_conv_z"]] + whole[T["osd_z"]] == N and whole[T["bp_conv_x"]] + whole[T["osd_x"]] == N
    ler = whole[T["total_err"]] / N
    assert 0.33 <= ler <= 0.63, ler
    assert max(whole[T["z_err"]], whole[T["x_err"]]) <= whole[T["total_err"]] <= whole[T["z_err"]] + whole[T["x_err"]]
    # per-trial verdicts agree with the tally of the same range
    assert head[T["z_err"]] == np.count_nonzero(outcomes & 1) and head[T["x_err"]] == np.count_nonzero(outcomes & 2)
    assert head[T["total_err"]] == np.count_nonzero(outcomes)


def test_workgroup_kernel_with_posteriors_in_global_memory(L, oracle, golden, monkeypatch):
    """Graphs whose posteriors do not fit next to the check states in LDS keep V in HBM/L2 (minsum_wg_*<VG = true>).  Forced on the
    golden circuit-level cases (bit-identical LLRs), then on a matrix that really needs it: [[288,12,18]] x 12 cycles (2016 x ~17.5K)."""
    VG = L.FLAG_WG_VGLOBAL
    for tag in ("circ72", "circ144"):
        g = golden(tag + "_decode")
        from qldpc_amd.data import load_circuit_matrices
        d = load_circuit_matrices(tag)
        for s in "ZX":
            n = int(d[f"Hdec{s}_shape"][1])
            graph = L.Graph(d[f"Hdec{s}_indptr"], d[f"Hdec{s}_indices"], n)
            for flags in (VG, VG | L.FLAG_FIXED_ITERS, VG | L.FLAG_WG_ROWMAJOR):
                err, conv, llr, it = L.minsum_decode_batch(graph, g[f"{s}_syndromes"], g[f"llrs_{s}"], int(g["max_iter"]), "dynamical", 1.0, flags=flags)
                assert np.array_equal(err, g[f"{s}_err"]) and np.array_equal(conv.astype(bool), g[f"{s}_conv"].astype(bool))
                assert np.array_equal(llr, g[f"{s}_llr"]) and np.array_equal(it, g[f"{s}_iter"]), (tag, s, flags)
            # non-clean priors (a -0.0 and an infinity) take the generic kernel
            pr = g[f"llrs_{s}"].copy(); pr[3] = -0.0; pr[7] = np.inf
            e2, c2, l2, i2 = L.minsum_decode_batch(graph, g[f"{s}_syndromes"][:2], pr, 20, "dynamical", 1.0, flags=VG)
            eo, co, lo, io = oracle.minsum_decode_batch(d[f"Hdec{s}_indptr"], d[f"Hdec{s}_indices"], n, g[f"{s}_syndromes"][:2], pr, max_iter=20)
            assert np.array_equal(e2, eo) and np.array_equal(l2, lo) and np.array_equal(i2, io)
    from qldpc_amd.data import load_code
    from qldpc_amd.codes.bb_code import BBCodeCircuit
    from qldpc_amd.noise.builder import build_decoding_matrices
    from qldpc_amd.simulation.engine import prior_llrs
    c = load_code("bb288")
    cb = BBCodeCircuit(c["Hx"], c["Hz"], num_cycles=12, ell=c["ell"], m=c["m_dim"], a_x_powers=c["a_x_powers"], a_y_powers=c["a_y_powers"],
                       b_y_powers=c["b_y_powers"], b_x_powers=c["b_x_powers"])
    M = build_decoding_matrices(cb, c["Lx"], c["Lz"], 0.003, verbose=False)
    ip, ix, shape = L.canonical_csr(M["HdecZ"])
    m, n = shape
    assert n * 8 + 24 * m > 160 * 1024                      # V does not fit: the VG kernel is the one that runs
    graph = L.Graph(ip, ix, n)
    prior = prior_llrs(np.asarray(M["channel_probsZ"], dtype=np.float64))
    rng = np.random.default_rng(8)
    E = (rng.random((6, n)) < 0.0015).astype(np.int8)
    E[0] = 0
    synd = oracle.syndrome_check_batch(ip, ix, E) if hasattr(oracle, "syndrome_check_batch") else np.stack([oracle.syndrome_check(ip, ix, e) for e in E])
    err, conv, llr, it = L.minsum_decode_batch(graph, synd, prior, 30, "dynamical", 1.0)
    eo, co, lo, io = oracle.minsum_decode_batch(ip, ix, n, synd, prior, max_iter=30)
    assert np.array_equal(err, eo) and np.array_equal(llr, lo) and np.array_equal(it, io) and np.array_equal(conv.astype(bool), co.astype(bool))
    assert conv[0] and conv.sum() >= 2


def test_non_finite_priors_follow_the_reference(L, oracle, golden, monkeypatch):
    """The reference evaluates damping * q + (1 - damping) * Q_old even for damping == 1 (kernels.py:336); Q_old of iteration 0 is the
    unclipped prior, so a +-inf / NaN prior turns every later message of that column into NaN.  All kernels reproduce that (bit-exact
    against the oracle's literal arithmetic), with and without damping."""
    from qldpc_amd.data import load_code, load_circuit_matrices
    rng = np.random.default_rng(99)
    cases = []
    for tag in ("bb144", "steane"):
        c = load_code(tag)
        cases.append((tag, c["Hx_indptr"], c["Hx_indices"], int(c["n"]), [0, L.FLAG_KERNEL_GENERIC, L.FLAG_KERNEL_STREAM] if tag == "bb144" else [0, L.FLAG_KERNEL_STREAM]))
    d = load_circuit_matrices("circ72")
    cases.append(("circ72", d["HdecZ_indptr"], d["HdecZ_indices"], int(d["HdecZ_shape"][1]), [0, L.FLAG_KERNEL_STREAM, "vg"]))
    for tag, ip, ix, n, kernels in cases:
        graph = L.Graph(ip, ix, n)
        m = len(ip) - 1
        E = (rng.random((5, n)) < 0.03).astype(np.int8)
        synd = np.stack([oracle.syndrome_check(ip, ix, e) for e in E])
        base = np.log((1 - 0.03) / 0.03) + rng.normal(0, 0.3, n)
        for variant in ("inf", "mixed"):
            pr = base.copy()
            pr[rng.integers(0, n)] = np.inf
            if variant == "mixed":
                pr[rng.integers(0, n)] = -np.inf; pr[rng.integers(0, n)] = np.nan; pr[rng.integers(0, n)] = -0.0
            for damping, clip, iters in ((1.0, 20.0, 12), (0.75, 9.0, 8)):
                eo, co, lo, io = oracle.minsum_decode_batch(ip, ix, n, synd, pr, max_iter=iters, damping=damping, clip_llr=clip)
                for kern in kernels:
                    fl = L.FLAG_WG_VGLOBAL if kern == "vg" else kern
                    e2, c2, l2, i2 = L.minsum_decode_batch(graph, synd, pr, iters, "dynamical", 1.0, damping=damping, clip_llr=clip, flags=fl)
                    assert np.array_equal(i2, io) and np.array_equal(e2, eo), (tag, variant, damping, kern)
                    assert np.array_equal(np.isnan(l2), np.isnan(lo)) and np.array_equal(l2[~np.isnan(lo)], lo[~np.isnan(lo)]), (tag, variant, damping, kern)


def test_osd0_with_row_transform_in_global_memory(L, oracle, golden, monkeypatch):
    """OSD-0 for 1024 < m <= 4096: the row transform (1 MB per shot at m = 2880) lives in HBM/L2 -- the free-pivot kernel (csrc/osd_gjg.hip) for every
    shot, the reference-order kernel (osd0_lds_kernel<UG = true>) for right-hand sides outside the column space or, forced, for all.  Forced on the
    golden circuit-level cases, then on matrices that need it, against the oracle (consistent and inconsistent syndromes: the latter pin the
    reference's pivot-ROW choice, not just the pivot columns)."""
    import ctypes as C
    from qldpc_amd.decoding.osd import performOSD_enhanced
    from qldpc_amd.data import load_code, load_circuit_matrices
    for tag in ("circ72", "circ144"):
        g = golden(tag + "_decode")
        d = load_circuit_matrices(tag)
        for s in "ZX":
            ip, ix, n = d[f"Hdec{s}_indptr"], d[f"Hdec{s}_indices"], int(d[f"Hdec{s}_shape"][1])
            H = L.csr_to_scipy(ip, ix, n) if hasattr(L, "csr_to_scipy") else None
            from scipy.sparse import csr_matrix
            H = csr_matrix((np.ones(len(ix), np.int8), ix, ip), shape=(len(ip) - 1, n))
            for t, case in enumerate(g[f"{s}_osd_cases"]):
                for kfl in (L.FLAG_OSD_UG, L.FLAG_OSD_UG | L.FLAG_OSD_REFORDER):
                    sol = performOSD_enhanced(H, g[f"{s}_syndromes"][case], g[f"{s}_llr"][case], g[f"{s}_err"][case], order=0, ordering=g[f"{s}_osd_ordering"][t],
                                              flags=kfl)
                    assert np.array_equal(sol, g[f"{s}_osd_solution"][t]), (tag, s, t, kfl)
    from qldpc_amd.codes.bb_code import BBCodeCircuit
    from qldpc_amd.noise.builder import build_decoding_matrices
    c = load_code("bb288")
    rng = np.random.default_rng(3)
    for cycles in (12, 18):                       # m = 2016 (32 words per row) and m = 2880 (45 words)
        cb = BBCodeCircuit(c["Hx"], c["Hz"], num_cycles=cycles, ell=c["ell"], m=c["m_dim"], a_x_powers=c["a_x_powers"], a_y_powers=c["a_y_powers"],
                           b_y_powers=c["b_y_powers"], b_x_powers=c["b_x_powers"])
        M = build_decoding_matrices(cb, c["Lx"], c["Lz"], 0.004, verbose=False)
        ip, ix, shape = L.canonical_csr(M["HdecX"])
        m, n = shape
        assert m > 1024
        graph = L.Graph(ip, ix, n)
        B = 4
        E = (rng.random((B, n)) < 0.004).astype(np.int8)
        synd = np.stack([oracle.syndrome_check(ip, ix, e) for e in E])
        synd[B - 1] = (rng.random(m) < 0.5)                          # almost surely outside the column space
        llr = rng.normal(4.0, 3.0, (B, n)); llr[1, :50] = 0.0        # ties
        hard = (rng.random((B, n)) < 0.002).astype(np.int8)
        want = [oracle.osd0(ip, ix, n, synd[b], llr[b], hard[b]) for b in range(B)]
        for fl in (0, L.FLAG_OSD_REFORDER):
            sol = L.osd0_batch(graph, synd, llr, hard, flags=fl)
            for b in range(B):
                assert np.array_equal(sol[b], want[b]), (cycles, fl, b, int((sol[b] != want[b]).sum()))
            if b < B - 1:
                assert np.array_equal(oracle.syndrome_check(ip, ix, sol[b]), synd[b])


def test_random_irregular_graphs_all_kernels(Lb, oracle, monkeypatch):
    """Differential sweep over seeded random Tanner graphs (ragged degrees, empty rows, isolated and degree-1 columns, duplicate rows) of
    sizes that select every decoder: resident (small), workgroup-per-shot generic / lean in LDS and with posteriors in HBM/L2, streaming;
    random alpha mode, iteration cap, clip, damping and priors (some zero, some negative).  Everything bit-identical to the oracle."""
    L = Lb
    rng = np.random.default_rng(2026)
    shapes = [(5, 9, 3), (12, 30, 4), (40, 90, 5), (64, 200, 7), (150, 600, 6), (300, 2100, 9), (700, 5000, 12), (1100, 9000, 20)]
    ran = {"vg": 0, "stream": 0}
    for gi, (m, n, rd) in enumerate(shapes):
        rows = []
        for i in range(m):
            d = 0 if (i % 17 == 5) else int(rng.integers(1, rd + 1))
            rows.append(np.sort(rng.choice(n, size=min(d, n), replace=False)))
        if m > 3:
            rows[2] = rows[1].copy()                                   # duplicate check
        ip = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
        ix = (np.concatenate(rows) if ip[-1] else np.zeros(0)).astype(np.int32)
        graph = L.Graph(ip, ix, n)
        B = int(rng.integers(3, 40))
        E = (rng.random((B, n)) < 0.05).astype(np.int8)
        synd = np.stack([oracle.syndrome_check(ip, ix, e) for e in E])
        synd[0] ^= (rng.random(m) < 0.3)                               # not necessarily a realisable syndrome
        for trial in range(3):
            prior = rng.normal(2.0, 2.0, n)
            prior[rng.integers(0, n, 3)] = 0.0
            if trial == 2:
                prior[rng.integers(0, n)] = -0.0                       # non-clean: the generic kernels
            mode, alpha = [("dynamical", 1.0), ("alvarado", float(rng.uniform(0.5, 1.0))), ("alvarado-autoregressive", rng.uniform(0.4, 1.0, 3))][trial]
            damping = [1.0, 1.0, 0.8][int(rng.integers(0, 3))]
            clip = float(rng.choice([20.0, 6.5, 50.0]))
            iters = int(rng.integers(1, 25))
            ref = oracle.minsum_decode_batch(ip, ix, n, synd, prior, max_iter=iters, alpha=alpha, alpha_mode=mode, damping=damping, clip_llr=clip)
            variants = [(0, None), (L.FLAG_FIXED_ITERS, None), (L.FLAG_KERNEL_STREAM, None), (L.FLAG_WG_ROWMAJOR, None), (L.FLAG_WG_EDGE_LANES, None),
                        (L.FLAG_WG_ROWMAJOR | L.FLAG_FIXED_ITERS, None), (L.FLAG_WG_IDXLOAD, None)]
            if n >= 2000:
                variants.append((L.FLAG_WG_VGLOBAL, "1"))              # posteriors in global memory
                variants.append((L.FLAG_WG_VGLOBAL | L.FLAG_WG_ROWMAJOR, "1"))
            for flags, vg in for_build(L, variants, key=lambda v: v[0]):
                if vg:
                    ran["vg"] += 1
                out = L.minsum_decode_batch(graph, synd, prior, iters, mode, alpha, damping=damping, clip_llr=clip, flags=flags)
                ran["stream"] += flags == L.FLAG_KERNEL_STREAM
                for name, a, b in zip(("err", "conv", "llr", "iter"), (out[0], out[1].astype(bool), out[2], out[3]), (ref[0], ref[1].astype(bool), ref[2], ref[3])):
                    assert np.array_equal(a, b, equal_nan=True), (gi, trial, flags, vg, name, mode, damping, clip, iters)
    assert L.BUILD == "experiments" or (ran["vg"] >= 6 and ran["stream"] >= 20)


def test_degree_one_checks_and_nan_posteriors(Lb, oracle):
    L = Lb
    """Degree-1 checks emit +-inf messages (kernels.py:301-314 with min2 = inf) and the next iteration turns inf - inf into 0 (kernels.py:328).
    (a) every column meets at most one of them: only the waves that hold such rows run the NaN test; (b) a column meets TWO with opposite
    syndromes: its posterior is NaN and every kernel must carry the NaN exactly like the reference.  Workgroup-per-shot kernels (register-
    resident indices and the index-loading form), fixed-work and early exit, against the oracle."""
    rng = np.random.default_rng(77)
    m0, n = 700, 2600
    rows = [np.sort(rng.choice(n, size=int(rng.integers(2, 20)), replace=False)) for _ in range(m0)]
    for variant in ("one_each", "two_on_a_column"):
        extra = [np.array([c]) for c in (5, 900, 1700)] if variant == "one_each" else [np.array([5]), np.array([5]), np.array([1700])]
        allrows = extra[:1] + rows[:350] + extra[1:2] + rows[350:] + extra[2:]
        ip = np.zeros(len(allrows) + 1, np.int32)
        ip[1:] = np.cumsum([len(r) for r in allrows])
        ix = np.concatenate(allrows).astype(np.int32)
        graph = L.Graph(ip, ix, n)
        m = len(allrows)
        B = 6
        E = (rng.random((B, n)) < 0.03).astype(np.int8)
        synd = np.stack([oracle.syndrome_check(ip, ix, e) for e in E])
        synd[:, 0] = 1; synd[:, 351] = 0                                # the two degree-1 checks on column 5 disagree in variant (b)
        prior = rng.normal(2.5, 1.0, n)
        ref = oracle.minsum_decode_batch(ip, ix, n, synd, prior, max_iter=9)
        if variant == "two_on_a_column":
            assert np.isnan(ref[2]).any()
        for fl in for_build(L, (0, L.FLAG_FIXED_ITERS, L.FLAG_WG_IDXLOAD, L.FLAG_WG_ROWMAJOR, L.FLAG_WG_GENERIC, L.FLAG_KERNEL_STREAM)):
            out = L.minsum_decode_batch(graph, synd, prior, 9, "dynamical", 1.0, flags=fl)
            for name, a, b in zip(("err", "conv", "llr", "iter"), (out[0], out[1].astype(bool), out[2], out[3]), (ref[0], ref[1].astype(bool), ref[2], ref[3])):
                assert np.array_equal(a, b, equal_nan=True), (variant, fl, name)


def test_random_matrices_osd0_all_kernels(Lb, oracle, monkeypatch):
    L = Lb
    """Differential sweep of OSD-0 over seeded random matrices (dependent rows, empty rows, heavy and empty columns, ties in |llr|,
    realisable and unrealisable syndromes) through its kernels: the one-wave literal elimination (small matrices), the row transform in LDS
    (round-1 and round-2 forms of its phases), in HBM/L2 (forced), the forward and pipelined variants, and the global-memory elimination (forced).
    Solutions identical to the oracle's."""
    import ctypes as C
    rng = np.random.default_rng(4242)
    for gi, (m, n, dens) in enumerate(((4, 9, 0.4), (20, 60, 0.15), (63, 200, 0.06), (64, 64, 0.1), (130, 700, 0.03), (257, 1500, 0.012), (70, 40, 0.1), (100, 600, 0.04), (128, 256, 0.05), (200, 500, 0.03), (250, 90, 0.08))):
        Hd = (rng.random((m, n)) < dens).astype(np.int8)
        if m > 6:
            Hd[5] = Hd[1] ^ Hd[2]; Hd[3] = 0
        Hd[:, n // 2] = 0
        ip, ix, shape = L.canonical_csr(Hd)
        graph = L.Graph(ip, ix, n)
        B = 9
        E = (rng.random((B, n)) < 0.05).astype(np.int8)
        synd = np.stack([oracle.syndrome_check(ip, ix, e) for e in E])
        synd[B - 1] = rng.random(m) < 0.5
        synd[B - 2] = 0
        llr = rng.normal(1.0, 3.0, (B, n)); llr[0, : n // 3] = 1.25; llr[1] = np.round(llr[1])     # ties
        llr[2, 0::2] = np.nextafter(1.5, 2.0); llr[2, 1::2] = 1.5          # keys that differ in the last mantissa bit only, against index order
        llr[3, : n // 2] = -(1.0 + np.arange(n // 2)[::-1] * 2.0 ** -45)      # a long run of keys equal in their 40 high bits, descending
        hard = (rng.random((B, n)) < 0.1).astype(np.int8)
        want = np.stack([oracle.osd0(ip, ix, n, synd[b], llr[b], hard[b]) for b in range(B)])
        # (0 = the one-wave literal elimination for m <= 128, n <= 1024, else the transform kernel; FLAG_OSD_LDS forces the latter)
        for env in for_build(L, (0, L.FLAG_OSD_LDS, L.FLAG_OSD_REFORDER, L.FLAG_OSD_LDS | L.FLAG_OSD_REFORDER, L.FLAG_OSD_UG, L.FLAG_OSD_UG | L.FLAG_OSD_REFORDER, L.FLAG_OSD_GLOBAL)):
            sol = L.osd0_batch(graph, synd, llr, hard, flags=env)
            assert np.array_equal(sol, want), (gi, env, np.flatnonzero((sol != want).any(1)))
        # the free-pivot kernels sort only a head of the |llr| order up front and the other columns when a sweep gets past it ("osd_presort"): short heads
        # (the rest path on nearly every shot, the ties above inside and across the head's key bound) and the whole order up front
        try:
            for presort in (16, 100, 333, 0):
                L.set_option("osd_presort", presort)
                for env in (L.FLAG_OSD_LDS, L.FLAG_OSD_UG):
                    sol = L.osd0_batch(graph, synd, llr, hard, flags=env)
                    assert np.array_equal(sol, want), (gi, presort, env, np.flatnonzero((sol != want).any(1)))
        finally:
            L.set_option("osd_presort", -1)


def _regular_graph(rng, m, cdeg, vdeg):
    """(cdeg, vdeg)-regular bipartite graph from cdeg / vdeg column blocks of m columns, each block a sum of vdeg random distinct cyclic
    shifts (row i meets columns blk * m + (i + s) mod m): no repeated edges by construction.  CSR with sorted rows."""
    nblk = cdeg // vdeg
    n = m * nblk
    rows = np.zeros((m, cdeg), np.int64)
    for blk in range(nblk):
        shifts = rng.choice(m, size=vdeg, replace=False)
        for t, sft in enumerate(shifts):
            rows[:, blk * vdeg + t] = blk * m + (np.arange(m) + sft) % m
    rows = np.sort(rows, axis=1)
    return np.arange(0, m * cdeg + 1, cdeg, dtype=np.int32), rows.reshape(-1).astype(np.int32), n


def test_other_regular_degrees(L, oracle):
    """The branch-free kernel is instantiated for (6,3), (4,2) and (8,4)-regular graphs; the bivariate-bicycle codes only exercise (6,3).
    Random (4,2) and (8,4) graphs: decode (all kernels) and the fused Monte-Carlo tally against the oracle."""
    rng = np.random.default_rng(31)
    for cdeg, vdeg, m in ((4, 2, 30), (4, 2, 64), (8, 4, 24), (8, 4, 60)):
        ip, ix, n = _regular_graph(rng, m, cdeg, vdeg)
        graph = L.Graph(ip, ix, n)
        B = 257
        E = (rng.random((B, n)) < 0.04).astype(np.int8)
        synd = np.stack([oracle.syndrome_check(ip, ix, e) for e in E])
        prior = np.full(n, np.log(0.96 / 0.04))
        for damping in (1.0, 0.85):
            ref = oracle.minsum_decode_batch(ip, ix, n, synd, prior, max_iter=30, damping=damping)
            for flags in (0, L.FLAG_FIXED_ITERS, L.FLAG_KERNEL_GENERIC, L.FLAG_KERNEL_STREAM):
                out = L.minsum_decode_batch(graph, synd, prior, 30, "dynamical", 1.0, damping=damping, flags=flags)
                for a, b in zip((out[0], out[1].astype(bool), out[2], out[3]), (ref[0], ref[1].astype(bool), ref[2], ref[3])):
                    assert np.array_equal(a, b, equal_nan=True), (cdeg, vdeg, m, damping, flags)
        Lmat = (rng.random((3, n)) < 0.3).astype(np.uint8)
        want = oracle.cc_sample_decode_tally(ip, ix, n, Lmat, 0.03, 77, 5, 6000, max_iter=25, threads=0)
        for flags in (0, L.FLAG_FIXED_ITERS, L.FLAG_MC_UNFUSED):
            got = L.cc_sample_decode_tally(graph, Lmat, 0.03, 77, 5, 6000, max_iter=25, flags=flags)
            assert np.array_equal(got, want), (cdeg, vdeg, m, flags, got.tolist(), want.tolist())
        assert want[L.TALLY["osd_z"]] > 0


@pytest.mark.parametrize("code,cycles,p", [("bb90", 4, 0.004), ("bb108", 3, 0.006)])
def test_other_codes_circuit_level_pipeline(L, oracle, code, cycles, p):
    """The whole circuit-level path on the reference's other codes (k = 8 logicals, other lattice sizes): circuit generator -> builder
    (GPU single-fault signatures) -> run_simulation; the tally equals the oracle's literal simulate-and-decode of the same Philox trials.
    (The circuit arrays for these codes are not pinned by a reference fixture -- only [[72]] and [[144]] are -- this pins sampler + decoders.)"""
    from qldpc_amd.data import load_code
    from qldpc_amd.codes.bb_code import BBCodeCircuit
    from qldpc_amd.noise.compiled import CompiledCircuit
    from qldpc_amd.noise.builder import build_decoding_matrices
    from qldpc_amd.simulation.engine import run_simulation, prior_llrs
    c = load_code(code)
    bb = dict(ell=c["ell"], m=c["m_dim"], a_x_powers=c["a_x_powers"], a_y_powers=c["a_y_powers"], b_y_powers=c["b_y_powers"], b_x_powers=c["b_x_powers"])
    cb = BBCodeCircuit(c["Hx"], c["Hz"], num_cycles=cycles, **bb)
    M = build_decoding_matrices(cb, c["Lx"], c["Lz"], p, verbose=False)
    comp = CompiledCircuit(cb.get_full_circuit(), cb.cycle * 2, cb.lin_order, cb.data_qubits, cb.Xchecks, cb.Zchecks)
    circ = oracle.make_circuit(comp, c["Lx"], c["Lz"])
    k = c["Lx"].shape[0]
    secs = []
    for s in "ZX":
        ip, ix, shape = L.canonical_csr(M[f"Hdec{s}"])
        flr = int(M[f"first_logical_row{s}"])
        lip, lix, _ = L.canonical_csr(np.asarray(M[f"H{s}_full"])[flr:flr + k])
        secs.append(oracle.make_sector(ip, ix, shape[1], prior_llrs(np.asarray(M[f"channel_probs{s}"], dtype=np.float64)), lip, lix))
    N = 300
    res = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], p, num_trials=N, num_cycles=cycles, maxIter=40, precomputed_matrices=M, base_seed=123, batch=128, **bb)
    ref = oracle.circuit_sample_decode_tally(circ, secs[0], secs[1], p, 123, 0, N, max_iter=40, threads=0)
    assert np.array_equal(res["tally"], ref), (res["tally"].tolist(), ref.tolist())
    assert ref[0] == N and ref[L.TALLY["unsat_z"]] == 0 and ref[L.TALLY["unsat_x"]] == 0 and ref[L.TALLY["osd_z"]] + ref[L.TALLY["osd_x"]] > 0


def _rank_run_simulation(rank, world, port, out_dir):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import json
    import numpy as np
    import torch  # noqa: F401  (before the HIP library)
    import torch.distributed as dist
    import qldpc_amd  # noqa: F401
    from qldpc_amd.data import load_code, load_precomputed_matrices
    from qldpc_amd.simulation.engine import run_simulation
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = load_code("bb72")
    bb = dict(ell=c["ell"], m=c["m_dim"], a_x_powers=c["a_x_powers"], a_y_powers=c["a_y_powers"], b_y_powers=c["b_y_powers"], b_x_powers=c["b_x_powers"])
    M = load_precomputed_matrices("circ72")
    out = {}
    r = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=501, num_cycles=6, maxIter=30, precomputed_matrices=M, base_seed=77, batch=64, **bb)
    out["plain"] = [int(x) for x in r["tally"]]
    r = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=501, num_cycles=6, maxIter=30, precomputed_matrices=M, base_seed=77, batch=64,
                       target_logical_errors=37, alpha_mode="alvarado", alpha_estimation_trials=200, **bb)
    out["target"] = [r["num_trials"], r["logical_errors"], r["logical_error_rate"], r["alpha_r2_z"]]
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as fh:
        json.dump(out, fh)
    dist.destroy_process_group()


def test_run_simulation_sharded_over_two_ranks(L, tmp_path):
    """N > 1 on the real device path: two processes (gloo; they share this box's one GPU) each run their contiguous trial shard; the
    all-reduced tally, the exact early stop and the estimated alpha are identical on both ranks and equal to the single-process run."""
    import json
    import socket
    import torch.multiprocessing as mp
    from qldpc_amd.data import load_code, load_precomputed_matrices
    from qldpc_amd.simulation.engine import run_simulation
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.get_context("spawn")
    mp.spawn(_rank_run_simulation, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert got[0] == got[1]
    c = load_code("bb72")
    bb = dict(ell=c["ell"], m=c["m_dim"], a_x_powers=c["a_x_powers"], a_y_powers=c["a_y_powers"], b_y_powers=c["b_y_powers"], b_x_powers=c["b_x_powers"])
    M = load_precomputed_matrices("circ72")
    r = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=501, num_cycles=6, maxIter=30, precomputed_matrices=M, base_seed=77, batch=64, **bb)
    assert [int(x) for x in r["tally"]] == got[0]["plain"]
    r = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=501, num_cycles=6, maxIter=30, precomputed_matrices=M, base_seed=77, batch=64,
                       target_logical_errors=37, alpha_mode="alvarado", alpha_estimation_trials=200, **bb)
    assert [r["num_trials"], r["logical_errors"], r["logical_error_rate"], r["alpha_r2_z"]] == got[0]["target"]
    assert r["logical_errors"] == 37 and r["num_trials"] < 501


def _num_workers_case(devices, oracle, golden, L):
    """run_simulation over the worker list `devices` == the one-plan run == the oracle: tallies, the exact stop trial, estimator factors."""
    from qldpc_amd.data import load_code, load_precomputed_matrices
    from qldpc_amd.simulation.engine import run_simulation
    c = load_code("bb72")
    g, circ, secs, graphs, priors, masks = _circuit_setup(L, oracle, "circ72", golden)
    bb = dict(ell=c["ell"], m=c["m_dim"], a_x_powers=c["a_x_powers"], a_y_powers=c["a_y_powers"], b_y_powers=c["b_y_powers"], b_x_powers=c["b_x_powers"])
    M = load_precomputed_matrices("circ72")
    kw = dict(num_cycles=6, maxIter=30, precomputed_matrices=M, base_seed=77, batch=64, **bb)
    ref = oracle.circuit_sample_decode_tally(circ, secs[0], secs[1], 0.005, 77, 0, 501, max_iter=30, threads=0)
    one = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=501, devices=[0], **kw)
    many = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=501, devices=devices, **kw)
    assert many["num_workers"] == len(devices) and many["devices"] == list(devices) and one["num_workers"] == 1
    assert np.array_equal(one["tally"], ref) and np.array_equal(many["tally"], ref)
    for k in ("logical_error_rate", "z_logical_error_rate", "x_logical_error_rate", "num_trials", "logical_errors"):
        assert one[k] == many[k]
    # the reference's own keyword (engine.py:204): a request beyond the visible GPUs is capped, never an error; osd_order = 2 as main.py:44 runs it
    capped = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=501, num_workers=8, osd_order=2, **kw)
    assert capped["num_workers"] == min(8, L.device_count()) and np.array_equal(capped["tally"], ref)
    # in-order early stop + estimated factors: the run ends AT the trial the one-plan run ends at, with identical factors
    kw2 = dict(kw, target_logical_errors=37, alpha_mode="alvarado", alpha_estimation_trials=200, scopt=True)
    a = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=501, devices=[0], **kw2)
    b = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=501, devices=devices, **kw2)
    for k in ("num_trials", "logical_errors", "logical_error_rate", "z_logical_error_rate", "x_logical_error_rate", "alpha_r2_z", "alpha_r2_x", "beta_z", "beta_x"):
        assert a[k] == b[k], k
    assert a["logical_errors"] == 37 and a["num_trials"] < 501
    # a ragged split: more workers than trials in a round, and a worker with an empty range
    tiny = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=2, devices=list(devices) + [devices[0]], **kw)
    assert np.array_equal(tiny["tally"], oracle.circuit_sample_decode_tally(circ, secs[0], secs[1], 0.005, 77, 0, 2, max_iter=30, threads=0))


def test_run_simulation_num_workers_two_plans_one_card(L, oracle, golden):
    """run_simulation(num_workers / devices): ONE call drives several plans from one host thread each (the reference's pool of engine.py:433-435 with a
    worker = a GPU plan).  On this pool: two plans sharing GPU 0."""
    _num_workers_case([0, 0], oracle, golden, L)


def test_two_gpus_run_simulation_num_workers(L, oracle, golden):
    """The same on two cards: num_workers=None takes every visible GPU."""
    if L.device_count() < 2:
        pytest.skip("needs two GPUs")
    from qldpc_amd.data import load_code, load_precomputed_matrices
    from qldpc_amd.simulation.engine import run_simulation
    _num_workers_case([0, 1], oracle, golden, L)
    c = load_code("bb72")
    bb = dict(ell=c["ell"], m=c["m_dim"], a_x_powers=c["a_x_powers"], a_y_powers=c["a_y_powers"], b_y_powers=c["b_y_powers"], b_x_powers=c["b_x_powers"])
    r = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], 0.005, num_trials=300, num_cycles=6, maxIter=30, precomputed_matrices=load_precomputed_matrices("circ72"),
                       base_seed=77, batch=64, **bb)
    assert r["num_workers"] == L.device_count() and r["devices"] == list(range(L.device_count()))


def test_device_resident_decode_osd_check_chain(L, oracle):
    """decode -> OSD-0 on the unconverged shots -> syndrome check entirely on torch CUDA tensors and one non-default stream
    (qldpc_minsum_decode_batch_dev, qldpc_osd0_batch_dev with a device-resident selection, qldpc_gf2_spmv_batch_dev): the device-side
    analogue of _run_single_trial_fast (engine.py:86-100).  Results equal the oracle's per-shot pipeline."""
    import ctypes as C
    import torch
    from qldpc_amd.data import load_code
    c = load_code("bb144")
    ip, ix, n, m = c["Hx_indptr"], c["Hx_indices"], c["n"], c["m"]
    g = L.Graph(ip, ix, n)
    rng = np.random.default_rng(19)
    B = 600
    E = (rng.random((B, n)) < 0.05).astype(np.int8)
    synd = np.stack([oracle.syndrome_check(ip, ix, e) for e in E])
    prior = np.full(n, np.log(0.95 / 0.05))
    err_o, conv_o, llr_o, it_o = oracle.minsum_decode_batch(ip, ix, n, synd, prior, max_iter=15, threads=0)
    want = err_o.copy()
    for b in np.flatnonzero(conv_o == 0):
        want[b] = oracle.osd0(ip, ix, n, synd[b], llr_o[b], err_o[b])
    assert 20 < (conv_o == 0).sum() < B
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ds = torch.from_numpy(synd).cuda(); dp = torch.from_numpy(prior).cuda()
        de = torch.empty((B, n), dtype=torch.int8, device="cuda"); dl = torch.empty((B, n), dtype=torch.float64, device="cuda")
        dc = torch.empty(B, dtype=torch.uint8, device="cuda"); di = torch.empty(B, dtype=torch.int32, device="cuda")
        dchk = torch.empty((B, m), dtype=torch.int8, device="cuda")
        st.synchronize()
        p = lambda t: C.c_void_p(t.data_ptr())       # noqa: E731
        s = C.c_void_p(st.cuda_stream)
        L.check(L.lib().qldpc_minsum_decode_batch_dev(g.handle, C.c_int64(B), p(ds), p(dp), C.c_int(15), C.c_int(L.ALPHA_DYNAMIC), C.c_double(1.0), None,
                                                      C.c_int(0), C.c_double(1.0), C.c_double(20.0), C.c_int(0), p(de), p(dl), p(dc), p(di), s))
        sel = torch.nonzero(dc == 0).flatten().to(torch.int32)          # stays on the device
        cnt = torch.tensor([sel.numel()], dtype=torch.int32, device="cuda")
        dsol = de.clone()                                               # converged shots keep the BP decision
        L.check(L.lib().qldpc_osd0_batch_dev(g.handle, C.c_int64(B), p(ds), p(dl), p(de), None, p(sel), p(cnt), 0, p(dsol), s))
        L.check(L.lib().qldpc_gf2_spmv_batch_dev(g.handle, C.c_int64(B), p(dsol), p(dchk), s))
        st.synchronize()
    assert np.array_equal(dsol.cpu().numpy(), want)
    assert np.array_equal(dchk.cpu().numpy(), synd)                     # every final answer reproduces its syndrome
    # without a selection every shot is solved
    with torch.cuda.stream(st):
        dall = torch.empty_like(de)
        L.check(L.lib().qldpc_osd0_batch_dev(g.handle, C.c_int64(B), p(ds), p(dl), p(de), None, None, None, 0, p(dall), s))
        st.synchronize()
    b = int(np.flatnonzero(conv_o == 1)[0])
    assert np.array_equal(dall.cpu().numpy()[b], oracle.osd0(ip, ix, n, synd[b], llr_o[b], err_o[b]))


def test_bench_starts_its_own_ranks(L, tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts two ranks by itself (gloo here: both share this box's one card; on an
    8-GPU node the same command line runs over RCCL).  The line it prints must account for both ranks' shots and trials."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["QLDPC_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "65536", "--p-sweep", "0.004,0.006",
                        "--circuit", "circ72", "--circuit-batch", "1024", "--circuit-steps", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout                                   # exactly one JSON line: rank 0's
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 2
    assert out["tally"]["trials"] == 2 * 2 * 65536                    # world x steps x batch shots went through the all-reduce
    assert out["circuit_level"]["tally"]["trials"] == 2 * 2 * 1024
    assert out["value"] > 0 and out["circuit_level"]["value"] > 0
    assert [pt["p"] for pt in out["p_sweep"]["points"]] == [0.004, 0.006] and all(pt["tally"]["trials"] == 2 * 2 * 65536 for pt in out["p_sweep"]["points"])
    assert len(out["per_rank"]["per_rank_ms_per_step"]) == 2 and len(out["per_rank"]["tally_allreduce_ms"]) == 2
    # the same shots on one rank give the same tally (the Philox streams are keyed by the global shot index)
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--batch", "65536",
                         "--circuit", "circ72", "--circuit-batch", "1024", "--circuit-steps", "4", "--no-cpu-baseline"],
                        env=env, capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][0])
    assert one["tally"] == out["tally"] and one["circuit_level"]["tally"] == out["circuit_level"]["tally"]
    assert set(one["circuit_level"]["phases_ms_per_step"]) == set(L.CIRCUIT_PHASES)
    assert one["roofline"]["clock_mhz"] is None or 500 < one["roofline"]["clock_mhz"] < 3000


def test_native_rccl_tally_allreduce(L):
    """(e) natively on RCCL through the C ABI (no torch): a world of one is the identity, through both ways of forming the communicator.
    N > 1 needs an N-GPU node and stays unmeasured here."""
    t = np.arange(16, dtype=np.int64) * 1000003 + 7
    comm = L.Comm.init_all(1)
    assert (comm.nranks, comm.nlocal) == (1, 1)
    assert np.array_equal(comm.allreduce(t), t)
    assert np.array_equal(comm.allreduce(t.reshape(1, 16)), t.reshape(1, 16))
    comm.close()
    uid = L.Comm.unique_id()
    assert len(uid) == 128
    c2 = L.Comm.init_rank(1, 0, uid, 0)
    assert np.array_equal(c2.allreduce(t), t)
    from qldpc_amd.parallel import allreduce_tally
    assert np.array_equal(allreduce_tally(t, comm=c2), t)              # the host mirror's collective on the native communicator
    c2.close()
    with pytest.raises(L.QldpcError):
        L.Comm.init_all(L.device_count() + 1)


def _two_gpus(L):
    return L.device_count() >= 2


def _nccl_rank(rank, world, port, out_dir):
    """one rank of the real N > 1 path: its own GPU, torch.distributed 'nccl' (= RCCL), shot range sharded, one all-reduce of the tally"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    import qldpc_amd  # noqa: F401
    from qldpc_amd import _lib, parallel
    from qldpc_amd.data import load_code
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    c = load_code("bb144")
    graph = _lib.Graph(c["Hx_indptr"], c["Hx_indices"], c["n"], device=rank)
    t = parallel.run_code_capacity(graph, c["Lx"], 0.03, 4242, 30001, max_iter=50, device=rank)
    np.save(os.path.join(out_dir, f"nccl_tally_{rank}.npy"), t)
    assert torch.cuda.current_device() == rank                          # the entry points put the caller's device back
    dist.destroy_process_group()


def test_two_gpus_torch_nccl_sharded_tally(L, oracle, tmp_path):
    """(e) on hardware: two ranks, two cards, RCCL through torch.distributed.  Skips on a one-GPU box (this pool's); runs the day two are visible."""
    if not _two_gpus(L):
        pytest.skip("needs two GPUs")
    import socket
    import torch.multiprocessing as mp
    from qldpc_amd.data import load_code
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_nccl_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    c = load_code("bb144")
    ref = oracle.cc_sample_decode_tally(c["Hx_indptr"], c["Hx_indices"], c["n"], c["Lx"], 0.03, 4242, 0, 30001, max_iter=50, threads=0)
    for r in range(2):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), f"nccl_tally_{r}.npy")), ref)


def _native_rank(rank, world, uid_path, out_dir):
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import qldpc_amd  # noqa: F401
    from qldpc_amd import _lib
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if rank == 0:
        uid = _lib.Comm.unique_id()
        with open(uid_path + ".tmp", "wb") as fh:
            fh.write(uid)
        os.replace(uid_path + ".tmp", uid_path)
    else:
        for _ in range(600):
            if os.path.exists(uid_path):
                break
            time.sleep(0.1)
        with open(uid_path, "rb") as fh:
            uid = fh.read()
    comm = _lib.Comm.init_rank(world, rank, uid, rank)
    t = (np.arange(16, dtype=np.int64) + 1) * (rank + 1)
    np.save(os.path.join(out_dir, f"native_{rank}.npy"), comm.allreduce(t))
    comm.close()


def test_two_gpus_native_rccl_communicators(L, tmp_path):
    """qldpc_comm_init_all(2) in one process and qldpc_comm_init_rank x 2 in two (the launcher-without-PyTorch path), on two cards."""
    if not _two_gpus(L):
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    comm = L.Comm.init_all(2)
    assert (comm.nranks, comm.nlocal) == (2, 2)
    t = np.stack([np.arange(16, dtype=np.int64) + 1, (np.arange(16, dtype=np.int64) + 1) * 10])
    got = comm.allreduce(t)                                             # one tally per local rank in, the sum on every local rank out
    assert np.array_equal(got, np.stack([t.sum(0), t.sum(0)]))
    comm.close()
    mp.spawn(_native_rank, args=(2, os.path.join(str(tmp_path), "uid.bin"), str(tmp_path)), nprocs=2, join=True)
    want = (np.arange(16, dtype=np.int64) + 1) * 3
    for r in range(2):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), f"native_{r}.npy")), want)


def test_two_gpus_bench_over_rccl(L):
    """`python bench.py --gpus 2` on two cards with the default backend (nccl = RCCL): the driver's scaling run in miniature, incl. the p-sweep of
    BASELINE config 4; tallies equal the one-rank run's."""
    if not _two_gpus(L):
        pytest.skip("needs two GPUs")
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "QLDPC_BENCH_BACKEND")}
    outs = []
    for gpus, steps in ((2, 2), (1, 4)):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--steps", str(steps), "--warmup", "1", "--batch", "65536", "--code", "bb288",
                            "--p-sweep", "0.004,0.006", "--circuit", "circ72", "--circuit-batch", "1024", "--circuit-steps", str(steps), "--no-cpu-baseline"],
                           env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0]))
    two, one = outs
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and two["collective_backend"] == "nccl"
    assert two["tally"] == one["tally"] and two["circuit_level"]["tally"] == one["circuit_level"]["tally"]
    assert [pt["tally"] for pt in two["p_sweep"]["points"]] == [pt["tally"] for pt in one["p_sweep"]["points"]]
    assert len(two["per_rank"]["per_rank_ms_per_step"]) == 2 and len(two["per_rank"]["tally_allreduce_ms"]) == 2


def test_run_simulation_osd_order_on_unsatisfiable_trials(L, oracle):
    """run_simulation(osd_order = 2) with FOREIGN decoding matrices (most columns removed, so the circuit's syndromes fall outside the
    column space): OSD-0 leaves those trials unsatisfied and the reference enters the combination sweep (osd.py:31-75).  The engine
    routes such batches through qldpc_osdw_batch; verdicts equal the per-trial pipeline assembled from the oracle's pieces."""
    import scipy.sparse as sp
    from qldpc_amd.data import load_code, load_precomputed_matrices
    from qldpc_amd.simulation.engine import run_simulation, prior_llrs
    from qldpc_amd.codes.bb_code import BBCodeCircuit
    from qldpc_amd.noise.compiled import CompiledCircuit
    c = load_code("bb72")
    bb = dict(ell=c["ell"], m=c["m_dim"], a_x_powers=c["a_x_powers"], a_y_powers=c["a_y_powers"], b_y_powers=c["b_y_powers"], b_x_powers=c["b_x_powers"])
    M = load_precomputed_matrices("circ72")
    rng = np.random.default_rng(17)
    F, secs = {"num_cycles": 6, "k": M["k"]}, []
    for s in "ZX":
        H = M[f"Hdec{s}"].tocsc()
        n = H.shape[1]
        keep = np.sort(rng.choice(n, 160, replace=False))
        lip, lix = M[f"H{s}_logical"]
        Ld = sp.csr_matrix((np.ones(len(lix), np.int8), lix, lip), shape=(len(lip) - 1, n)).toarray()[:, keep]
        F[f"Hdec{s}"] = sp.csr_matrix(H[:, keep])
        F[f"H{s}_logical"] = Ld
        F[f"channel_probs{s}"] = M[f"channel_probs{s}"][keep]
        ip, ix, _ = L.canonical_csr(F[f"Hdec{s}"])
        secs.append((ip, ix, len(keep), prior_llrs(F[f"channel_probs{s}"]), Ld))
    N, seed, p = 96, 4321, 0.005
    res = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], p, num_trials=N, num_cycles=6, maxIter=20, osd_order=2, precomputed_matrices=F,
                         base_seed=seed, batch=32, **bb)
    cb = BBCodeCircuit(c["Hx"], c["Hz"], num_cycles=6, **bb)
    comp = CompiledCircuit(cb.get_full_circuit(), cb.cycle * 2, cb.lin_order, cb.data_qubits, cb.Xchecks, cb.Zchecks)
    circ = oracle.make_circuit(comp, c["Lx"], c["Lz"])
    z_err = x_err = tot = unsat = 0
    for t in range(N):
        spz, tz, spx, tx = oracle.circuit_sample(circ, p, seed, t)
        bad = []
        for (ip, ix, n, prior, Ld), synd, true in zip(secs, (spz, spx), (tz, tx)):
            err, conv, llr, it = oracle.minsum_decode_batch(ip, ix, n, synd[None, :], prior, max_iter=20)
            det = err[0]
            if not conv[0]:
                det = oracle.osdw(ip, ix, n, synd, llr[0], err[0], 2)
                unsat += int(not np.array_equal(oracle.syndrome_check(ip, ix, det.astype(np.int8)), synd))
            bad.append(bool(np.any((Ld.astype(np.int64) @ det.astype(np.int64)) % 2 != true)))
        z_err += bad[0]; x_err += bad[1]; tot += bad[0] or bad[1]
    assert unsat > 10                                                  # the sweep really ran on unsatisfiable syndromes
    T = L.TALLY
    assert (res["num_trials"], res["logical_errors"]) == (N, tot)
    assert (int(res["tally"][T["z_err"]]), int(res["tally"][T["x_err"]])) == (z_err, x_err)
    # target_logical_errors goes through the same routing: the exact stop trial of the in-order loop
    stop = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], p, num_trials=N, num_cycles=6, maxIter=20, osd_order=2, precomputed_matrices=F,
                          base_seed=seed, batch=32, target_logical_errors=5, **bb)
    assert stop["logical_errors"] == 5 and stop["num_trials"] <= N
