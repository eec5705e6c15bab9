import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as d:
        return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (checker only)."""
    from oracle import oracle as orc
    orc.build()
    return orc


def experiments_lib():
    """Bindings of libqldpc_hip_experiments.so (same ABI + the measured-and-rejected kernels): a second copy of the package, imported under the
    name qldpc_amd_x so that its _lib module can point at the other build.  Graph handles and plans belong to the library that made them."""
    import importlib.util
    if "qldpc_amd_x" not in sys.modules:
        root = os.path.join(ROOT, "qldpc-branched-off_amd")
        spec = importlib.util.spec_from_file_location("qldpc_amd_x", os.path.join(root, "__init__.py"), submodule_search_locations=[root])
        mod = importlib.util.module_from_spec(spec)
        sys.modules["qldpc_amd_x"] = mod
        spec.loader.exec_module(mod)
        mod._lib.select_build("experiments")
    return sys.modules["qldpc_amd_x"]._lib


def package_of(L):
    """the package copy a bindings module belongs to (qldpc_amd or qldpc_amd_x)"""
    return sys.modules[L.__name__.rsplit(".", 1)[0]]


def assert_llr_close(a, b, tol=1e-5):
    """LLR comparison of the north-star contract: |a-b| <= tol, equal infinities equal, NaN==NaN."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape
    fin = np.isfinite(a) & np.isfinite(b)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    inf = np.isinf(a) | np.isinf(b)
    assert np.array_equal(a[inf], b[inf])
    if fin.any():
        assert np.max(np.abs(a[fin] - b[fin])) <= tol


# ---- f4 estimator fixtures (tests/golden/estimators.npz, produced by the reference's alpha.py / scopt.py) ----------------
class ReplayRng:
    """Stands in for the numpy Generator of the golden run: `random(shape)` returns 0.0 where the recorded draw was below the
    error rate and 1.0 elsewhere, in the recorded order (the fixture stores the error bits, not the uniform doubles)."""

    def __init__(self, error_bits):
        self.bits = np.asarray(error_bits, dtype=bool)
        self.row = 0

    def random(self, shape):
        rows, n = (1, int(shape)) if np.isscalar(shape) else (int(shape[0]), int(shape[1]))
        out = np.where(self.bits[self.row:self.row + rows, :n], 0.0, 1.0)
        assert out.shape[0] == rows, "estimator drew more patterns than the golden run"
        self.row += rows
        return out[0] if np.isscalar(shape) else out


def estimator_cases(G):
    """-> list of dicts describing every case of estimators.npz with its graph in CSR form (from the package data files)."""
    import qldpc_amd  # noqa: F401
    from qldpc_amd.data import load_code, load_circuit_matrices
    c = load_code("bb72")
    d = load_circuit_matrices("circ72")
    graphs = {"bb72": (c["Hx_indptr"], c["Hx_indices"], int(c["n"])),
              "circ72": (d["HdecZ_indptr"], d["HdecZ_indices"], int(d["HdecZ_shape"][1]))}
    cases = []
    for name in [str(x) for x in G["cases"]]:
        g = lambda k: G[f"{name}__{k}"]          # noqa: E731
        indptr, indices, n = graphs[str(g("graph"))]
        case = dict(name=name, kind=str(g("kind")), indptr=indptr, indices=indices, n=n, p=float(g("p")), prior=G[f"prior__{str(g('prior'))}"],
                    errors=np.unpackbits(g("errors"), axis=1)[:, :n].astype(np.int8), trials=int(g("trials")), bins=int(g("bins")),
                    hist=g("hist"), edges=g("edges"))
        for k in ("iters", "damping", "clip", "alpha_mode", "alpha"):
            if f"{name}__{k}" in G:
                v = g(k)
                case[k] = str(v) if k == "alpha_mode" else (v if v.ndim else v.item())
        case["out"] = {k.split("__out_")[1]: G[k] for k in G if k.startswith(f"{name}__out_")}
        cases.append(case)
    return cases


def reference_fit(samples, bits, bins, flip=False):
    """The reference's post-processing (alpha.py:23-66 / scopt.py:139-166) on explicit samples -> (slope, r2, hist0, hist1, edges)."""
    from scipy.optimize import curve_fit
    s0, s1 = samples[bits == 0], samples[bits == 1]
    s0, s1 = s0[np.isfinite(s0)], s1[np.isfinite(s1)]
    rng = (min(s0.min(), s1.min()), max(s0.max(), s1.max()))
    h0, edges = np.histogram(s0, bins=bins, range=rng, density=True)
    h1, _ = np.histogram(s1, bins=bins, range=rng, density=True)
    centres = (edges[:-1] + edges[1:]) / 2.0
    ok = (h0 > 0) & (h1 > 0)
    y = np.log(h1[ok] / h0[ok]) if flip else np.log(h0[ok] / h1[ok])
    x = centres[ok]
    popt, _ = curve_fit(lambda t, a: a * t, x, y)
    fit = popt[0] * x
    tot = np.sum((y - np.mean(y)) ** 2)
    return popt[0], 1.0 - (np.sum((y - fit) ** 2) / tot if tot > 0 else np.nan), h0, h1, edges


def osdw_cases(G):
    """Cases of tests/golden/osdw.npz (reference performOSD_enhanced with order > 0) with their matrices in CSR form."""
    from scipy.sparse import csr_matrix
    out = []
    for name in [str(x) for x in G["cases"]]:
        H = csr_matrix(G[f"graph__{str(G[f'{name}__graph'])}"].astype(np.int8))
        H.sort_indices()
        g = lambda k: G[f"{name}__{k}"]          # noqa: E731
        out.append(dict(name=name, H=H, indptr=H.indptr.astype(np.int32), indices=H.indices.astype(np.int32), n=H.shape[1], order=int(g("order")),
                        maxc=int(g("maxc")) or None, syndrome=g("syndrome"), llr=g("llr"), hard=g("hard"), ordering=g("ordering"),
                        solution=g("solution"), swept=bool(g("swept")), differs=bool(g("differs_from_osd0"))))
    return out
