"""Code and decoding-matrix DATA shipped with the package (re-packed from the reference's codes/*.npz and
matrix_cache/*.npz by tests/golden/make_golden.py; data files, no reference source)."""
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CODE_TAGS = ("steane", "bb72", "bb90", "bb108", "bb144", "bb288")


def _csr(H):
    H = np.asarray(H)
    rows, cols = np.nonzero(H)
    indptr = np.zeros(H.shape[0] + 1, np.int64)
    np.add.at(indptr, rows + 1, 1)
    return np.cumsum(indptr).astype(np.int32), cols.astype(np.int32)


def load_code(tag):
    """dict with Hx, Hz (uint8 dense), their CSR (Hx_indptr, ...), Lx, Lz (uint8 [k, n]), m, n, k and BB parameters."""
    with np.load(os.path.join(_HERE, "codes.npz")) as d:
        out = {"Hx": d[f"{tag}_Hx"], "Hz": d[f"{tag}_Hz"]}
        if tag == "steane":
            # self-dual CSS code: the all-ones vector is the logical operator of both types
            out["Lx"] = np.ones((1, 7), np.uint8)
            out["Lz"] = np.ones((1, 7), np.uint8)
        else:
            out["Lx"], out["Lz"] = d[f"{tag}_Lx"], d[f"{tag}_Lz"]
            ell, mm, dist = (int(x) for x in d[f"{tag}_params"])
            out.update(ell=ell, m_dim=mm, distance=dist)
            for k in ("a_x_powers", "a_y_powers", "b_y_powers", "b_x_powers"):
                out[k] = d[f"{tag}_{k}"]
    for h in ("Hx", "Hz"):
        out[h + "_indptr"], out[h + "_indices"] = _csr(out[h])
    out["m"], out["n"] = out["Hx"].shape
    out["k"] = out["Lx"].shape[0]
    return out


def load_circuit_matrices(tag):
    """Circuit-level decoding matrices the reference cached at p = 0.005, in CSR form: tags circ72 (6 cycles), circ90 / circ108 (10),
    circ144 (12), circ288 (18)."""
    with np.load(os.path.join(_HERE, f"{tag}_p005.npz")) as d:
        return {k: d[k] for k in d.files}


def load_precomputed_matrices(tag):
    """The same data as a dict run_simulation accepts as ``precomputed_matrices`` (sparse Hdec, logical rows as CSR tuples)."""
    import scipy.sparse as sp
    d = load_circuit_matrices(tag)
    out = {"channel_probsZ": d["channel_probsZ"], "channel_probsX": d["channel_probsX"], "num_cycles": int(d["num_cycles"]), "k": int(d["k"])}
    for s in ("Z", "X"):
        m, n = (int(x) for x in d[f"Hdec{s}_shape"])
        ix, ip = d[f"Hdec{s}_indices"], d[f"Hdec{s}_indptr"]
        out[f"Hdec{s}"] = sp.csr_matrix((np.ones(ix.size, np.int8), ix, ip), shape=(m, n))
        out[f"H{s}_logical"] = (d[f"H{s}_logical_indptr"], d[f"H{s}_logical_indices"])
    return out
