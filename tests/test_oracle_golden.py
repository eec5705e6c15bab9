"""CPU suite: the oracle (oracle/qldpc_oracle.c) against the golden vectors the reference's own
source produced (tests/golden/make_golden.py).  This is what PINS the oracle."""
import numpy as np
import pytest

from conftest import assert_llr_close

VARIANT_KW = {
    "const": dict(alpha=0.8, alpha_mode="alvarado"),
    "damp": dict(alpha=1.0, alpha_mode="dynamical", damping=0.7),
    "clip5": dict(alpha=1.0, alpha_mode="dynamical", clip_llr=5.0),
    "dampclip": dict(alpha=0.9, alpha_mode="alvarado", damping=0.5, clip_llr=6.0),
}


def check_decode(out, g, key, exact=True):
    err, conv, llr, it = out
    assert np.array_equal(err, g[key + "_err"])
    assert np.array_equal(conv, g[key + "_conv"])
    assert np.array_equal(it, g[key + "_iter"])
    if exact:      # same operation order in strict IEEE f64 -> bit identical, stronger than the 1e-5 contract
        assert np.array_equal(llr, g[key + "_llr"], equal_nan=True)
    assert_llr_close(llr, g[key + "_llr"])


def test_steane_all_syndromes(oracle, golden):
    g = golden("steane_minsum")
    mi = int(g["max_iter"])
    modes = {
        "dyn": dict(alpha=1.0, alpha_mode="dynamical"),
        "const": dict(alpha=0.8, alpha_mode="alvarado"),
        "seq": dict(alpha=g["seq_alpha"], alpha_mode="alvarado-autoregressive"),
        "none0": dict(alpha=0, alpha_mode=None),
        "none1": dict(alpha=0.9, alpha_mode=None),
    }
    for tag, kw in modes.items():
        out = oracle.minsum_decode_batch(g["indptr"], g["indices"], 7, g["syndromes"], g["prior"], max_iter=mi, **kw)
        check_decode(out, g, tag)
    out = oracle.minsum_decode_batch(g["indptr"], g["indices"], 7, g["syndromes"], g["prior2"], max_iter=mi)
    check_decode(out, g, "p2")


@pytest.mark.parametrize("tag", ["bb72", "bb144", "bb288"])
def test_bb_code_capacity(oracle, golden, tag):
    g = golden(tag + "_minsum")
    for h in ("Hx", "Hz"):
        ip, ix = g[h + "_indptr"], g[h + "_indices"]
        n = int(g[h + "_shape"][1])
        for p in ("p005", "p030", "p080"):
            base = f"{h}_{p}"
            synd, prior = g[base + "_syndromes"], g[base + "_prior"]
            for mi in (1, 5, 50):
                out = oracle.minsum_decode_batch(ip, ix, n, synd, prior, max_iter=mi)
                check_decode(out, g, f"{base}_dyn_it{mi}")
            if h == "Hx":
                for v, kw in VARIANT_KW.items():
                    out = oracle.minsum_decode_batch(ip, ix, n, synd, prior, max_iter=30, **kw)
                    check_decode(out, g, f"{base}_{v}")
                out = oracle.minsum_decode_batch(ip, ix, n, synd, prior, max_iter=30, alpha=g["seq_alpha"],
                                                 alpha_mode="alvarado-autoregressive")
                check_decode(out, g, f"{base}_seq")


def test_golden_exercises_hard_cases(golden):
    """The fixtures must contain multi-iteration and non-converged decodes, not only 1-iteration ones."""
    g = golden("bb144_minsum")
    assert (g["Hx_p080_dyn_it50_conv"] == 0).any()
    assert (g["Hx_p030_dyn_it50_iter"] > 2).any()
    assert (g["Hx_p005_dyn_it50_conv"] == 1).all()


def test_core_passes(oracle, golden):
    g = golden("core_passes")
    m, n = (int(x) for x in g["shape"])
    ip, ix = g["indptr"], g["indices"]
    mask = np.zeros((m, n), bool)
    for i in range(m):
        mask[i, ix[ip[i]:ip[i + 1]]] = True
    for t in range(g["Q"].shape[0]):
        R, Rs = oracle.minsum_core_sparse(ip, ix, n, g["Q"][t], g["syndrome_sign"][t], float(g["alphas"][t]))
        assert np.array_equal(R, g["R_flat"][t], equal_nan=True)
        assert np.array_equal(Rs, g["R_sum"][t], equal_nan=True)
        Qd = np.zeros((m, n)); Qd[mask] = g["Q"][t]
        Rd = oracle.minsum_core_dense(Qd, g["syndrome_sign"][t], mask, float(g["alphas"][t]))
        assert np.array_equal(Rd, g["R_dense"][t], equal_nan=True)
    for t in range(g["bp_Q"].shape[0]):
        Qd = np.zeros((m, n)); Qd[mask] = g["bp_Q"][t]
        Rb = oracle.bp_core_dense(Qd, g["syndrome_sign"][t], mask, 0.9999999)
        assert_llr_close(Rb, g["bp_R_dense"][t], tol=1e-5)     # libm tanh/atanh vs numpy's: ulps apart
    sc = np.array([oracle.syndrome_check(ip, ix, c) for c in g["sc_candidates"]])
    assert np.array_equal(sc, g["sc_syndromes"])


def test_dense_and_bp_drivers(oracle, golden):
    g = golden("core_passes")
    m, n = (int(x) for x in g["shape"])
    ip, ix = g["indptr"], g["indices"]
    H = np.zeros((m, n))
    for i in range(m):
        H[i, ix[ip[i]:ip[i + 1]]] = 1.0
    for t, s in enumerate(g["bpdrv_syndromes"]):
        e, c, v, it = oracle.minsum_dense_driver(H, s, g["bpdrv_prior"], max_iter=12)
        assert np.array_equal(e, g["dense_err"][t]) and c == bool(g["dense_conv"][t]) and it == g["dense_iter"][t]
        assert np.array_equal(v, g["dense_llr"][t], equal_nan=True)
        e, c, v, it = oracle.bp_dense_driver(H, s, g["bpdrv_prior"], max_iter=12)
        assert np.array_equal(e, g["bpdrv_err"][t]) and c == bool(g["bpdrv_conv"][t]) and it == g["bpdrv_iter"][t]
        assert_llr_close(v, g["bpdrv_llr"][t], tol=1e-5)
    e, c, R, it = oracle.minsum_dense_driver(H, g["bpdrv_syndromes"][3], g["bpdrv_prior"], max_iter=12, alpha_estimation=True)
    assert it == 0 and c is False and not e.any()
    assert np.array_equal(R, g["alphaest_R"], equal_nan=True)


def test_dense_equals_sparse_entry_point(oracle, golden):
    """dense and sparse min-sum paths agree bit for bit."""
    g = golden("core_passes")
    m, n = (int(x) for x in g["shape"])
    e, c, v, it = oracle.minsum_decode_batch(g["indptr"], g["indices"], n, g["bpdrv_syndromes"], g["bpdrv_prior"], max_iter=12)
    assert np.array_equal(e, g["dense_err"]) and np.array_equal(c, g["dense_conv"]) and np.array_equal(it, g["dense_iter"])
    assert np.array_equal(v, g["dense_llr"], equal_nan=True)


def test_gf2_elimination(oracle, golden):
    g = golden("gf2_elimination")
    for tag in g["cases"]:
        A, b = g[f"{tag}_A"], g[f"{tag}_b"]
        Ar, br, pr, pc = oracle.gf2_elimination(A, b)
        assert np.array_equal(Ar, g[f"{tag}_A_red"]) and np.array_equal(br, g[f"{tag}_b_red"])
        assert np.array_equal(pr, g[f"{tag}_pivot_rows"]) and np.array_equal(pc, g[f"{tag}_pivot_cols"])
        P, bp, pr2, pc2 = oracle.gf2_elimination_packed(A, b)
        assert np.array_equal(P, g[f"{tag}_A_packed_red"]) and np.array_equal(bp, g[f"{tag}_b_red"])
        assert np.array_equal(pr2, pr) and np.array_equal(pc2, pc)


@pytest.mark.parametrize("tag", ["circ72", "circ144"])
def test_noise_kernels(oracle, golden, tag):
    g = golden(tag + "_noise")
    cap = int(g["max_circuit_size"])
    tq = int(g["total_qubits"])
    for t, p in enumerate(g["error_rates"]):
        L, oo, o1, o2 = oracle.generate_noisy_circuit(g["base_ops"], g["base_q1"], g["base_q2"], float(p),
                                                      g["random_vals"][t], g["random_paulis"][t], g["random_two_qubit"][t], cap)
        assert L == g["noisy_len"][t]
        assert np.array_equal(oo[:L], g["noisy_ops"][t][:L]) and np.array_equal(o1[:L], g["noisy_q1"][t][:L])
        assert np.array_equal(o2[:L], g["noisy_q2"][t][:L])
        ops = np.concatenate([oo[:L], g["suffix_ops"]]); q1 = np.concatenate([o1[:L], g["suffix_q1"]])
        q2 = np.concatenate([o2[:L], g["suffix_q2"]])
        hz, sz, ncz, ecz = oracle.simulate_circuit("Z", ops, q1, q2, tq, int(g["max_syndromes_x"]))
        hx, sx, ncx, ecx = oracle.simulate_circuit("X", ops, q1, q2, tq, int(g["max_syndromes_z"]))
        assert np.array_equal(hz, g["hist_z"][t]) and np.array_equal(sz, g["state_z"][t])
        assert np.array_equal(hx, g["hist_x"][t]) and np.array_equal(sx, g["state_x"][t])
        assert [ncz, ecz, ncx, ecx] == g["counts"][t].tolist()
        a, b, c, d = oracle.run_trial(g, float(p), g["random_vals"][t], g["random_paulis"][t], g["random_two_qubit"][t])
        assert np.array_equal(a, g["sparse_z"][t]) and np.array_equal(b, g["true_z"][t])
        assert np.array_equal(c, g["sparse_x"][t]) and np.array_equal(d, g["true_x"][t])
    assert (g["sparse_z"].sum(axis=1) > 0).all()


@pytest.mark.parametrize("tag", ["circ72", "circ144"])
def test_circuit_level_decode_and_osd(oracle, golden, tag):
    import os
    from conftest import ROOT
    g = golden(tag + "_decode")
    with np.load(os.path.join(ROOT, "qldpc-branched-off_amd", "data", f"{tag}_p005.npz")) as d:
        data = {k: d[k] for k in d.files}
    for s in ("Z", "X"):
        llr0 = oracle.prior_llrs(data[f"channel_probs{s}"])
        assert np.array_equal(llr0, g[f"llrs_{s}"])                       # a15 incl. p_j > 1 -> 0
        ip, ix = data[f"Hdec{s}_indptr"], data[f"Hdec{s}_indices"]
        n = int(data[f"Hdec{s}_shape"][1])
        out = oracle.minsum_decode_batch(ip, ix, n, g[f"{s}_syndromes"], llr0, max_iter=int(g["max_iter"]))
        check_decode(out, g, s)
        for t, case in enumerate(g[f"{s}_osd_cases"]):
            sol = oracle.osd0(ip, ix, n, g[f"{s}_syndromes"][case], g[f"{s}_llr"][case], g[f"{s}_err"][case],
                              ordering=g[f"{s}_osd_ordering"][t])
            assert np.array_equal(sol, g[f"{s}_osd_solution"][t])
            # the OSD-0 answer satisfies the syndrome
            assert np.array_equal(oracle.syndrome_check(ip, ix, sol), g[f"{s}_syndromes"][case])
    if tag == "circ144":
        assert (data["channel_probsZ"] > 1).sum() == 1 and (g["llrs_Z"] == 0).sum() >= 1   # log(negative) -> NaN -> 0
    assert np.isinf(g["X_llr"]).any()           # degree-1 checks -> +-inf posteriors (SURVEY hard parts)


def test_philox_circuit_trial_equals_explicit_random_path(oracle, golden):
    """The oracle's Philox-driven trial (the checker of the GPU sampler) == the golden-pinned explicit-random composition
    when the random arrays are derived from the same Philox words."""
    g = golden("circ72_noise")
    circ = oracle.make_circuit(g, g["Lx"], g["Lz"])
    n_locs = int(g["num_error_locs"])
    for p, seed, trial in ((0.005, 7, 3), (0.05, 11, 123456789012)):
        thr = oracle.bernoulli_threshold(p)
        rv, rp, rt = np.ones(n_locs), np.zeros(n_locs, np.int32), np.zeros(n_locs, np.int32)
        for l in range(n_locs):
            o = oracle.philox([trial & 0xFFFFFFFF, trial >> 32, l >> 2, 1], [seed, 0])
            if o[l & 3] < thr:
                w = oracle.philox([trial & 0xFFFFFFFF, trial >> 32, l, 2], [seed, 0])
                rv[l], rp[l], rt[l] = 0.0, w[0] % 3, w[0] % 15
        want = oracle.run_trial(g, p, rv, rp, rt)
        got = oracle.circuit_sample(circ, p, seed, trial)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        assert (rv == 0).sum() > 0


def test_estimator_trial_loops(oracle, golden):
    """f4: the oracle's restatement of the alpha / beta estimator trial loops (alpha.py:119-137, 206-255; scopt.py:80-134), followed by
    the reference's own numpy post-processing, reproduces the histograms and fitted factors the reference produced."""
    from conftest import estimator_cases, reference_fit
    seen = set()
    for c in estimator_cases(golden("estimators")):
        seen.add(c["kind"])
        cols = np.asarray(c["indices"])
        if c["kind"] == "alvarado":
            R = oracle.alpha_messages(c["indptr"], c["indices"], c["n"], c["errors"], c["prior"])
            a, r2, h0, h1, edges = reference_fit(R.ravel(), c["errors"][:, cols].ravel(), c["bins"])
            assert np.array_equal(h0, c["hist"][0]) and np.array_equal(h1, c["hist"][1]) and np.array_equal(edges, c["edges"][0]), c["name"]
            assert abs(a - c["out"]["alpha"]) <= 1e-9 and abs(r2 - c["out"]["r2"]) <= 1e-9
        elif c["kind"] == "autoregressive":
            alphas = []
            for k in range(c["iters"]):
                E = c["errors"][k * c["trials"]:(k + 1) * c["trials"]]
                R = oracle.alpha_messages(c["indptr"], c["indices"], c["n"], E, c["prior"], alpha_prev=alphas, damping=c["damping"], clip_llr=c["clip"])
                a, r2, h0, h1, edges = reference_fit(R.ravel(), E[:, cols].ravel(), c["bins"])
                assert np.array_equal(h0, c["hist"][2 * k]) and np.array_equal(h1, c["hist"][2 * k + 1]) and np.array_equal(edges, c["edges"][k]), (c["name"], k)
                assert abs(a - c["out"]["alpha"][k]) <= 1e-9 and abs(r2 - c["out"]["r2"][k]) <= 1e-9
                alphas.append(float(c["out"]["alpha"][k]))          # the reference's own value, so later iterations see identical inputs
        else:
            V = oracle.scopt_values(c["indptr"], c["indices"], c["n"], c["errors"], c["prior"], max_iter=c["iters"], alpha=c["alpha"],
                                    alpha_mode=c["alpha_mode"], damping=c["damping"], clip_llr=c["clip"])
            b, r2, h0, h1, edges = reference_fit(V.ravel(), c["errors"].ravel(), c["bins"], flip=True)
            assert np.array_equal(h0, c["hist"][0]) and np.array_equal(h1, c["hist"][1]) and np.array_equal(edges, c["edges"][0]), c["name"]
            assert abs(b - c["out"]["beta"]) <= 1e-9 and abs(r2 - c["out"]["r2"]) <= 1e-9
    assert seen == {"alvarado", "autoregressive", "scopt"}


def test_osd_order_w_sweep(oracle, golden):
    """f1: performOSD_enhanced with order > 0 (osd.py:31-75: the combination sweep that runs when OSD-0 misses the syndrome)."""
    from conftest import osdw_cases
    cases = osdw_cases(golden("osdw"))
    for c in cases:
        for ordering in (c["ordering"], None):          # the fixtures have no |llr| ties, so the default order must agree too
            sol = oracle.osdw(c["indptr"], c["indices"], c["n"], c["syndrome"], c["llr"], c["hard"], c["order"], c["maxc"], ordering=ordering)
            assert np.array_equal(sol, c["solution"]), (c["name"], c["order"], c["maxc"])
        if not c["differs"]:
            assert np.array_equal(oracle.osd0(c["indptr"], c["indices"], c["n"], c["syndrome"], c["llr"], c["hard"]), c["solution"])
    assert sum(c["differs"] for c in cases) >= 10 and any(not c["swept"] for c in cases) and any(c["maxc"] for c in cases)


def test_bb_256_syndromes_per_point(oracle, golden):
    """>= 256 reference-decoded syndromes per (code, p) point, decoder defaults: the oracle's posteriors are bit-identical."""
    from qldpc_amd.data import load_code
    g = golden("bb_256")
    for tag in ("bb72", "bb144", "bb288"):
        c = load_code(tag)
        n = int(g[f"{tag}_shape"][1])
        for p in (0.005, 0.02, 0.05):
            k = f"{tag}_p{int(round(p * 1000)):03d}"
            errs = np.unpackbits(g[f"{k}_errors"], axis=1, bitorder="little")[:, :n].astype(np.int8)
            hard = np.unpackbits(g[f"{k}_hard"], axis=1, bitorder="little")[:, :n].astype(np.int8)
            synd = np.stack([oracle.syndrome_check(c["Hx_indptr"], c["Hx_indices"], e) for e in errs])
            err, conv, llr, it = oracle.minsum_decode_batch(c["Hx_indptr"], c["Hx_indices"], n, synd, np.full(n, np.log((1 - p) / p)), max_iter=50, threads=0)
            assert np.array_equal(err, hard) and np.array_equal(conv.astype(bool), g[f"{k}_conv"].astype(bool))
            assert np.array_equal(it, g[f"{k}_iter"]) and np.array_equal(llr, g[f"{k}_llr"])


def test_gf2_elimination_production_size(oracle, golden):
    """1008 x 8785 LLR-ordered instance of gf2_elimination_packed (kernels.py:48-106): reduced matrix, rhs and pivots of the reference."""
    from qldpc_amd.data import load_circuit_matrices
    g = golden("gf2_big")
    d = load_circuit_matrices("circ144")
    m, n = (int(x) for x in d["HdecZ_shape"])
    H = np.zeros((m, n), np.uint8)
    ip, ix = d["HdecZ_indptr"], d["HdecZ_indices"]
    for i in range(m):
        H[i, ix[ip[i]:ip[i + 1]]] = 1
    P, b, pr, pc = oracle.gf2_elimination_packed(H[:, g["ordering"].astype(np.int64)], g["b"])
    assert np.array_equal(pr, g["pivot_rows"]) and np.array_equal(pc, g["pivot_cols"]) and np.array_equal(b, g["b_red"])
    assert np.array_equal(P, g["A_packed_red"])


def _osd0_truncated(H, s, llr, hard, order):
    """OSD-0 (osd.py:5-29 with the first-row pivot rule of kernels.py:66-92) that ENDS its sweep as soon as the reduced right-hand side has no one left at or
    below the diagonal -- the shortcut the HIP kernels take.  Returns (solution, columns swept, pivots, whether the sweep ended on the test)."""
    m, n = H.shape
    A = H[:, order].astype(np.uint8) % 2
    b = (s.astype(np.int64) + H.astype(np.int64) @ hard.astype(np.int64)) % 2
    b = b.astype(np.uint8)
    rank, pivcols, stopped, swept = 0, [], False, 0
    for c in range(n):
        if not b[rank:].any():
            stopped = True
            break
        if rank >= m:
            break
        swept = c + 1
        rows = np.flatnonzero(A[rank:, c]) + rank
        if rows.size == 0:
            continue
        p = rows[0]
        if p != rank:
            A[[p, rank]] = A[[rank, p]]
            b[[p, rank]] = b[[rank, p]]
        for r in np.flatnonzero(A[:, c]):
            if r != rank:
                A[r] ^= A[rank]
                b[r] ^= b[rank]
        pivcols.append(c)
        rank += 1
    stopped = stopped or not b[rank:].any()                      # (gone with the last column swept)
    e = np.zeros(n, np.uint8)
    for t, c in enumerate(pivcols):
        e[order[c]] = b[t]
    return ((hard.astype(np.uint8) + e) % 2).astype(np.int8), swept, rank, stopped


def test_osd0_sweep_may_end_when_the_residual_is_gone(oracle):
    """The identity behind the round-4 OSD-0 kernels, checked on the CPU against the literal port of the reference: once no row at or below the diagonal holds a
    one of the reduced right-hand side, the rest of the sweep changes nothing -- for a realisable syndrome the truncated sweep gives the reference's solution
    (usually after a fraction of the columns); for a syndrome outside the column space the test never fires and the sweep runs to its end."""
    rng = np.random.default_rng(20261005)
    fired = early = 0
    for trial in range(300):
        m, n = int(rng.integers(3, 24)), int(rng.integers(4, 60))
        H = (rng.random((m, n)) < rng.choice([0.08, 0.2, 0.45])).astype(np.int8)
        if m > 4 and trial % 3 == 0:
            H[2] = H[0] ^ H[1]                                   # dependent rows
        ip = np.concatenate([[0], np.cumsum(H.sum(1))]).astype(np.int32)
        ix = np.concatenate([np.flatnonzero(r) for r in H]).astype(np.int32) if H.any() else np.zeros(0, np.int32)
        llr = np.round(rng.normal(0, 2, n), int(rng.integers(0, 3)))             # ties
        hard = (rng.random(n) < 0.15).astype(np.int8)
        order = np.argsort(np.abs(llr), kind="stable")
        consistent = trial % 4 != 0
        s = (H.astype(np.int64) @ (rng.random(n) < 0.2).astype(np.int64) % 2).astype(np.int8) if consistent else (rng.random(m) < 0.5).astype(np.int8)
        ref = oracle.osd0(ip, ix, n, s, llr, hard)
        sol, swept, rank, stopped = _osd0_truncated(H, s, llr, hard, order)
        realisable = np.array_equal((H.astype(np.int64) @ ref.astype(np.int64)) % 2, s % 2)
        if realisable:
            assert stopped and np.array_equal(sol, ref), (trial, m, n)
            fired += 1
            early += swept < n
        else:
            assert not stopped, (trial, m, n)                    # a one in an unused row survives to the end
            assert np.array_equal(sol, ref), (trial, m, n)       # (the full sweep, literally)
    assert fired > 150 and early > 100
