#!/usr/bin/env python3
"""Randomised differential soak: GPU entry points vs the CPU checker on seeded random configurations for a time budget.
  python tools/soak.py [--seconds 240] [--seed 1]      (exit code 1 and a repro line on the first mismatch)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
import qldpc_amd  # noqa: F401,E402
from qldpc_amd import _lib as L  # noqa: E402
from qldpc_amd.data import load_code, load_circuit_matrices  # noqa: E402
from oracle import oracle  # noqa: E402  (checker)

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=240)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
codes = {t: load_code(t) for t in ("bb72", "bb90", "bb108", "bb144", "bb288", "steane")}
circ = {t: load_circuit_matrices(t) for t in ("circ72", "circ144")}
graphs = {}


def graph(key, ip, ix, n):
    if key not in graphs:
        graphs[key] = L.Graph(ip, ix, n)
    return graphs[key]


def fail(msg):
    print("MISMATCH", msg, flush=True)
    sys.exit(1)


t_end = time.time() + a.seconds
t_note = time.time() + 60
count = {"decode": 0, "tally": 0, "plan": 0, "osd": 0, "circuit": 0, "stats": 0, "osdw": 0}
gold = {}
for t in ("circ72",):
    with np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", t + "_noise.npz")) as z:
        gold[t] = {k: z[k] for k in z.files}
cstate = {}


def circuit_setup(tag):
    if tag not in cstate:
        g, d = gold[tag], circ[tag]
        co = oracle.make_circuit(g, g["Lx"], g["Lz"])
        secs, grs, prs, mks = [], [], [], []
        for s2 in "ZX":
            n2 = int(d[f"Hdec{s2}_shape"][1])
            pr = oracle.prior_llrs(d[f"channel_probs{s2}"])
            secs.append(oracle.make_sector(d[f"Hdec{s2}_indptr"], d[f"Hdec{s2}_indices"], n2, pr, d[f"H{s2}_logical_indptr"], d[f"H{s2}_logical_indices"]))
            grs.append(L.Graph(d[f"Hdec{s2}_indptr"], d[f"Hdec{s2}_indices"], n2)); prs.append(pr)
            mks.append(L.logical_column_masks((d[f"H{s2}_logical_indptr"], d[f"H{s2}_logical_indices"]), n2))
        cstate[tag] = (g, co, secs, grs, prs, mks)
    return cstate[tag]

while time.time() < t_end:
    if time.time() > t_note:
        print(f"  ... {count}", flush=True)          # progress line (a silent GPU job is taken to be hung)
        t_note = time.time() + 60
    kind = rng.choice(["decode", "decode", "tally", "plan", "osd", "circuit", "stats", "osdw"])
    if kind == "stats":                                   # estimator trial loops (f4): range, finite counts, both histograms
        tag = str(rng.choice(["bb72", "bb144", "steane", "bb90"]))
        c = codes[tag]; ip, ix, n = c["Hx_indptr"], c["Hx_indices"], int(c["n"])
        g = graph(tag, ip, ix, n)
        B = int(rng.integers(20, 400)); p = float(rng.choice([0.01, 0.04, 0.1]))
        E = (rng.random((B, n)) < p).astype(np.int8)
        prior = np.log((1 - p) / p) + rng.normal(0, 0.6, n)
        bins = int(rng.integers(5, 80))
        if rng.random() < 0.5:
            prev = list(rng.uniform(0.4, 1.0, int(rng.integers(0, 5))))
            damping = float(rng.choice([1.0, 0.8])); clip = float(rng.choice([20.0, 7.0]))
            smp = oracle.alpha_messages(ip, ix, n, E, prior, alpha_prev=prev, damping=damping, clip_llr=clip).ravel()
            bits = E[:, np.asarray(ix)].ravel()
            st = L.MessageStats(g, E, prior, L.STATS_CHECK_MESSAGES, len(prev), alpha_mode="alvarado-autoregressive" if prev else "dynamical",
                                alpha=np.array(prev) if prev else 1.0, damping=damping, clip_llr=clip)
        else:
            iters = int(rng.integers(1, 30))
            smp = oracle.scopt_values(ip, ix, n, E, prior, max_iter=iters).ravel()
            bits = E.ravel()
            st = L.MessageStats(g, E, prior, L.STATS_POSTERIOR, iters)
        fin = np.isfinite(smp)
        ok = st.finite == (int((fin & (bits == 0)).sum()), int((fin & (bits == 1)).sum()))
        if ok and fin.any():
            ok = st.range == (smp[fin].min(), smp[fin].max())
            edges = np.histogram_bin_edges(np.zeros(0), bins=bins, range=st.range)
            h0, h1 = st.histogram(edges)
            ok = ok and np.array_equal(h0, np.histogram(smp[fin & (bits == 0)], bins=bins, range=st.range)[0]) and \
                np.array_equal(h1, np.histogram(smp[fin & (bits == 1)], bins=bins, range=st.range)[0])
        st.close()
        if not ok:
            fail(f"stats {tag} B={B} p={p} bins={bins} seed={a.seed} n={count}")
        count[kind] += 1
        continue
    if kind == "osdw":                                    # OSD-w sweep (f1) on small matrices with dependent rows
        import ctypes as C
        m2, n2 = int(rng.integers(3, 14)), int(rng.integers(6, 30))
        Hd = (rng.random((m2, n2)) < 0.3).astype(np.int8)
        if m2 > 3:
            Hd[m2 // 2] = Hd[0] ^ Hd[1]
        ip2, ix2, _ = L.canonical_csr(Hd)
        g2 = L.Graph(ip2, ix2, n2)
        synd = (rng.random(m2) < 0.5).astype(np.int8); llr = rng.normal(0, 3, n2); hard = (rng.random(n2) < 0.15).astype(np.int8)
        order = int(rng.integers(1, 5)); maxc = int(rng.choice([0, 0, 3, 17]))
        sol = np.zeros((1, n2), np.int8)
        L.check(L.lib().qldpc_osdw_batch(g2.handle, C.c_int64(1), L.ptr(synd.reshape(1, -1), C.c_int8), L.ptr(llr.reshape(1, -1), C.c_double),
                                         L.ptr(hard.reshape(1, -1), C.c_int8), None, C.c_int(order), C.c_int64(maxc), L.ptr(sol, C.c_int8)))
        want = oracle.osdw(ip2, ix2, n2, synd, llr, hard, order, maxc or None)
        if not np.array_equal(sol[0], want):
            fail(f"osdw m={m2} n={n2} order={order} maxc={maxc} seed={a.seed} n={count}")
        count[kind] += 1
        continue
    if kind == "circuit":
        g, co, secs, grs, prs, mks = circuit_setup("circ72")
        p = float(rng.choice([0.001, 0.003, 0.005, 0.01])); iters = int(rng.integers(1, 45)); N = int(rng.integers(1, 120))
        seed = int(rng.integers(0, 2 ** 62)); begin = int(rng.integers(0, 10 ** 8)); use_osd = bool(rng.random() < 0.85)
        mode, alpha = [("dynamical", 1.0), ("alvarado", float(rng.uniform(0.4, 1.0)))][int(rng.integers(0, 2))]
        plan = L.CircuitPlan(g, g["Lx"], g["Lz"], grs[0], grs[1], prs[0], prs[1], mks[0], mks[1], p, max_iter=iters, alpha_z=alpha, alpha_x=alpha,
                             alpha_mode=mode, use_osd=use_osd, batch=int(rng.choice([16, 64, 4096])))
        plan.run(seed, begin, N)
        got = plan.read()
        plan.close()
        want = oracle.circuit_sample_decode_tally(co, secs[0], secs[1], p, seed, begin, N, max_iter=iters, alpha=alpha, alpha_mode=mode, use_osd=use_osd, threads=0)
        if not np.array_equal(got, want):
            fail(f"circuit p={p} iters={iters} N={N} seed={seed} begin={begin} osd={use_osd} mode={mode} alpha={alpha} got={got.tolist()} want={want.tolist()}")
        count[kind] += 1
        continue
    if kind == "decode":
        if rng.random() < 0.7:
            tag = str(rng.choice(list(codes)))
            c = codes[tag]; ip, ix, n = c["Hx_indptr"], c["Hx_indices"], int(c["n"])
        else:
            tag = str(rng.choice(list(circ))); s = str(rng.choice(["Z", "X"]))
            d = circ[tag]; ip, ix, n = d[f"Hdec{s}_indptr"], d[f"Hdec{s}_indices"], int(d[f"Hdec{s}_shape"][1]); tag += s
        g = graph(tag, ip, ix, n)
        B = int(rng.integers(1, 300 if n < 1000 else 24))
        p = float(rng.choice([0.002, 0.01, 0.04, 0.09]))
        E = (rng.random((B, n)) < p).astype(np.int8)
        synd = np.stack([oracle.syndrome_check(ip, ix, e) for e in E])
        # uniform / three-valued priors take the LDS-resident workgroup decoder on the circuit-level matrices (classes by (degree, prior)), noisy ones its fall-back
        shape = rng.random()
        prior = np.full(n, np.log((1 - p) / p)) + (rng.normal(0, 0.5, n) if shape < 0.35 else (rng.choice([0.0, 1.25, -0.75], n) if shape < 0.7 else 0.0))
        if rng.random() < 0.15:
            prior[rng.integers(0, n)] = rng.choice([0.0, -0.0, np.inf, -np.inf, np.nan])
        mode, alpha = [("dynamical", 1.0), ("alvarado", float(rng.uniform(0.3, 1.1))), ("alvarado-autoregressive", rng.uniform(0.3, 1.0, int(rng.integers(1, 6))))][int(rng.integers(0, 3))]
        damping = float(rng.choice([1.0, 1.0, 0.9, 0.5]))
        clip = float(rng.choice([20.0, 5.0, 50.0]))
        iters = int(rng.integers(1, 60))
        flags = int(rng.choice([0, L.FLAG_FIXED_ITERS, L.FLAG_KERNEL_STREAM, L.FLAG_KERNEL_GENERIC if n < 1000 else 0]))
        env = rng.random() < 0.2 and n > 1000
        if env:
            flags |= L.FLAG_WG_VGLOBAL
        if n > 1000 and rng.random() < 0.3:
            flags |= int(rng.choice([L.FLAG_WG_ROWMAJOR, L.FLAG_WG_GENERIC, L.FLAG_WG_TABLES]))
        try:
            out = L.minsum_decode_batch(g, synd, prior, iters, mode, alpha, damping=damping, clip_llr=clip, flags=flags)
        except L.QldpcError as e:
            if "does not support" in str(e):
                continue
            raise
        ref = oracle.minsum_decode_batch(ip, ix, n, synd, prior, max_iter=iters, alpha=alpha, alpha_mode=mode, damping=damping, clip_llr=clip)
        for nm, x, y in zip(("err", "conv", "llr", "iter"), (out[0], out[1].astype(bool), out[2], out[3]), (ref[0], ref[1].astype(bool), ref[2], ref[3])):
            if not np.array_equal(x, y, equal_nan=True):
                fail(f"decode {tag} B={B} p={p} mode={mode} alpha={alpha} damping={damping} clip={clip} iters={iters} flags={flags} vg={env} field={nm} seed={a.seed} n={count}")
    elif kind == "plan":            # a plan over several batches: concurrent small batches / the side-stream tail, several run() calls before one read()
        tag = str(rng.choice(["bb72", "bb90", "bb108", "bb144", "bb288"]))
        c = codes[tag]; ip, ix, n = c["Hx_indptr"], c["Hx_indices"], int(c["n"])
        g = graph(tag, ip, ix, n)
        p = float(rng.choice([0.003, 0.01, 0.03, 0.08])); iters = int(rng.integers(1, 55)); use_osd = bool(rng.random() < 0.8)
        batch = int(rng.choice([257, 1000, 4096, 30000, 40000])); seed = int(rng.integers(0, 2 ** 62))
        flags = int(rng.choice([0, 0, L.FLAG_FIXED_ITERS, L.FLAG_KERNEL_GENERIC, L.FLAG_KERNEL_GENERIC | L.FLAG_FIXED_ITERS]))
        L.set_option("mc_first_iteration", int(rng.random() < 0.8)); L.set_option("mc_tail_overlap", int(rng.random() < 0.8))
        plan = L.CodeCapacityPlan(g, c["Lx"], p, max_iter=iters, use_osd=use_osd, flags=flags, batch=batch,
                                  min_launch=int(rng.choice([0, 0, 0, 32768, 2048])))       # mostly the batch taken literally: several pieces per call
        want = np.zeros(16, np.int64); begin = int(rng.integers(0, 10 ** 9))
        for _ in range(int(rng.integers(1, 4))):
            N = int(rng.integers(1, 3 * batch + 2))
            plan.run(seed, begin, N)
            want += oracle.cc_sample_decode_tally(ip, ix, n, c["Lx"], p, seed, begin, N, max_iter=iters, use_osd=use_osd, threads=0)
            begin += N
        got = plan.read()
        plan.close()
        L.set_option("mc_first_iteration", 1); L.set_option("mc_tail_overlap", 1)
        if not np.array_equal(got, want):
            fail(f"plan {tag} p={p} batch={batch} seed={seed} iters={iters} osd={use_osd} flags={flags} got={got.tolist()} want={want.tolist()}")
    elif kind == "tally":
        tag = str(rng.choice(["bb72", "bb90", "bb108", "bb144", "bb288", "steane"]))
        c = codes[tag]; ip, ix, n = c["Hx_indptr"], c["Hx_indices"], int(c["n"])
        g = graph(tag, ip, ix, n)
        p = float(rng.choice([0.003, 0.01, 0.03, 0.06]))
        N = int(rng.integers(1, 8000)); begin = int(rng.integers(0, 10 ** 9)); seed = int(rng.integers(0, 2 ** 62))
        iters = int(rng.integers(1, 55)); use_osd = bool(rng.random() < 0.8)
        flags = int(rng.choice([0, L.FLAG_FIXED_ITERS, L.FLAG_MC_UNFUSED, L.FLAG_MC_UNFUSED | L.FLAG_KERNEL_STREAM]))
        got = L.cc_sample_decode_tally(g, c["Lx"], p, seed, begin, N, max_iter=iters, use_osd=use_osd, flags=flags)
        want = oracle.cc_sample_decode_tally(ip, ix, n, c["Lx"], p, seed, begin, N, max_iter=iters, use_osd=use_osd, threads=0)
        if not np.array_equal(got, want):
            fail(f"tally {tag} p={p} N={N} begin={begin} seed={seed} iters={iters} osd={use_osd} flags={flags} got={got.tolist()} want={want.tolist()}")
    else:
        import ctypes as C
        tag = str(rng.choice(list(circ))); s = str(rng.choice(["Z", "X"]))
        d = circ[tag]; ip, ix, n = d[f"Hdec{s}_indptr"], d[f"Hdec{s}_indices"], int(d[f"Hdec{s}_shape"][1])
        g = graph(tag + s, ip, ix, n); m = len(ip) - 1
        B = 3
        E = (rng.random((B, n)) < 0.01).astype(np.int8)
        synd = np.stack([oracle.syndrome_check(ip, ix, e) for e in E])
        if rng.random() < 0.3:
            synd[0] = rng.random(m) < 0.5
        llr = rng.normal(3, 4, (B, n))
        if rng.random() < 0.5:
            llr = np.round(llr)
        hard = (rng.random((B, n)) < 0.01).astype(np.int8)
        if rng.random() < 0.3:
            llr = np.clip(llr, -4.0, 4.0)                    # many columns at the clip bound: long runs of equal keys
        variants = [0, 0, 0, L.FLAG_OSD_UG, L.FLAG_OSD_UG | L.FLAG_OSD_REFORDER, L.FLAG_OSD_LDS, L.FLAG_OSD_REFORDER] + ([L.FLAG_OSD_GLOBAL] if tag == "circ72" else [])
        env = int(rng.choice(variants))
        sol = L.osd0_batch(g, synd, llr, hard, flags=env)
        for b in range(B):
            if not np.array_equal(sol[b], oracle.osd0(ip, ix, n, synd[b], llr[b], hard[b])):
                fail(f"osd {tag}{s} env={env} b={b} seed={a.seed} n={count}")
    count[kind] += 1
print(f"soak ok: {count} cases in {a.seconds:.0f}s (seed {a.seed})", flush=True)
