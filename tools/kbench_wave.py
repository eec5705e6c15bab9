#!/usr/bin/env python3
"""Wave-private vs team kernel of the code-capacity decoder (GPU): parity against the oracle and against each other, then kernel times.
usage: python tools/kbench_wave.py [--code bb144] [--batch N] [--steps K] [--variants cpl:rst:grid,...] [--skip-parity]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import qldpc_amd  # noqa: F401,E402
from qldpc_amd import _lib  # noqa: E402
from qldpc_amd.data import load_code  # noqa: E402

_lib.select_build("experiments")          # the wave-private kernel lives in libqldpc_hip_experiments.so (make -C csrc experiments)

ap = argparse.ArgumentParser()
ap.add_argument("--code", default="bb144")
ap.add_argument("--batch", type=int, default=1 << 20)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--p", type=float, default=0.005)
ap.add_argument("--max-iter", type=int, default=50)
ap.add_argument("--variants", default="0:0:0")
ap.add_argument("--skip-parity", action="store_true")
a = ap.parse_args()
c = load_code(a.code)
n, m = c["n"], c["m"]
g = _lib.Graph(c["Hx_indptr"], c["Hx_indices"], n)


def setopt(kernel, cpl=0, rst=0, grid=0):
    _lib.set_option("regular_kernel", kernel)
    _lib.set_option("wave_cpl", cpl)
    _lib.set_option("wave_rst", rst)
    _lib.set_option("wave_grid", grid)


variants = [tuple(int(x) for x in v.split(":")) for v in a.variants.split(",")]
if not a.skip_parity:
    from oracle import oracle as orc
    for p in (0.03, 0.08):
        rng = np.random.default_rng(5)
        errs = (rng.random((3000, n)) < p).astype(np.int8)
        synd = np.array([orc.syndrome_check(c["Hx_indptr"], c["Hx_indices"], e) for e in errs])
        prior = np.full(n, np.log((1 - p) / p))
        prior[::7] *= 0.9                      # a non-uniform prior exercises the general iteration 0
        for mi in (0, 1, 7, 50):
            ref = orc.minsum_decode_batch(c["Hx_indptr"], c["Hx_indices"], n, synd, prior, max_iter=mi)
            for (cpl, rst, grid) in variants:
                for fl in (0, _lib.FLAG_FIXED_ITERS):
                    setopt(2, cpl, rst, grid)
                    out = _lib.minsum_decode_batch(g, synd, prior, mi, "dynamical", 1.0, flags=fl)
                    for x, y, what in zip(out, ref, ("errors", "converged", "llr", "iterations")):
                        if not np.array_equal(x, y, equal_nan=(what == "llr")):
                            bad = np.argwhere(np.asarray(x) != np.asarray(y))[:5].tolist()
                            raise SystemExit(f"PARITY FAIL decode p={p} max_iter={mi} cpl={cpl} rst={rst} flags={fl}: {what} differs at {bad}")
        print(f"decode parity ok p={p}: wave kernel == oracle bit for bit (3000 syndromes x max_iter 0/1/7/50 x fixed/early x {len(variants)} variants)", flush=True)
    for p in (0.005, 0.04):
        setopt(1)
        t_team = _lib.cc_sample_decode_tally(g, c["Lx"], p, 77, 1000, 300000, max_iter=a.max_iter)
        for (cpl, rst, grid) in variants:
            for fl in (0, _lib.FLAG_FIXED_ITERS):
                setopt(2, cpl, rst, grid)
                t_wave = _lib.cc_sample_decode_tally(g, c["Lx"], p, 77, 1000, 300000, max_iter=a.max_iter, flags=fl)
                if not np.array_equal(t_team, t_wave):
                    raise SystemExit(f"PARITY FAIL tally p={p} cpl={cpl} rst={rst} flags={fl}: team {t_team.tolist()} wave {t_wave.tolist()}")
        print(f"Monte-Carlo parity ok p={p}: tallies identical to the team kernel's: {t_team[:9].tolist()}", flush=True)

for mode in ("fixed", "ref"):
    fl = _lib.FLAG_FIXED_ITERS if mode == "fixed" else 0
    for (kernel, cpl, rst, grid) in [(1, 0, 0, 0)] + [(2,) + v for v in variants]:
        setopt(kernel, cpl, rst, grid)
        plan = _lib.CodeCapacityPlan(g, c["Lx"], a.p, max_iter=a.max_iter, flags=fl, batch=a.batch)
        plan.run(1, 0, a.batch); plan.read(clear=True); plan.kernel_time()
        t0 = time.perf_counter()
        for k in range(a.steps):
            plan.run(2, k * a.batch, a.batch)
        t = plan.read()
        dt = time.perf_counter() - t0
        ms, nl = plan.kernel_time()
        name = "team" if kernel == 1 else f"wave cpl={cpl} rst={rst} grid={grid}"
        print(f"{a.code} {mode:5s} {name:28s} decode {ms / nl:8.3f} ms/launch  pipeline {dt / a.steps * 1e3:8.3f} ms/step  -> "
              f"{a.batch * a.steps / dt / 1e6:8.2f} Mshots/s  tally={t[:9].tolist()}", flush=True)
        plan.close()
setopt(0)
