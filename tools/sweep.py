#!/usr/bin/env python3
"""The experiment loop of the reference's main.py (codes x physical error rates -> logical error rates) on the GPU path, without the
plots: cache key -> load or build the decoding matrices -> run_simulation with the reference's own knobs (maxIter = 20, osd_order = 2,
alpha_mode = 'alvarado-autoregressive', stop at a target number of logical errors).
  python tools/sweep.py [--codes bb72,bb144] [--rates 0.006,0.005,0.004] [--target 200] [--max-trials 200000] [--cache-dir DIR]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
import qldpc_amd  # noqa: F401,E402
from qldpc_amd.data import load_code  # noqa: E402
from qldpc_amd.codes.bb_code import BBCodeCircuit  # noqa: E402
from qldpc_amd.noise.builder import build_decoding_matrices  # noqa: E402
from qldpc_amd.simulation.engine import run_simulation  # noqa: E402
from qldpc_amd.utils.caching import compute_cache_key, load_matrices, save_matrices  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--codes", default="bb72,bb90,bb108,bb144")
ap.add_argument("--rates", default="0.006,0.005,0.004")
ap.add_argument("--target", type=int, default=200)
ap.add_argument("--max-trials", type=int, default=200000)
ap.add_argument("--max-iter", type=int, default=20)
ap.add_argument("--alpha-mode", default="alvarado-autoregressive")
ap.add_argument("--cache-dir", default=None)
a = ap.parse_args()
print(f"alpha_mode={a.alpha_mode} maxIter={a.max_iter} osd_order=2 target_logical_errors={a.target} max_trials={a.max_trials}")
print(f"{'code':8s} {'cycles':>6s} {'p':>8s} {'LER':>11s} {'trials':>8s} {'errors':>7s} {'alpha_z[0:3]':>22s} {'build s':>8s} {'run s':>7s} {'trials/s':>9s}")
for tag in a.codes.split(","):
    c = load_code(tag)
    cycles = int(c["distance"])
    bb = dict(ell=c["ell"], m=c["m_dim"], a_x_powers=c["a_x_powers"], a_y_powers=c["a_y_powers"], b_y_powers=c["b_y_powers"], b_x_powers=c["b_x_powers"])
    cb = BBCodeCircuit(c["Hx"], c["Hz"], num_cycles=cycles, **bb)
    for p in (float(x) for x in a.rates.split(",")):
        t0 = time.perf_counter()
        M = None
        key = compute_cache_key(c["Hx"].astype(np.int64), c["Hz"].astype(np.int64), c["Lx"], c["Lz"], cycles, p)
        if a.cache_dir:
            M = load_matrices(a.cache_dir, key)
        if M is None:
            M = build_decoding_matrices(cb, c["Lx"], c["Lz"], p, verbose=False)
            if a.cache_dir:
                os.makedirs(a.cache_dir, exist_ok=True)
                save_matrices(a.cache_dir, key, M)
        t1 = time.perf_counter()
        r = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], p, num_cycles=cycles, maxIter=a.max_iter, osd_order=2, precomputed_matrices=M,
                           alpha_mode=a.alpha_mode, target_logical_errors=a.target, max_trials=a.max_trials, base_seed=20260206, **bb)
        t2 = time.perf_counter()
        az = r.get("alpha_values_z")
        azs = np.round(az[:3], 3).tolist() if az is not None else "-"
        print(f"{tag:8s} {cycles:6d} {p:8.4g} {r['logical_error_rate']:11.4e} {r['num_trials']:8d} {r['logical_errors']:7d} {str(azs):>22s} "
              f"{t1 - t0:8.2f} {t2 - t1:7.2f} {r['num_trials'] / (t2 - t1):9.0f}", flush=True)
