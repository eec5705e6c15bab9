#!/usr/bin/env python3
"""Estimator (f4) timing on one GPU, CPU checker timed beside it:  python tools/kbench_estimators.py [--tag circ144] [--iters 10]

Workload = what run_simulation(alpha_mode=...) does before the Monte-Carlo loop for BASELINE config 5: 500 trials per fit on HdecZ."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
import qldpc_amd  # noqa: F401,E402
from qldpc_amd import _lib  # noqa: E402
from qldpc_amd.data import load_circuit_matrices  # noqa: E402
from qldpc_amd.decoding.alpha import estimate_alpha_alvarado, estimate_alpha_alvarado_autoregressive  # noqa: E402
from qldpc_amd.decoding.scopt import estimate_scopt_beta  # noqa: E402
from qldpc_amd.simulation.engine import prior_llrs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tag", default="circ144")
ap.add_argument("--trials", type=int, default=500)
ap.add_argument("--iters", type=int, default=10, help="maxIter of the autoregressive estimator")
ap.add_argument("--cpu-trials", type=int, default=50)
a = ap.parse_args()
d = load_circuit_matrices(a.tag)
n = int(d["HdecZ_shape"][1])
ip, ix = d["HdecZ_indptr"], d["HdecZ_indices"]
g = _lib.Graph(ip, ix, n)
prior = prior_llrs(d["channel_probsZ"])
p = 0.005
print(f"{a.tag} HdecZ {len(ip) - 1}x{n} nnz={len(ix)}  trials/fit={a.trials}", flush=True)
estimate_alpha_alvarado(g, p, trials=32, rng=np.random.default_rng(0), llrs=prior)          # warm-up
t0 = time.perf_counter(); al, r2 = estimate_alpha_alvarado(g, p, trials=a.trials, rng=np.random.default_rng(1), llrs=prior); t_a = time.perf_counter() - t0
print(f"GPU alvarado:        alpha={al:.4f} r2={r2:.3f}  {t_a * 1e3:.1f} ms  ({a.trials * len(ix) / t_a / 1e9:.2f} G messages/s)", flush=True)
t0 = time.perf_counter(); av, rv = estimate_alpha_alvarado_autoregressive(g, p, maxIter=a.iters, trials=a.trials, rng=np.random.default_rng(2), llrs=prior)
t_r = time.perf_counter() - t0
passes = a.iters * (a.iters + 1) // 2
print(f"GPU autoregressive:  {a.iters} fits, {passes} check passes x {a.trials} trials: {t_r:.3f} s  ({passes * a.trials * len(ix) / t_r / 1e9:.2f} G messages/s) alphas={np.round(av, 3).tolist()}", flush=True)
t0 = time.perf_counter(); b, r2b = estimate_scopt_beta(g, p, trials=a.trials, alpha=1.0, alpha_mode="dynamical", maxIter=50, rng=np.random.default_rng(3), llrs=prior)
t_s = time.perf_counter() - t0
print(f"GPU scopt beta:      beta={b:.4f} r2={r2b:.3f}  {t_s:.3f} s ({a.trials / t_s:.0f} decodes/s)", flush=True)

from oracle import oracle as orc  # noqa: E402  (CPU checker timed for comparison only)
E = (np.random.default_rng(1).random((a.cpu_trials, n)) < p).astype(np.int8)
t0 = time.perf_counter(); orc.alpha_messages(ip, ix, n, E, prior); c_a = time.perf_counter() - t0
prev = list(av[:a.iters - 1])
t0 = time.perf_counter(); orc.alpha_messages(ip, ix, n, E, prior, alpha_prev=prev); c_r = time.perf_counter() - t0
print(f"CPU (1 core, C port) alvarado trial loop: {a.cpu_trials * len(ix) / c_a / 1e9:.3f} G messages/s; deepest autoregressive fit "
      f"({len(prev)} iterations + 1 pass): {a.cpu_trials * (len(prev) + 1) * len(ix) / c_r / 1e9:.3f} G messages/s", flush=True)
