#!/usr/bin/env python3
"""Kernel micro-benchmark (GPU): decode-kernel time of the code-capacity plan for a few settings.
usage: python tools/kbench.py [--code bb144] [--batch N] [--steps K] [--p 0.005]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import qldpc_amd  # noqa: F401,E402
from qldpc_amd import _lib  # noqa: E402
from qldpc_amd.data import load_code  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--code", default="bb144")
ap.add_argument("--batch", type=int, default=1 << 20)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--p", type=float, default=0.005)
ap.add_argument("--max-iter", type=int, default=50)
ap.add_argument("--modes", default="fixed,ref")
ap.add_argument("--flags", type=lambda x: int(x, 0), default=0)
ap.add_argument("--opt", default="", help="qldpc_set_option settings, e.g. regular_own=0")
ap.add_argument("--build", default="product", help="library build (product / experiments / timers or a .so file name in csrc/)")
a = ap.parse_args()
_lib.select_build(a.build)
for kv in filter(None, a.opt.split(",")):
    _lib.set_option(kv.split("=")[0], int(kv.split("=")[1]))
c = load_code(a.code)
g = _lib.Graph(c["Hx_indptr"], c["Hx_indices"], c["n"])
for mode in a.modes.split(","):
    fl = (_lib.FLAG_FIXED_ITERS if mode == "fixed" else 0) | a.flags
    plan = _lib.CodeCapacityPlan(g, c["Lx"], a.p, max_iter=a.max_iter, flags=fl, batch=a.batch)
    plan.run(1, 0, a.batch); plan.read(clear=True); plan.kernel_time()
    t0 = time.perf_counter()
    for k in range(a.steps):
        plan.run(2, k * a.batch, a.batch)
    t = plan.read()
    dt = time.perf_counter() - t0
    ms, nl = plan.kernel_time()
    print(f"{a.code} {mode:5s} flags={a.flags:#x} S={os.environ.get('QLDPC_RES_S','auto'):>4s} decode {ms / nl:8.3f} ms/launch  "
          f"pipeline {dt / a.steps * 1e3:8.3f} ms/step  -> {a.batch * a.steps / dt / 1e6:8.2f} Mshots/s  tally={t[:9].tolist()}", flush=True)
    plan.close()
