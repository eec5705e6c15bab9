#!/usr/bin/env python3
"""Circuit-level run on a code whose decoding matrices exceed the LDS-resident kernels ([[288,12,18]], 18 cycles as in the reference's
main.py): builds the matrices with the GPU builder, then times run_simulation.  python tools/kbench_big.py [--code bb288 --cycles 18 --trials 256]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
import qldpc_amd  # noqa: F401,E402
from qldpc_amd import _lib  # noqa: E402
from qldpc_amd.data import load_code  # noqa: E402
from qldpc_amd.codes.bb_code import BBCodeCircuit  # noqa: E402
from qldpc_amd.noise.builder import build_decoding_matrices  # noqa: E402
from qldpc_amd.simulation.engine import run_simulation  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--code", default="bb288")
ap.add_argument("--cycles", type=int, default=18)
ap.add_argument("--trials", type=int, default=256)
ap.add_argument("--p", type=float, default=0.005)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--flags", type=lambda x: int(x, 0), default=0)
ap.add_argument("--timers", action="store_true", help="load libqldpc_hip_timers.so (make -C csrc timers): in-kernel phase counters")
ap.add_argument("--build", default="", help="a library file name in csrc/ (A/B builds made by tools/ab_build.sh)")
a = ap.parse_args()
if a.build:
    _lib.select_build(a.build)
elif a.timers:
    _lib.select_build("timers")
c = load_code(a.code)
bb = dict(ell=c["ell"], m=c["m_dim"], a_x_powers=c["a_x_powers"], a_y_powers=c["a_y_powers"], b_y_powers=c["b_y_powers"], b_x_powers=c["b_x_powers"])
cb = BBCodeCircuit(c["Hx"], c["Hz"], num_cycles=a.cycles, **bb)
t0 = time.perf_counter()
M = build_decoding_matrices(cb, c["Lx"], c["Lz"], a.p, verbose=False)
print(f"builder: {time.perf_counter() - t0:.1f}s  HdecZ {M['HdecZ'].shape} HdecX {M['HdecX'].shape}", flush=True)
t0 = time.perf_counter()
r = run_simulation(c["Hx"], c["Hz"], c["Lx"], c["Lz"], a.p, num_trials=a.trials, num_cycles=a.cycles, maxIter=50, precomputed_matrices=M,
                   base_seed=5, batch=a.batch, flags=a.flags, **bb)
dt = time.perf_counter() - t0
t = r["tally"]
try:
    from qldpc_amd import _lib
    h = _lib.osd_timers(reset=True).astype(float)
    if h[0]:
        print(f"  [osd timers] shots={h[0]:.0f} chunks/shot={h[1] / h[0]:.2f} cols/shot={h[2] / h[0]:.1f} pivots/shot={h[3] / h[0]:.1f} kills/shot={h[5] / h[0]:.1f} "
              f"blocks/shot={h[6] / h[0]:.1f} kcycles/shot={h[4] / h[0] / 1e3:.1f} (sort {h[8] / h[0] / 1e3:.0f} p1 {h[9] / h[0] / 1e3:.0f} p2 {h[10] / h[0] / 1e3:.0f} "
              f"p3 {h[11] / h[0] / 1e3:.0f} [mask conversion {h[14] / h[0] / 1e3:.0f}, decision pass {h[7] / h[0] / 1e3:.0f}] kill {h[12] / h[0] / 1e3:.0f} backsub {h[13] / h[0] / 1e3:.0f})", flush=True)
except Exception:
    pass
if "phase_ms_per_batch" in r:
    print("  phases ms/batch: " + " ".join(f"{k}={v:.1f}" for k, v in r["phase_ms_per_batch"].items()), flush=True)
print(f"{a.code} x {a.cycles} cycles flags={a.flags:#x}: {a.trials / dt:.1f} trials/s ({dt:.1f}s) LER={r['logical_error_rate']:.3f} conv_z={t[4] / t[0]:.2f} osd={t[6]}+{t[7]} unsat={t[12]}+{t[13]}", flush=True)
