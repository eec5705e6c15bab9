#!/usr/bin/env python3
"""Bit-sliced first iteration (csrc/mc_first.hip) + full decoder on the listed shots, against the full decoder on every shot and the oracle:
tallies must be identical; then step times of the early-exit pipeline.  usage: python tools/kbench_first.py [--codes bb72,bb144,bb288] [--skip-parity]"""
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
ap.add_argument("--codes", default="bb72,bb144,bb288")
ap.add_argument("--batch", type=int, default=1 << 20)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--skip-parity", action="store_true")
ap.add_argument("--bits", default="8,16,32")
ap.add_argument("--list-shots", default="0", help="mc_list_shots settings to time, e.g. 0,1,2")
ap.add_argument("--lanes", default="3", help="mc_big_lanes settings to time, e.g. 2,3,4")
ap.add_argument("--spread", default="1", help="mc_first_spread settings to time, e.g. 0,1")
ap.add_argument("--overlap", default="1", help="mc_tail_overlap settings to time, e.g. 0,1")
a = ap.parse_args()

for tag in a.codes.split(","):
    c = load_code(tag)
    n = c["n"]
    g = _lib.Graph(c["Hx_indptr"], c["Hx_indices"], n)
    if not a.skip_parity:
        from oracle import oracle as orc
        for p, mi, count, begin in ((0.005, 50, 70001, 12345), (0.03, 50, 40000, 7), (0.08, 7, 20011, 0), (0.005, 1, 30000, 99), (0.02, 2, 30000, 5), (0.3, 50, 3000, 1)):
            ref = orc.cc_sample_decode_tally(c["Hx_indptr"], c["Hx_indices"], n, c["Lx"], p, 4242, begin, count, max_iter=mi, threads=0)
            _lib.set_option("mc_first_iteration", 0)
            full = _lib.cc_sample_decode_tally(g, c["Lx"], p, 4242, begin, count, max_iter=mi)
            if not np.array_equal(full, ref):
                raise SystemExit(f"PARITY FAIL {tag} p={p} max_iter={mi}: full pipeline {full.tolist()} oracle {ref.tolist()}")
            for bits in (int(x) for x in a.bits.split(",")):
                _lib.set_option("mc_first_iteration", 1)
                _lib.set_option("mc_first_bits", bits)
                got = _lib.cc_sample_decode_tally(g, c["Lx"], p, 4242, begin, count, max_iter=mi)
                if not np.array_equal(got, ref):
                    raise SystemExit(f"PARITY FAIL {tag} p={p} max_iter={mi} bits={bits}: first-iteration pipeline {got.tolist()} oracle {ref.tolist()}")
            print(f"{tag} p={p} max_iter={mi} shots={count}: first-iteration pipeline == full pipeline == oracle: {ref[:13].tolist()}", flush=True)
    for (first, bits, ov, ls, sp) in [(0, 8, 1, 0, 1)] + [(1, int(x), int(o), int(l), int(q)) for x in a.bits.split(",") for o in a.overlap.split(",") for l in a.list_shots.split(",")
                                       for q in a.lanes.split(",")]:
        _lib.set_option("mc_big_lanes", max(2, sp))
        _lib.set_option("mc_tail_overlap", ov)
        _lib.set_option("mc_list_shots", ls)
        _lib.set_option("mc_first_iteration", first)
        _lib.set_option("mc_first_bits", bits)
        plan = _lib.CodeCapacityPlan(g, c["Lx"], 0.005, max_iter=50, flags=0, batch=a.batch)
        plan.run(1, 0, a.batch); plan.read(clear=True); plan.kernel_time()
        t0 = time.perf_counter()
        for k in range(a.steps):
            plan.run(2, k * a.batch, a.batch)
        t = plan.read()
        dt = time.perf_counter() - t0
        ms1 = plan.first_iteration_time()
        ms, nl = plan.kernel_time()
        name = "full decoder on every shot" if not first else f"first iteration bit-sliced, {bits} shots per lane, overlap {ov}, list shots {ls}, lanes {sp}"
        print(f"{tag} early-exit  {name:74s} decode {ms / nl:8.3f} ms/launch (first iteration {ms1 / nl:6.3f})  pipeline {dt / a.steps * 1e3:8.3f} ms/step  -> "
              f"{a.batch * a.steps / dt / 1e6:8.2f} Mshots/s  tally={t[:9].tolist()}", flush=True)
        plan.close()
_lib.set_option("mc_first_iteration", 1)
_lib.set_option("mc_first_bits", 8)
_lib.set_option("mc_list_shots", 0)
