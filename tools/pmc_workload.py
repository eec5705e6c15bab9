#!/usr/bin/env python3
"""The fixed workload the PMC passes profile (tools/pmc_passes.sh runs it once per counter group under rocprofv3).
Launch sequence (the summariser relies on it):
  code capacity [[144,12,12]] p = 0.005, 1,048,576 shots per launch: fixed-work kernel x3 (1 warm-up + 2), early-exit kernel x3;
  circuit level circ144, both sectors on ONE stream: warm-up batch of 2,048 trials, then one batch of 16,384 (BP Z, OSD Z, BP X, OSD X).
Writes the units each counted launch processed to --out (JSON)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401
import qldpc_amd  # noqa: F401,E402
from qldpc_amd import _lib  # noqa: E402
from qldpc_amd.data import load_code, load_circuit_matrices  # noqa: E402
from qldpc_amd.codes.bb_code import BBCodeCircuit  # noqa: E402
from qldpc_amd.noise.compiled import CompiledCircuit  # noqa: E402
from qldpc_amd.simulation.engine import prior_llrs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="")
ap.add_argument("--code", default="bb144")
ap.add_argument("--circuit", default="circ144")
ap.add_argument("--batch", type=int, default=1 << 20)
ap.add_argument("--circuit-batch", type=int, default=16384)
ap.add_argument("--osd-counts", default="", help="JSON of tools/kbench_circuit.py --timers --counts-out: OSD-0 workload counts per shot (for its floor)")
a = ap.parse_args()
T = _lib.TALLY
out = {"code": a.code, "circuit": a.circuit}

c = load_code(a.code)
g = _lib.Graph(c["Hx_indptr"], c["Hx_indices"], c["n"])
for mode, fl in (("fixed", _lib.FLAG_FIXED_ITERS), ("early_exit", 0)):
    plan = _lib.CodeCapacityPlan(g, c["Lx"], 0.005, max_iter=50, flags=fl, batch=a.batch)
    plan.run(1, 0, a.batch); plan.read(clear=True)
    for k in range(2):
        plan.run(20260206, k * a.batch, a.batch)
    t = plan.read()
    out["cc_" + mode] = {"counted_launches": 2, "shots_per_launch": a.batch, "max_iter": 50, "iterations_executed": int(t[T["iters_z"]]),
                         "m": int(c["m"]), "n": int(c["n"]), "nnz": int(c["Hx_indptr"][-1])}
    plan.close()

d = load_circuit_matrices(a.circuit)
cc = load_code(str(d["code"]))
cb = BBCodeCircuit(cc["Hx"], cc["Hz"], num_cycles=int(d["num_cycles"]), ell=cc["ell"], m=cc["m_dim"], a_x_powers=cc["a_x_powers"],
                   a_y_powers=cc["a_y_powers"], b_y_powers=cc["b_y_powers"], b_x_powers=cc["b_x_powers"])
comp = CompiledCircuit(cb.get_full_circuit(), cb.cycle * 2, cb.lin_order, cb.data_qubits, cb.Xchecks, cb.Zchecks)
gr, pr, mk, dims = [], [], [], []
for s in "ZX":
    n = int(d[f"Hdec{s}_shape"][1])
    gr.append(_lib.Graph(d[f"Hdec{s}_indptr"], d[f"Hdec{s}_indices"], n))
    pr.append(prior_llrs(d[f"channel_probs{s}"]))
    mk.append(_lib.logical_column_masks((d[f"H{s}_logical_indptr"], d[f"H{s}_logical_indices"]), n))
    ip = d[f"Hdec{s}_indptr"]
    dims.append({"m": int(d[f"Hdec{s}_shape"][0]), "n": n, "nnz": int(ip[-1]), "nonempty_rows": int(np.count_nonzero(np.diff(ip)))})
plan = _lib.CircuitPlan(comp, cc["Lx"], cc["Lz"], gr[0], gr[1], pr[0], pr[1], mk[0], mk[1], 0.005, max_iter=50, use_osd=True,
                        flags=_lib.FLAG_MC_UNFUSED, batch=a.circuit_batch)
plan.run(1, 0, 2048); plan.read(clear=True)
plan.run(20260206, 0, a.circuit_batch)
t = plan.read()
out["circuit_level"] = {"batch": a.circuit_batch, "iters_z": int(t[T["iters_z"]]), "iters_x": int(t[T["iters_x"]]), "osd_z": int(t[T["osd_z"]]),
                        "osd_x": int(t[T["osd_x"]]), "dims": dims}
plan.close()
if a.osd_counts and os.path.exists(a.osd_counts):
    with open(a.osd_counts) as fh:
        out["osd_counts"] = json.load(fh)
print(json.dumps(out), flush=True)
if a.out:
    with open(a.out, "w") as fh:
        json.dump(out, fh)
