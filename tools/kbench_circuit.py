#!/usr/bin/env python3
"""Circuit-level (config 5) throughput probe on one GPU: python tools/kbench_circuit.py [--tag circ144] [--trials N] [--batch B]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import qldpc_amd  # noqa: F401,E402
from qldpc_amd import _lib  # noqa: E402
from qldpc_amd.data import load_code, load_circuit_matrices  # noqa: E402
from qldpc_amd.codes.bb_code import BBCodeCircuit  # noqa: E402
from qldpc_amd.noise.compiled import CompiledCircuit  # noqa: E402
from qldpc_amd.simulation.engine import prior_llrs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tag", default="circ144")
ap.add_argument("--trials", type=int, default=65536)
ap.add_argument("--batch", type=int, default=16384)
ap.add_argument("--max-iter", type=int, default=50)
ap.add_argument("--no-osd", action="store_true")
ap.add_argument("--flags", type=lambda x: int(x, 0), default=0)
ap.add_argument("--serial", action="store_true", help="both sectors on one stream (QLDPC_FLAG_MC_UNFUSED): per-phase times are then exclusive")
ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--counts-out", default="", help="with --timers: write the OSD-0 workload counts per shot (pivots, columns, touched row updates) as JSON")
ap.add_argument("--cpu-trials", type=int, default=0, help="also time the CPU checker (C port of the reference loop, all host threads) on this many trials")
ap.add_argument("--timers", action="store_true", help="load libqldpc_hip_timers.so (make -C csrc timers): in-kernel phase counters")
ap.add_argument("--build", default="", help="a library file name in csrc/ (A/B builds made by tools/ab_build.sh)")
ap.add_argument("--opt", default="", help="qldpc_set_option settings, e.g. osd_presort=0")
a = ap.parse_args()
if a.build:
    _lib.select_build(a.build)
elif a.timers:
    _lib.select_build("timers")
for kv in filter(None, a.opt.split(",")):
    _lib.set_option(kv.split("=")[0], int(kv.split("=")[1]))
d = load_circuit_matrices(a.tag)
c = load_code(str(d["code"]))
cb = BBCodeCircuit(c["Hx"], c["Hz"], num_cycles=int(d["num_cycles"]), ell=c["ell"], m=c["m_dim"], a_x_powers=c["a_x_powers"],
                   a_y_powers=c["a_y_powers"], b_y_powers=c["b_y_powers"], b_x_powers=c["b_x_powers"])
comp = CompiledCircuit(cb.get_full_circuit(), cb.cycle * 2, cb.lin_order, cb.data_qubits, cb.Xchecks, cb.Zchecks)
gr, pr, mk = [], [], []
for s in "ZX":
    n = int(d[f"Hdec{s}_shape"][1])
    gr.append(_lib.Graph(d[f"Hdec{s}_indptr"], d[f"Hdec{s}_indices"], n))
    pr.append(prior_llrs(d[f"channel_probs{s}"]))
    mk.append(_lib.logical_column_masks((d[f"H{s}_logical_indptr"], d[f"H{s}_logical_indices"]), n))
t0 = time.perf_counter()
if a.serial:
    a.flags |= _lib.FLAG_MC_UNFUSED
plan = _lib.CircuitPlan(comp, c["Lx"], c["Lz"], gr[0], gr[1], pr[0], pr[1], mk[0], mk[1], 0.005, max_iter=a.max_iter, use_osd=not a.no_osd,
                        flags=a.flags, batch=a.batch)
print(f"plan create (signature tables): {time.perf_counter() - t0:.2f}s", flush=True)
plan.run(1, 0, min(a.batch, 256)); plan.read(clear=True)
t0 = time.perf_counter()
spz, tz, spx, tx = plan.sample(5, 0, min(a.trials, 4096))
dt = time.perf_counter() - t0
print(f"sampler alone (incl. D2H): {min(a.trials, 4096) / dt:.0f} trials/s; mean syndrome weight Z {spz.sum(1).mean():.1f} X {spx.sum(1).mean():.1f}", flush=True)
T = _lib.TALLY
for rep in range(a.reps):
    plan.read(clear=True); plan.phase_times()
    try:
        _lib.osd_timers(reset=True)
        timers = True
    except _lib.QldpcError:
        timers = False
    t0 = time.perf_counter()
    plan.run(5, 0, a.trials)
    t = plan.read()
    dt = time.perf_counter() - t0
    ph, nb = plan.phase_times()
    print(f"{a.tag} max_iter={a.max_iter} osd={not a.no_osd} flags={a.flags:#x}: {a.trials / dt:.1f} trials/s ({dt:.2f}s); LER={t[T['total_err']] / t[0]:.3f} "
          f"conv_z={t[T['bp_conv_z']] / t[0]:.2f} conv_x={t[T['bp_conv_x']] / t[0]:.2f} osd={t[T['osd_z']]}+{t[T['osd_x']]} "
          f"mean_it_z={t[T['iters_z']] / t[0]:.1f} unsat={t[T['unsat_z']]}+{t[T['unsat_x']]}", flush=True)
    print("  phases ms/batch: " + " ".join(f"{k}={v / max(nb, 1):.2f}" for k, v in ph.items()) + f"  tally={t[:14].tolist()}", flush=True)
    if a.flags & _lib.FLAG_CLOCK_PROBE:
        print("  clock MHz (bp, osd):", plan.clock(), flush=True)
    if timers:
        h = _lib.osd_timers(reset=True).astype(float)
        if h[17]:
            print(f"  [bp timers] wave-iterations={h[17]:.0f}; cycles per iteration (mean over waves): check {h[18] / h[17]:.0f} + barrier {h[19] / h[17]:.0f} + freeze {h[20] / h[17]:.0f} "
                  f"+ variable {h[21] / h[17]:.0f} + barrier {h[22] / h[17]:.0f}", flush=True)
        if h[0]:
            print(f"  [osd timers] shots={h[0]:.0f} chunks/shot={h[1] / h[0]:.2f} cols/shot={h[2] / h[0]:.1f} pivots/shot={h[3] / h[0]:.1f} kills/shot={h[5] / h[0]:.1f} "
                  f"blocks/shot={h[6] / h[0]:.1f} kcycles/shot={h[4] / h[0] / 1e3:.1f} (sort {h[14] / h[0] / 1e3:.0f} init {(h[8] - h[14]) / h[0] / 1e3:.0f} columns+selectors {h[9] / h[0] / 1e3:.0f} chain||rows {h[10] / h[0] / 1e3:.0f} "
                  f"last rows {h[11] / h[0] / 1e3:.0f} chunk tests {h[12] / h[0] / 1e3:.0f} chain {h[13] / h[0] / 1e3:.0f} collect {h[7] / h[0] / 1e3:.0f}; touched (row, operation) pairs per shot {h[15] / h[0]:.1f}; row updates of waves 0, 2, .. 14 without the barrier, kcycles: " + " ".join(f"{x / h[0] / 1e3:.0f}" for x in h[24:32]) + ")", flush=True)
            if a.counts_out:
                import json
                mz = int(d["HdecZ_shape"][0])
                with open(a.counts_out, "w") as fh:
                    json.dump({"source": f"tools/kbench_circuit.py --timers --tag {a.tag} --trials {a.trials} (diagnostic build, both sectors)", "shots": h[0],
                               "m": mz, "n": int((int(d["HdecZ_shape"][1]) + int(d["HdecX_shape"][1])) / 2), "mw": (mz + 63) // 64,
                               "cdeg": int(max(np.diff(d["HdecZ_indptr"]).max() and 6, 6)), "pivots": round(h[3] / h[0], 1), "cols": round(h[2] / h[0], 1),
                               "blocks": round(h[6] / h[0], 2), "touched": round(h[15] / h[0], 1), "kcycles": round(h[4] / h[0] / 1e3, 1),
                               "kcycles_by_phase": {"sort": round(h[8] / h[0] / 1e3, 1), "columns_and_selectors": round(h[9] / h[0] / 1e3, 1),
                                                    "chain_beside_row_updates": round(h[10] / h[0] / 1e3, 1), "last_row_updates": round(h[11] / h[0] / 1e3, 1),
                                                    "chunk_start_tests": round(h[12] / h[0] / 1e3, 1), "chain_alone": round(h[13] / h[0] / 1e3, 1),
                                                    "collect": round(h[7] / h[0] / 1e3, 1)}}, fh)

if a.cpu_trials > 0:
    from oracle import oracle as orc            # CPU checker, timed beside the GPU path (never part of it)
    circ = orc.make_circuit(comp, c["Lx"], c["Lz"])
    secs = [orc.make_sector(d[f"Hdec{s}_indptr"], d[f"Hdec{s}_indices"], int(d[f"Hdec{s}_shape"][1]), orc.prior_llrs(d[f"channel_probs{s}"]),
                            d[f"H{s}_logical_indptr"], d[f"H{s}_logical_indices"]) for s in "ZX"]
    t0 = time.perf_counter()
    ref = orc.circuit_sample_decode_tally(circ, secs[0], secs[1], 0.005, 5, 0, a.cpu_trials, max_iter=a.max_iter, use_osd=not a.no_osd, threads=0)
    dt = time.perf_counter() - t0
    chk = _lib.CircuitPlan(comp, c["Lx"], c["Lz"], gr[0], gr[1], pr[0], pr[1], mk[0], mk[1], 0.005, max_iter=a.max_iter, use_osd=not a.no_osd, batch=a.batch)
    chk.run(5, 0, a.cpu_trials)
    same = bool(np.array_equal(chk.read(), ref))
    print(f"CPU port ({orc.num_threads()} threads): {a.cpu_trials / dt:.1f} trials/s ({dt:.1f}s); tally identical to the GPU's: {same}", flush=True)
