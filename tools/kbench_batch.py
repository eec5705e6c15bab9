#!/usr/bin/env python3
"""Code-capacity plan throughput against the batch size (BASELINE config 2 is quoted at batch 4096): one run() call over `shots` shots.
usage: python tools/kbench_batch.py [--code bb72] [--shots 2097152] [--batches 4096,8192,16384,32768,65536,1048576]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: F401,E402
import qldpc_amd  # noqa: F401,E402
from qldpc_amd import _lib  # noqa: E402
from qldpc_amd.data import load_code  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--code", default="bb72")
ap.add_argument("--shots", type=int, default=1 << 21)
ap.add_argument("--batches", default="4096,8192,16384,32768,65536,1048576")
ap.add_argument("--p", type=float, default=0.005)
ap.add_argument("--granule", default="", help="min_launch values (per-plan launch granule) to time; default 0 = the batch taken literally")
a = ap.parse_args()
c = load_code(a.code)
g = _lib.Graph(c["Hx_indptr"], c["Hx_indices"], c["n"])
for gran in ([0] if not a.granule else [int(x) for x in a.granule.split(",")]):
    for batch in (int(x) for x in a.batches.split(",")):
        for mode, fl in (("fixed", _lib.FLAG_FIXED_ITERS), ("early-exit", 0)):
            plan = _lib.CodeCapacityPlan(g, c["Lx"], a.p, max_iter=50, flags=fl, batch=batch, min_launch=gran)
            plan.run(1, 0, min(a.shots, 8 * batch)); plan.read(clear=True)
            t0 = time.perf_counter()
            plan.run(2, 0, a.shots)
            t = plan.read()
            dt = time.perf_counter() - t0
            print(f"{a.code} granule={gran} batch={batch:8d} {mode:10s} {a.shots / dt / 1e6:9.2f} Mshots/s  tally={t[:4].tolist()}", flush=True)
            plan.close()
