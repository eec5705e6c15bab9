import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import qldpc_amd
from qldpc_amd import _lib as L
from oracle import oracle
for (m, n, dens) in ((20, 60, 0.15), (31, 100, 0.1), (33, 100, 0.1), (40, 120, 0.08), (50, 150, 0.08), (62, 200, 0.06), (63, 200, 0.06), (64, 200, 0.06), (65, 200, 0.06), (100, 300, 0.05), (130, 700, 0.03), (511, 1500, 0.01), (520, 1500, 0.01), (600, 2500, 0.008), (1000, 4000, 0.004), (1024, 3000, 0.005), (1100, 3000, 0.004), (2100, 5000, 0.002)):
    rng = np.random.default_rng(4242)
    Hd = (rng.random((m, n)) < dens).astype(np.int8)
    ip, ix, shape = L.canonical_csr(Hd)
    graph = L.Graph(ip, ix, n)
    B = 6
    E = (rng.random((B, n)) < 0.05).astype(np.int8)
    synd = np.stack([oracle.syndrome_check(ip, ix, e) for e in E])
    llr = rng.normal(1.0, 3.0, (B, n))
    hard = (rng.random((B, n)) < 0.1).astype(np.int8)
    want = np.stack([oracle.osd0(ip, ix, n, synd[b], llr[b], hard[b]) for b in range(B)])
    res = {}
    for name, fl in (("fwd", L.FLAG_OSD_FWD), ("fwd_nokill", L.FLAG_OSD_FWD | L.FLAG_OSD_NOKILL), ("default", 0), ("piped", L.FLAG_OSD_PIPED), ("ug", L.FLAG_OSD_UG)):
        sol = L.osd0_batch(graph, synd, llr, hard, flags=fl)
        res[name] = int((sol != want).any(1).sum())
    cd = int(np.diff(np.concatenate([[0], np.cumsum(Hd.sum(0))])).max())
    print(m, n, "maxcoldeg", cd, res, flush=True)
