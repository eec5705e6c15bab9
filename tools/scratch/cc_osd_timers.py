import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import qldpc_amd
from qldpc_amd import _lib
from qldpc_amd.data import load_code
c = load_code("bb144")
import scipy.sparse as sp
H = sp.csr_matrix(c["Hx"]); 
g = _lib.Graph(H.indptr.astype(np.int32), H.indices.astype(np.int32), H.shape[1])
plan = _lib.CodeCapacityPlan(g, c["Lx"], 0.005, max_iter=50, batch=1 << 20)
plan.run(5, 0, 1 << 20); plan.read(clear=True)
_lib.osd_timers(reset=True)
plan.run(5, 1 << 20, 10 << 20); t = plan.read()
h = _lib.osd_timers(reset=True).astype(float)
print("tally", t[:8])
print(f"shots={h[0]:.0f} chunks/shot={h[1]/h[0]:.2f} cols/shot={h[2]/h[0]:.1f} pivots/shot={h[3]/h[0]:.1f} kills/shot={h[5]/h[0]:.1f} blocks/shot={h[6]/h[0]:.1f} kcycles/shot={h[4]/h[0]/1e3:.1f} (sort {h[8]/h[0]/1e3:.1f} p1 {h[9]/h[0]/1e3:.1f} p2 {h[10]/h[0]/1e3:.1f} p3 {h[11]/h[0]/1e3:.1f} kill {h[12]/h[0]/1e3:.1f})")
