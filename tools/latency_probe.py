import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import qldpc_amd
from qldpc_amd.data import load_code
from qldpc_amd.decoding.sparse import performMinSum_Symmetric_Sparse
from scipy.sparse import csr_matrix
c = load_code("bb144")
H = csr_matrix(c["Hx"])
n = H.shape[1]
rng = np.random.default_rng(0)
prior = np.full(n, np.log(0.995/0.005))
e = (rng.random(n) < 0.005).astype(np.int8); s = (H @ e % 2).astype(np.int8)
import inspect
print(inspect.signature(performMinSum_Symmetric_Sparse))
for _ in range(20): performMinSum_Symmetric_Sparse(H, s, prior, maxIter=50)
t0 = time.perf_counter()
for _ in range(200): performMinSum_Symmetric_Sparse(H, s, prior, maxIter=50)
print("single-shot wrapper latency: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
from qldpc_amd import _lib
g = _lib.graph_for(*_lib.canonical_csr(H)[:2], n)
S = np.tile(s, (1, 1))
t0 = time.perf_counter()
for _ in range(200): _lib.minsum_decode_batch(g, S, prior, 50, "dynamical", 1.0)
print("minsum_decode_batch B=1 latency: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
