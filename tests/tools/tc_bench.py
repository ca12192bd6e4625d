"""Triangle counting (masked SpGEMM L x L^T .* L + reduce) on RMAT-<scale>: GPU time vs the
oracle's SimpleReferenceTc on one host core.  usage: python tests/tools/tc_bench.py [scale]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 16
s, d, n = rmat_edges(scale, 16, seed=1)
gr = finalize_edges(s, d, n, symmetrize=True, want_csc=False)
ptr, ind = gr["csr"]
A = g.Matrix(n, n, np.int32)
assert A.build_csr(ptr, ind, np.ones(ind.size, dtype=np.int32)) == 0
desc = g.Descriptor(); desc.loadArgs()
L, B = g.Matrix(n, n, np.int32), g.Matrix(n, n, np.int32)
assert g.tril(L, A, desc) == 0
lp, li, lv = L.host_csr()
info, ntris, res = g.tc(L, B, desc)
t0 = time.perf_counter()
info, ntris, res = g.tc(L, B, desc)
wall = (time.perf_counter() - t0) * 1e3
print("rmat%d: n %d, nnz(L) %d, triangles %d, GPU tight %.3f ms (wall %.3f ms)" % (scale, n, li.size, ntris, res["tight_ms"], wall))
if "--cpu" in sys.argv:
    from oracle import simple_reference as sr
    t0 = time.perf_counter()
    want = sr.tc(lp, li)[0]
    print("oracle SimpleReferenceTc: %d triangles, %.1f ms on one core" % (want, (time.perf_counter() - t0) * 1e3))
    assert want == ntris
