import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges
F = np.float32
s, d, n = rmat_edges(14, 16, seed=1)
gr = finalize_edges(s, d, n, symmetrize=True)
ptr, ind = gr["csr"]
deg = np.diff(ptr)
rows = np.repeat(np.arange(n), deg)
vals = (F(0.85) / deg[rows].astype(F)).astype(F)
A = g.Matrix(n, n); assert A.build_csr(ptr, ind, vals) == 0
rng = np.random.default_rng(0)
u = rng.random(n).astype(F)
desc = g.Descriptor(); desc.loadArgs(mxvmode=2)
for name in ("mxv", "vxm"):
    x = g.Vector(n); x.build(u, n)
    w = g.Vector(n)
    if name == "mxv":
        assert g.mxv(w, None, None, "PlusMultiplies", A, x, desc) == 0
        ref = np.zeros(n); np.add.at(ref, rows, vals.astype(np.float64) * u[ind])
    else:
        assert g.vxm(w, None, None, "PlusMultiplies", x, A, desc) == 0
        ref = np.zeros(n); np.add.at(ref, ind, vals.astype(np.float64) * u[rows])
    got = w.extractTuples()[1]
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)
    bad = np.argsort(-rel)[:8]
    print(name, "max rel", rel.max(), "bad rows", bad, "deg", deg[bad], "got", got[bad], "ref", ref[bad])
print("max deg", deg.max(), "rows>2048:", int((deg > 2048).sum()))
# PageRank step by step
from oracle import simple_reference as sr
want = sr.pr(ptr, ind, 0.85, 1e-8, 10)[0]
d0 = g.Descriptor(); d0.loadArgs(mxvmode=2, max_niter=10)
p = g.Vector(n)
info, res = g.pr(p, A, 0.85, 1e-8, d0)
got = p.extractTuples()[1]
rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
bad = np.argsort(-rel)[:8]
print("pr", res, "max rel", rel.max(), bad, deg[bad], got[bad], want[bad])
# float64 power iteration
pr = np.full(n, 1.0 / n)
for it in range(10):
    nx = np.full(n, 0.15 / n)
    np.add.at(nx, ind, 0.85 * (pr[rows] / deg[rows]))
    pr = nx
print("f64 vs oracle", np.max(np.abs(pr - want) / pr), "f64 vs gpu", np.max(np.abs(pr - got) / pr))
