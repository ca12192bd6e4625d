"""Kernel time (tight) and per-level records of the one-launch BFS for all 64 bench sources: which sources are slow, and where."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges, random_sources
dev = torch.device("cuda", 0)
s_, d_, n = rmat_edges(22, 16, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
tptr, tind = gr["csr"]; nnz = gr["nnz"]
tval = torch.ones(nnz, dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), nnz, tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), keep=(tptr, tind, tval)) == 0
ptr = tptr.cpu().numpy()
srcs = [int(np.argmax(np.diff(ptr)))] + random_sources(ptr, 63, seed=0)
desc = g.Descriptor(); desc.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1, edgeswitch=0.08)
v = g.Vector(n)
for s in srcs[:4]: g.bfs(v, A, s, desc, fused=True)
rows = []
for s in srcs:
    t = min(g.bfs(v, A, s, desc, fused=True)[1]["tight_ms"] for _ in range(3))
    info, r = g.bfs(v, A, s, desc, fused=True, profile=1)
    rows.append((t, s, r["per_level"]))
rows.sort(key=lambda x: -x[0])
print("tight ms: mean %.4f median %.4f" % (np.mean([r[0] for r in rows]), np.median([r[0] for r in rows])))
for t, s, lv in rows[:6] + rows[-3:]:
    print("src %8d tight %.4f  " % (s, t) + " | ".join("%s nf=%d e=%d %.0fus" % (L["direction"][:2], L["frontier"], L["frontier_edges"], L["ms"] * 1e3) for L in lv))
