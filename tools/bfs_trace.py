"""Phase timeline of the persistent BFS kernel (GRB_BFS_TRACE=1) on RMAT-<scale>."""
import sys, os
os.environ["GRB_BFS_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges, random_sources
dev = torch.device("cuda", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
s_, d_, n = rmat_edges(scale, 16, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
ptr, ind = gr["csr"]; nnz = gr["nnz"]
val = torch.ones(nnz, dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_device_csr(ptr.data_ptr(), ind.data_ptr(), val.data_ptr(), nnz, ptr.data_ptr(), ind.data_ptr(), val.data_ptr(), keep=(ptr, ind, val)) == 0
d = g.Descriptor(); d.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1, edgeswitch=float(os.environ.get("EDGESWITCH", "0.08")))
srcs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else random_sources(ptr.cpu().numpy(), 4, seed=0)
v = g.Vector(n)
for s in srcs + srcs:
    info, r = g.bfs(v, A, int(s), d, fused=True, profile=1)
    print(s, r["levels"], "%.1f us" % (r["tight_ms"] * 1e3),
          " ".join("%s:%d(%d)>%d:%.0fus" % (L["direction"][:3], L["frontier"], L["frontier_edges"], L["discovered"], L["ms"] * 1e3) for L in r["per_level"]))
