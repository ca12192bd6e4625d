"""Op-by-op algorithm::bfs (what an application compiled against the C++ frontend runs) vs the
one-launch traversal on RMAT-<scale>."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges, random_sources
dev = torch.device("cuda", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
s, d, n = rmat_edges(scale, 16, seed=1, device=dev)
gr = finalize_edges(s, d, n, symmetrize=True)
ptr, ind = gr["csr"]; nnz = gr["nnz"]
val = torch.ones(nnz, dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_device_csr(ptr.data_ptr(), ind.data_ptr(), val.data_ptr(), nnz, ptr.data_ptr(), ind.data_ptr(), val.data_ptr(), keep=(ptr, ind, val)) == 0
desc = g.Descriptor(); desc.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1)
v = g.Vector(n)
srcs = random_sources(ptr.cpu().numpy(), 4, seed=0)
for fused in (False, True):
    for s_ in srcs: g.bfs(v, A, int(s_), desc, fused=fused)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lv = 0
    for s_ in srcs:
        info, r = g.bfs(v, A, int(s_), desc, fused=fused); lv += r["levels"]
    torch.cuda.synchronize()
    print("%s: %.3f ms per traversal (%d levels total)" % ("one launch" if fused else "op by op ", (time.perf_counter() - t0) / len(srcs) * 1e3, lv))
