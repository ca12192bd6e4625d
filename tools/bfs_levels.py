"""Per-level breakdown of the fused BFS on the bench graph (debug/tuning tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges, random_sources
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
es = float(sys.argv[2]) if len(sys.argv) > 2 else 0.08
dev = torch.device("cuda", 0)
s_, d_, n = rmat_edges(scale, 16, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
tptr, tind = gr["csr"]; nnz = gr["nnz"]
tval = torch.ones(nnz, dtype=torch.float32, device=dev)
torch.cuda.synchronize()
A = g.Matrix(n, n)
assert A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), nnz, tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), keep=(tptr, tind, tval)) == 0
ptr = tptr.cpu().numpy()
srcs = [int(np.argmax(np.diff(ptr)))] + random_sources(ptr, 63, seed=0)
desc = g.Descriptor(); desc.loadArgs(mxvmode=0, struconly=1, opreuse=1, edgeswitch=es)
v = g.Vector(n)
for s in srcs[:3]: g.bfs(v, A, s, desc, fused=True, profile=1)
for s in srcs[:6]:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    info, r = g.bfs(v, A, s, desc, fused=True, profile=1)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    t1 = time.perf_counter(); g.bfs(v, A, s, desc, fused=True, profile=0); torch.cuda.synchronize(); wall0 = (time.perf_counter() - t1) * 1e3
    print("src %d: wall %.3f ms (no events %.3f), tight %.3f ms, levels %d, kernel-sum %.3f" % (s, wall, wall0, r["tight_ms"], r["levels"], sum(L["ms"] for L in r["per_level"])))
    for L in r["per_level"]:
        print("    %-4s nf=%-8d edges=%-10d found=%-8d %.4f ms" % (L["direction"], L["frontier"], L["frontier_edges"], L["discovered"], L["ms"]))
