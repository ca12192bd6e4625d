"""One SSSP on a small grid under rocprofv3 --kernel-trace: kernels per iteration of the op-by-op driver."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
import graphblast_amd as g
from graphblast_amd.graphgen import grid_edges, finalize_edges
dev = torch.device("cuda", 0)
e = grid_edges(256, keep=0.8)
gr = finalize_edges(torch.as_tensor(e[0]).to(dev), torch.as_tensor(e[1]).to(dev), e[2], symmetrize=True)
ptr, ind = gr["csr"]; nnz = gr["nnz"]; n = gr["n"]
val = torch.ones(nnz, dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_device_csr(ptr.data_ptr(), ind.data_ptr(), val.data_ptr(), nnz, ptr.data_ptr(), ind.data_ptr(), val.data_ptr(), keep=(ptr, ind, val)) == 0
d = g.Descriptor(); d.loadArgs(mxvmode=0)
v = g.Vector(n)
src = int(torch.nonzero(ptr[1:] - ptr[:-1])[0])
g.sssp(v, A, src, d)
torch.cuda.synchronize(); t0 = time.perf_counter()
info, r = g.sssp(v, A, src, d)
torch.cuda.synchronize()
print("iterations", r["iterations"], "wall ms", (time.perf_counter() - t0) * 1e3, "per iteration us", (time.perf_counter() - t0) * 1e6 / r["iterations"])
