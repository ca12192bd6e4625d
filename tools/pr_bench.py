import sys, numpy as np, torch
sys.path.insert(0, ".")
import graphblast_amd as g
g.spmv_set_reuse_threshold(0)   # measurement scripts: the band format at the first product (the library waits for 48 by default)
from graphblast_amd.graphgen import rmat_edges, finalize_edges
dev = torch.device("cuda", 0)
s, d, n = rmat_edges(22, 16, seed=1, device=dev)
gr = finalize_edges(s, d, n, symmetrize=True)
ptr, ind = gr["csr"]; nnz = gr["nnz"]
deg = (ptr[1:] - ptr[:-1]).to(torch.float32)
rows = torch.repeat_interleave(torch.arange(n, device=dev), (ptr[1:] - ptr[:-1]).long())
val = (0.85 / deg)[rows].contiguous()
cval = (0.85 / deg)[ind.long()].contiguous()
A = g.Matrix(n, n)
assert A.build_device_csr(ptr.data_ptr(), ind.data_ptr(), val.data_ptr(), nnz, ptr.data_ptr(), ind.data_ptr(), cval.data_ptr(), keep=(ptr, ind, val, cval)) == 0
d_ = g.Descriptor(); d_.loadArgs(mxvmode=2, max_niter=10)
p = g.Vector(n)
for _ in range(2):
    info, res = g.pr(p, A, 0.85, 0.0, d_)
print("pr: %d iterations, %.3f ms total, %.3f ms per iteration" % (res["iterations"], res["tight_ms"], res["tight_ms"] / res["iterations"]))
