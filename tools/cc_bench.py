"""FastSV connected components (algorithm::cc) on RMAT-<scale>: time and iterations."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
g.spmv_set_reuse_threshold(0)   # measurement scripts: the band format at the first product (the library waits for 48 by default)
from graphblast_amd.graphgen import rmat_edges, finalize_edges
dev = torch.device("cuda", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
s, d, n = rmat_edges(scale, 16, seed=1, device=dev)
gr = finalize_edges(s, d, n, symmetrize=True)
ptr, ind = gr["csr"]; nnz = gr["nnz"]
val = torch.ones(nnz, dtype=torch.int32, device=dev)
A = g.Matrix(n, n, np.int32)
assert A.build_device_csr(ptr.data_ptr(), ind.data_ptr(), val.data_ptr(), nnz, ptr.data_ptr(), ind.data_ptr(), val.data_ptr(), keep=(ptr, ind, val)) == 0
d_ = g.Descriptor(); d_.loadArgs(mxvmode=int(os.environ.get("MXVMODE", "0")))
v = g.Vector(n, np.int32)
for fused in (1, 0):
    g.cc_set_fused(fused)
    g.cc(v, A, 0, d_)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    info, r = g.cc(v, A, 0, d_)
    torch.cuda.synchronize()
    print("rmat%d cc (%s): %d iterations, tight %.3f ms, wall %.3f ms" % (
        scale, "element-wise tail in one launch" if fused else "the reference's call sequence", r["iterations"], r["tight_ms"],
        (time.perf_counter() - t0) * 1e3))
g.cc_set_fused(1)
