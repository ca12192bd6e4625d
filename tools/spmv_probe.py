"""SpMV tuning probe: how much of the kernel time is the random gather of u?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
g.spmv_set_reuse_threshold(0)   # measurement scripts: the band format at the first product (the library waits for 48 by default)
from graphblast_amd.graphgen import rmat_edges, finalize_edges
dev = torch.device("cuda", 0)
s_, d_, n = rmat_edges(22, 16, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
tptr, tind = gr["csr"]; nnz = gr["nnz"]
tval = torch.rand(nnz, dtype=torch.float32, device=dev)
x = torch.rand(n, dtype=torch.float32, device=dev); y = torch.empty(n, dtype=torch.float32, device=dev)
def timeit(ind, label):
    A = g.Matrix(n, n)
    assert A.build_device_csr(tptr.data_ptr(), ind.data_ptr(), tval.data_ptr(), nnz, keep=(tptr, ind, tval)) == 0
    torch.cuda.synchronize()
    for _ in range(3): g.k_spmv(A, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr())
    g.timer_start()
    for _ in range(10): g.k_spmv(A, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr())
    ms = g.timer_stop() / 10
    print("%-28s %.3f ms  -> %.0f GB/s algorithmic" % (label, ms, g.k_spmv_bytes(A, 0) / ms / 1e6))
timeit(tind, "rmat22 original")
for w in (1 << 16, 1 << 19, 1 << 20, 1 << 21):
    timeit((tind % w).contiguous(), "columns mod %d" % w)
# sorted-by-block locality: columns replaced by a smooth function of position (perfect streaming)
lin = (torch.arange(nnz, device=dev, dtype=torch.int64) * n // nnz).to(torch.int32)
timeit(lin, "columns monotone")
