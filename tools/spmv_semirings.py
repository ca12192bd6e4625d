"""SpMV kernel time by semiring / dtype on RMAT-<scale> (same matrix pattern)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
g.spmv_set_reuse_threshold(0)   # measurement scripts: the band format at the first product (the library waits for 48 by default)
from graphblast_amd.graphgen import rmat_edges, finalize_edges
dev = torch.device("cuda", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
s_, d_, n = rmat_edges(scale, 16, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
ptr, ind = gr["csr"]; nnz = gr["nnz"]
for dt, tdt, sems in ((np.float32, torch.float32, ("PlusMultiplies", "MinimumPlus", "MinimumSelectSecond")),
                      (np.int32, torch.int32, ("PlusMultiplies", "MinimumSelectSecond"))):
    val = torch.ones(nnz, dtype=tdt, device=dev)
    x = torch.ones(n, dtype=tdt, device=dev); y = torch.empty(n, dtype=tdt, device=dev)
    A = g.Matrix(n, n, dt)
    assert A.build_device_csr(ptr.data_ptr(), ind.data_ptr(), val.data_ptr(), nnz, keep=(ptr, ind, val)) == 0
    for sem in sems:
        for _ in range(3): assert g.k_spmv(A, 0, sem, x.data_ptr(), None, 0, 0, y.data_ptr()) == 0
        g.timer_start()
        for _ in range(10): g.k_spmv(A, 0, sem, x.data_ptr(), None, 0, 0, y.data_ptr())
        print("%-8s %-22s %.4f ms" % (np.dtype(dt).name, sem, g.timer_stop() / 10))
