"""SpMV on a road-like grid (natural column order, no skew): what the kernel does when the gathers are local."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
g.spmv_set_reuse_threshold(0)   # measurement scripts: the band format at the first product (the library waits for 48 by default)
from graphblast_amd.graphgen import grid_edges, finalize_edges
dev = torch.device("cuda", 0)
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
e = grid_edges(side, keep=0.9)
gr = finalize_edges(torch.as_tensor(e[0]).to(dev), torch.as_tensor(e[1]).to(dev), e[2], symmetrize=True)
ptr, ind = gr["csr"]; nnz = gr["nnz"]; n = gr["n"]
val = torch.rand(nnz, dtype=torch.float32, device=dev)
x = torch.rand(n, dtype=torch.float32, device=dev); y = torch.empty(n, dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_device_csr(ptr.data_ptr(), ind.data_ptr(), val.data_ptr(), nnz, keep=(ptr, ind, val)) == 0
for _ in range(3): g.k_spmv(A, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr())
g.timer_start()
for _ in range(20): g.k_spmv(A, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr())
ms = g.timer_stop() / 20
b = g.k_spmv_bytes(A, 0)
print("grid %d^2: n %d nnz %d: %.4f ms -> %.0f GB/s algorithmic (%.1f %% of 8 TB/s)" % (side, n, nnz, ms, b / ms / 1e6, b / ms / 1e6 / 80))
