import sys, os, ctypes, shutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variant = sys.argv[1]
if variant != "base":
    shutil.copy(os.path.join(root, "graphblast_amd", "libgrb_hip_%s.so" % variant), os.path.join(root, "graphblast_amd", "libgrb_hip.so"))
sys.path.insert(0, root)
import numpy as np, torch
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges
dev = torch.device("cuda", 0)
s_, d_, n = rmat_edges(22, 16, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
tptr, tind = gr["csr"]; nnz = gr["nnz"]
tval = torch.rand(nnz, dtype=torch.float32, device=dev)
x = torch.rand(n, dtype=torch.float32, device=dev); y = torch.empty(n, dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), nnz, keep=(tptr, tind, tval)) == 0
torch.cuda.synchronize()
for _ in range(3): g.k_spmv(A, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr())
g.timer_start()
for _ in range(10): g.k_spmv(A, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr())
ms = g.timer_stop() / 10
print("%-8s %.3f ms  -> %.0f GB/s algorithmic" % (variant, ms, g.k_spmv_bytes(A, 0) / ms / 1e6))
