"""SpMV timing: grb_k_spmv on RMAT-<scale>, checked against a float64 torch matvec.  Point
GRB_HIP_LIB at a variant built by tools/build_spmv_variants.sh to A/B kernel parameters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphblast_amd as g
g.spmv_set_reuse_threshold(0)   # measurement scripts: the band format at the first product (the library waits for 48 by default)
from graphblast_amd.graphgen import rmat_edges, finalize_edges
dev = torch.device("cuda", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
s_, d_, n = rmat_edges(scale, 16, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
tptr, tind = gr["csr"]; nnz = gr["nnz"]
tval = torch.rand(nnz, dtype=torch.float32, device=dev)
x = torch.rand(n, dtype=torch.float32, device=dev); y = torch.empty(n, dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), nnz, keep=(tptr, tind, tval)) == 0
for _ in range(3): g.k_spmv(A, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr())
torch.cuda.synchronize()
rows = torch.repeat_interleave(torch.arange(n, device=dev), (tptr[1:] - tptr[:-1]).long())
ref = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, rows, tval.double() * x.double()[tind.long()])
err = ((y.double() - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
g.timer_start()
for _ in range(20): g.k_spmv(A, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr())
ms = g.timer_stop() / 20
print("%s scale %d: %.4f ms -> %.0f GB/s algorithmic, max rel err %.2e" % (
    os.path.basename(os.environ.get("GRB_HIP_LIB", "libgrb_hip.so")), scale, ms,
    g.k_spmv_bytes(A, 0) / ms / 1e6, err))
