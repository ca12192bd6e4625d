"""One number: PlusMultiplies f32 through the column-sorted format on RMAT-<scale> (random values; ISO=1 in the
environment: all ones), 40 launches.  For A/B runs of variant libraries (GRB_HIP_LIB, tools/spmv_cband_variants.sh)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                    # noqa: E402
import graphblast_amd as g                                      # noqa: E402
g.spmv_set_reuse_threshold(0)   # measurement scripts: the band format at the first product (the library waits for 48 by default)
from graphblast_amd.graphgen import rmat_edges, finalize_edges  # noqa: E402
dev = torch.device("cuda", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
op = sys.argv[2] if len(sys.argv) > 2 else "PlusMultiplies"
s_, d_, n = rmat_edges(scale, 16, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
tptr, tind = gr["csr"]
nnz = gr["nnz"]
tval = torch.ones(nnz, dtype=torch.float32, device=dev) if os.environ.get("ISO") else torch.rand(nnz, dtype=torch.float32, device=dev)
x = torch.rand(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), nnz, keep=(tptr, tind, tval)) == 0
for _ in range(4):
    g.k_spmv(A, 0, op, x.data_ptr(), None, 0, 0, y.data_ptr())
torch.cuda.synchronize()
g.timer_start()
for _ in range(40):
    g.k_spmv(A, 0, op, x.data_ptr(), None, 0, 0, y.data_ptr())
ms = g.timer_stop() / 40
print("%-28s %s%s scale %d: %.4f ms  %s" % (os.path.basename(os.environ.get("GRB_HIP_LIB", "libgrb_hip.so")), op,
                                          " iso" if os.environ.get("ISO") else "", scale, ms, g.spmv_format_info(A, 0)))
