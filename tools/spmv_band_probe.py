"""Column-band tiling of the hub-packed SpMV, priced before building it.

1. coverage: the share of the nonzeros whose column is among the k * 32 Ki most referenced ones (one LDS
   prefix per band) -- what k bands could serve from LDS at best;
2. an OPTIMISTIC emulation of k bands with the existing kernel: columns of rank [32 Ki, k * 32 Ki) are
   renamed onto the top 32 Ki columns, so their gathers become LDS hits without any of the costs real bands
   have (one more pass over the row pointers and over w per band, a second row order).  The products are
   wrong by construction; only the time means something.  (GPU box)"""
import sys
import torch
sys.path.insert(0, ".")
import graphblast_amd as g
g.spmv_set_reuse_threshold(0)   # measurement scripts: the band format at the first product (the library waits for 48 by default)
from graphblast_amd.graphgen import rmat_edges, finalize_edges

dev = torch.device("cuda", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
src, dst, n = rmat_edges(scale, 16, seed=1, device=dev)
gr = finalize_edges(src, dst, n, symmetrize=True)
ptr, ind = gr["csr"]
nnz = gr["nnz"]
H = 32768
cnt = torch.bincount(ind.long(), minlength=n)
order = torch.argsort(cnt, descending=True, stable=True)            # order[rank] = column
rank = torch.empty_like(order)
rank[order] = torch.arange(n, device=dev)
cum = torch.cumsum(cnt[order], 0).double() / nnz
print("coverage of the nonzeros by the top k x 32 Ki columns:")
for k in (1, 2, 3, 4, 6, 8, 16, 32):
    print("  k = %2d  %.3f" % (k, float(cum[min(n, k * H) - 1])))
rows = torch.repeat_interleave(torch.arange(n, device=dev), (ptr[1:] - ptr[:-1]).long())
rk = rank[ind.long()]


def run(tag, tind):
    val = torch.ones(nnz, dtype=torch.float32, device=dev)
    x = torch.rand(n, dtype=torch.float32, device=dev)
    y = torch.empty(n, dtype=torch.float32, device=dev)
    A = g.Matrix(n, n)
    assert A.build_device_csr(ptr.data_ptr(), tind.data_ptr(), val.data_ptr(), nnz, keep=(ptr, tind, val)) == 0
    for _ in range(3):
        assert g.k_spmv(A, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr()) == 0
    g.timer_start()
    for _ in range(20):
        g.k_spmv(A, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr())
    ms = g.timer_stop() / 20
    print("%-44s %.4f ms  %.0f GB/s algorithmic  plan %s" % (tag, ms, g.k_spmv_bytes(A, 0) / ms / 1e6, g.spmv_plan_info(A, 0)))


run("as is (GRB_SPMV_BANDS prefixes)", ind)
if len(sys.argv) > 2 and sys.argv[2] == "first":
    sys.exit(0)
for k in (2, 4, 8, 16):
    fold = (rk >= H) & (rk < k * H)
    tind = torch.where(fold, order[rk % H], ind.long()).to(torch.int32).contiguous()
    # the number of (row, band) pieces a banded layout would have to sum up again
    band = torch.clamp(rk // H, max=k)                               # bands 0..k-1, k = the cold rest
    pieces = torch.unique(rows * (k + 1) + band).numel()
    run("%2d bands emulated (no band overheads), %.1f M row pieces" % (k, pieces / 1e6), tind)
