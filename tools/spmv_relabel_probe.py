"""Does sorting each row's nonzeros by column RANK (instead of by column id) help the hub-packed SpMV?
Emulated without touching the kernel: relabel the vertices by reference-count rank (A' = P A P^T), so
that ordinary column-sorted rows are rank-sorted rows, and time both.  (GPU box)"""
import sys
import numpy as np
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
deg = (ptr[1:] - ptr[:-1]).long()


def run(tag, tptr, tind, mode):
    val = torch.ones(tind.numel(), dtype=torch.float32, device=dev)
    x = torch.rand(n, dtype=torch.float32, device=dev)
    y = torch.empty(n, dtype=torch.float32, device=dev)
    A = g.Matrix(n, n)
    assert A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), val.data_ptr(), tind.numel(), keep=(tptr, tind, val)) == 0
    for _ in range(3):
        assert g.k_spmv(A, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr()) == 0
    g.timer_start()
    for _ in range(20):
        g.k_spmv(A, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr())
    ms = g.timer_stop() / 20
    print("%-34s %.4f ms  %.0f GB/s algorithmic" % (tag, ms, g.k_spmv_bytes(A, 0) / ms / 1e6))


run("original labels", ptr, ind, 0)
# relabel by descending degree (symmetric graph: degree == reference count); ties by id
order = torch.argsort(deg, descending=True, stable=True)          # order[rank] = old id
rank = torch.empty_like(order)
rank[order] = torch.arange(n, device=dev)
rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
g2 = finalize_edges(rank[rows], rank[ind.long()], n, symmetrize=False)
run("relabelled by rank (rows too)", g2["csr"][0], g2["csr"][1], 0)
# rank-sorted columns but rows kept in the original order: permute rows of g2 back
p2, i2 = g2["csr"]
newdeg = (p2[1:] - p2[:-1]).long()
# row r of the original = row rank[r] of g2
sel_start = p2[:-1].long()[rank]
d = deg
optr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
optr[1:] = torch.cumsum(d, 0).to(torch.int32)
offs = torch.arange(nnz, device=dev) - torch.repeat_interleave(optr[:-1].long(), d)
oind = i2[(torch.repeat_interleave(sel_start, d) + offs)]
run("rank-sorted columns, original rows", optr, oind.contiguous(), 0)
