"""Device ingest (symmetrise + drop self loops + drop duplicates + sort + CSR/CSC) of an RMAT edge
list, against torch's unique-based construction used by the bench.  usage: tools/ingest_bench.py [scale]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges
dev = torch.device("cuda", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
s, d, n = rmat_edges(scale, 16, seed=1, device=dev)
s32, d32 = s.to(torch.int32).contiguous(), d.to(torch.int32).contiguous()
torch.cuda.synchronize()
for rep in range(2):
    A = g.Matrix(n, n)
    t0 = time.perf_counter()
    assert A.ingest_device(s32.data_ptr(), d32.data_ptr(), None, s32.numel(), symmetrize=True, keep=(s32, d32)) == 0
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("ingest_device: %d input edges -> nnz %d in %.1f ms" % (s32.numel(), A.nvals(), (t1 - t0) * 1e3))
t0 = time.perf_counter()
gr = finalize_edges(s, d, n, symmetrize=True)
torch.cuda.synchronize()
print("torch unique/bincount construction: nnz %d in %.1f ms" % (gr["nnz"], (time.perf_counter() - t0) * 1e3))
ptr, ind = gr["csr"]
hp, hi, hv = A.host_csr()
assert np.array_equal(hp, ptr.cpu().numpy()) and np.array_equal(hi, ind.cpu().numpy())
print("identical CSR")
