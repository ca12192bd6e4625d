"""Timing of the op-by-op algorithm drivers (sssp / pr / bfs) on RMAT and on a grid
(road-like, high diameter).  usage: tools/algo_bench.py [rmat_scale] [grid_side]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
g.spmv_set_reuse_threshold(0)   # measurement scripts: the band format at the first product (the library waits for 48 by default)
from graphblast_amd.graphgen import rmat_edges, grid_edges, finalize_edges
dev = torch.device("cuda", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
side = int(sys.argv[2]) if len(sys.argv) > 2 else 1024

def matrix(gr, weights):
    ptr, ind = gr["csr"]; nnz = gr["nnz"]; n = gr["n"]
    ptr = torch.as_tensor(ptr).to(dev); ind = torch.as_tensor(ind).to(dev)
    if weights:
        # symmetric integer weights 1..64 from the endpoint ids, so CSR == CSC values
        rows = torch.repeat_interleave(torch.arange(n, device=dev), (ptr[1:] - ptr[:-1]).long())
        w = (((rows.long() ^ ind.long()) * 2654435761) >> 7) % 64 + 1
        val = w.to(torch.float32)
    else:
        val = torch.ones(nnz, dtype=torch.float32, device=dev)
    A = g.Matrix(n, n)
    assert A.build_device_csr(ptr.data_ptr(), ind.data_ptr(), val.data_ptr(), nnz, ptr.data_ptr(), ind.data_ptr(),
                              val.data_ptr(), keep=(ptr, ind, val)) == 0
    return A, n, nnz, ptr

def timeit(label, fn, reps=3):
    fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = None
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    print("%-34s %9.3f ms  %s" % (label, (time.perf_counter() - t0) / reps * 1e3, out))

for name, gr in (("rmat%d" % scale, finalize_edges(*rmat_edges(scale, 16, seed=1, device=dev), symmetrize=True)),
                 ("grid%d" % side, (lambda e: finalize_edges(torch.as_tensor(e[0]).to(dev), torch.as_tensor(e[1]).to(dev), e[2], symmetrize=True))(grid_edges(side, keep=0.8)))):
    A, n, nnz, ptr = matrix(gr, True)
    src = int(torch.nonzero(ptr[1:] - ptr[:-1])[0])
    print("----", name, "n", n, "nnz", nnz)
    d = g.Descriptor(); d.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1, edgeswitch=0.08)
    v = g.Vector(n)
    timeit("bfs persistent", lambda: (lambda r: (r[1]["levels"], round(r[1]["tight_ms"], 3)))(g.bfs(v, A, src, d, fused=True)))
    d2 = g.Descriptor(); d2.loadArgs(mxvmode=0)
    timeit("sssp op-by-op", lambda: (lambda r: (r[1]["iterations"], round(r[1]["tight_ms"], 3)))(g.sssp(v, A, src, d2)), reps=1)
    p = g.Vector(n)
    d3 = g.Descriptor(); d3.loadArgs(mxvmode=2, max_niter=20)
    timeit("pagerank 20 it op-by-op", lambda: (lambda r: (r[1]["iterations"], round(r[1]["tight_ms"], 3)))(g.pr(p, A, 0.85, 0.0, d3)), reps=1)
