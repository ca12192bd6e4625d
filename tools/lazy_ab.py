"""The reference's PageRank iteration (algorithm/pr.hpp:66-80) as the call sequence an application would write,
with the queue of element-wise calls on and off (grb_set_lazy): ms per iteration, same vectors.
usage: python tools/lazy_ab.py [scale]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges
dev = torch.device("cuda", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
s_, d_, n = rmat_edges(scale, 16, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
ptr, ind = gr["csr"]; nnz = gr["nnz"]
deg = (ptr[1:] - ptr[:-1]).clamp(min=1).to(torch.float32)
val = (0.85 / deg)[torch.repeat_interleave(torch.arange(n, device=dev), (ptr[1:] - ptr[:-1]).long())].contiguous()
A = g.Matrix(n, n)
assert A.build_device_csr(ptr.data_ptr(), ind.data_ptr(), val.data_ptr(), nnz, keep=(ptr, ind, val)) == 0
d = g.Descriptor(); d.loadArgs(mxvmode=2)


def run(iters):
    p, p_prev, p_swap, r, r_temp = (g.Vector(n) for _ in range(5))
    assert p.fill(1.0 / n) == 0
    err = 0.0
    for _ in range(iters):
        assert p_prev.dup(p) == 0
        assert g.mxv(p_swap, None, None, "PlusMultiplies", A, p_prev, d) == 0
        assert g.eWiseAdd(p, None, None, "PlusMultiplies", p_swap, 0.15 / n, d) == 0
        assert g.eWiseMult(r, None, None, "PlusMinus", p, p_prev, d) == 0
        assert g.eWiseAdd(r_temp, None, None, "MultipliesMultiplies", r, r, d) == 0
        info, e = g.reduce(None, "Plus", r_temp, d)
        err = e ** 0.5
    return p.extractTuples()[1].copy(), err


out = {}
for lazy in (0, 1, 0, 1):
    g.set_lazy(lazy)
    run(60)                                                 # past the SpMV format's reuse threshold
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vec, err = run(40)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 40
    print("rmat%d PageRank as a call sequence, queue %s: %.4f ms per iteration, error %.6g" % (scale, "on " if lazy else "off", ms, err))
    out[lazy] = vec
print("vectors identical:", bool(np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32))))
