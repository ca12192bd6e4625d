"""MIS / graph colouring timings (GPU box): python tests/tools/gc_bench.py [scale ...]
Op-by-op drivers (grb_mis, grb_gc algo 0/1/2) and grb_graph_color on symmetrised RMAT graphs;
every result is checked with the restated SimpleVerifyMis / SimpleVerifyGc."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import graphblast_amd as g  # noqa: E402
from graphblast_amd.graphgen import rmat_edges, finalize_edges  # noqa: E402
from oracle import simple_reference as sr  # noqa: E402

dev = torch.device("cuda", 0)
import os
MAXC = int(os.environ.get("MAXC", 1 << 16))
for scale in [int(x) for x in sys.argv[1:]] or [16, 20]:
    s_, d_, n = rmat_edges(scale, 16, seed=1, device=dev)
    gr = finalize_edges(s_, d_, n, symmetrize=True)
    tptr, tind = gr["csr"]
    nnz = gr["nnz"]
    ones = torch.ones(nnz, dtype=torch.int32, device=dev)
    A = g.Matrix(n, n, np.int32)
    assert A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), ones.data_ptr(), nnz, tptr.data_ptr(), tind.data_ptr(),
                              ones.data_ptr(), keep=(tptr, tind, ones)) == 0
    ptr, ind = tptr.cpu().numpy(), tind.cpu().numpy()
    w = (np.random.RandomState(7).permutation(n) + 1).astype(np.int32)
    wv = g.Vector(n, np.int32)
    assert wv.build(w, n) == 0
    print("RMAT-%d sym: n=%d nnz=%d max degree %d" % (scale, n, nnz, int(np.diff(ptr).max())))
    d = g.Descriptor()
    assert d.loadArgs(mxvmode=0, max_niter=100000) == 0
    v = g.Vector(n, np.int32)
    for rep in range(2):
        t0 = time.perf_counter()
        info, res = g.mis(v, A, 0, d, weights=wv)
        t1 = time.perf_counter()
    assert info == 0
    err, size = sr.mis_verify(ptr, ind, v.extractTuples()[1])
    print("  mis            %8.2f ms (wall %8.2f)  rounds %4d  set size %d  errors %d" % (
        res["tight_ms"], (t1 - t0) * 1e3, res["iterations"], size, err))
    for algo, name in ((2, "gc IS"), (1, "gc MIS"), (0, "gc JP")):
        for rep in range(2):
            t0 = time.perf_counter()
            info, res = g.gc(v, A, 0, MAXC, algo, d, weights=wv)
            t1 = time.perf_counter()
        assert info == 0
        err, ncol, unc = sr.gc_verify(ptr, ind, v.extractTuples()[1])
        print("  %-14s %8.2f ms (wall %8.2f)  iter %5d  colours %d  errors %d uncoloured %d" % (
            name, res["tight_ms"], (t1 - t0) * 1e3, res["iterations"], ncol, err, unc))
    for rep in range(2):
        t0 = time.perf_counter()
        info, ncol = g.graph_color(v, A, d)
        t1 = time.perf_counter()
    err, _, _ = sr.gc_verify(ptr, ind, v.extractTuples()[1] + 1)
    print("  graph_color    %8.2f ms wall  colours %d  errors %d" % ((t1 - t0) * 1e3, ncol, err))
    del A
