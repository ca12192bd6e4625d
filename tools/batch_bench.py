#!/usr/bin/env python3
"""Times grb_bfs_batch (64 traversals per sweep) on the bench graph and checks a few label vectors
against the one-launch single-source traversal.  python tools/batch_bench.py [scale] [mode]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges, random_sources

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
src, dst, n = rmat_edges(scale, 16, seed=1, device=dev)
gr = finalize_edges(src, dst, n, symmetrize=True)
tptr, tind = gr["csr"]
nnz = gr["nnz"]
tval = torch.ones(nnz, dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), nnz, tptr.data_ptr(), tind.data_ptr(),
                          tval.data_ptr(), keep=(tptr, tind, tval)) == 0
ptr = tptr.cpu().numpy()
sources = [int(np.argmax(np.diff(ptr)))] + random_sources(ptr, 63, seed=0)
desc = g.Descriptor()
assert desc.loadArgs(mxvmode=mode, struconly=1, opreuse=1, earlyexit=1) == 0
vs = [g.Vector(n) for _ in sources]
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info, res = g.bfs_batch(vs, A, sources, desc)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    assert info == 0
    print("batch of %d: %.3f ms wall, %.3f ms tight, %d levels, %.3e TEPS (%.1f us per traversal)"
          % (len(sources), dt, res["tight_ms"], res["levels"], res["edges_traversed"] / (dt * 1e-3),
             dt * 1e3 / len(sources)))
v1 = g.Vector(n)
d1 = g.Descriptor()
assert d1.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1, edgeswitch=0.08) == 0
for i in (0, 1, 17, 63):
    assert g.bfs(v1, A, sources[i], d1, fused=True)[0] == 0
    assert np.array_equal(v1.extractTuples()[1], vs[i].extractTuples()[1]), i
print("labels of sources 0, 1, 17, 63 equal the single-source traversal's")
