#!/usr/bin/env python3
"""grb_bfs_batch on a high-diameter graph (a thinned side x side grid): 64 small frontiers that never grow, thousands
of levels.  python tools/batch_grid_bench.py [side]   (GRB_BATCH_TAIL=0: every level through the host loop)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphblast_amd as g
from graphblast_amd.graphgen import grid_edges, finalize_edges

side = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda", 0)
s, d, n = grid_edges(side, keep=0.62, seed=42)
gr = finalize_edges(s, d, n, symmetrize=True)
ptr, ind = gr["csr"]
ptr, ind = np.asarray(ptr.cpu() if hasattr(ptr, "cpu") else ptr), np.asarray(ind.cpu() if hasattr(ind, "cpu") else ind)
A = g.Matrix(n, n)
one = np.ones(ind.size, np.float32)
assert A.build_csr(ptr, ind, one, csc=(ptr, ind, one)) == 0
rng = np.random.default_rng(1)
sources = [int(x) for x in rng.integers(0, n, 64)]
desc = g.Descriptor()
assert desc.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1) == 0
vs = [g.Vector(n) for _ in sources]
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info, res = g.bfs_batch(vs, A, sources, desc)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    assert info == 0
    print("grid %d^2, batch of 64: %.3f ms wall, %d levels (%.1f us per level)" % (side, dt, res["levels"], dt * 1e3 / max(res["levels"], 1)))
v1 = g.Vector(n)
d1 = g.Descriptor()
assert d1.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1) == 0
for i in (0, 17, 63):
    assert g.bfs(v1, A, sources[i], d1, fused=True)[0] == 0
    assert np.array_equal(v1.extractTuples()[1], vs[i].extractTuples()[1]), i
print("labels of sources 0, 17, 63 equal the single-source traversal's")
