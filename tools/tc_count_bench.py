#!/usr/bin/env python3
"""grb_tc on the bench's triangle-count graph: the count on the degree-ordered orientation (csrc/tc_count.hip; first call
= with its preparation, then with the orientation kept by the matrix) against the reference's product + reduce.
python tools/tc_count_bench.py [scale] [edge factor]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 28
dev = torch.device("cuda", 0)
s, d, n = rmat_edges(scale, ef, seed=6, device=dev)
gr = finalize_edges(s, d, n, symmetrize=True)
ptr, ind = gr["csr"][0].cpu().numpy(), gr["csr"][1].cpu().numpy()
del gr, s, d
rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(ptr))
keep = ind < rows
lp = np.zeros(n + 1, dtype=np.int32)
np.cumsum(np.bincount(rows[keep], minlength=n), out=lp[1:])
li = ind[keep]
del rows, keep
L = g.Matrix(n, n, np.int32)
assert L.build_csr(lp, li, np.ones(li.size, dtype=np.int32)) == 0
B = g.Matrix(n, n, np.int32)
print("n %d nnz(L) %d" % (n, li.size), flush=True)
counts = []
for rep in range(4):
    dd = g.Descriptor()
    dd.loadArgs()
    info, ntri, res = g.tc(L, B, dd)
    assert info == 0
    counts.append(ntri)
    print("count: %d triangles, tight %.2f ms, %s" % (ntri, res["tight_ms"], g.tc_last()[1]), flush=True)
g.tc_set_product(1)
for rep in range(2):
    dd = g.Descriptor()
    dd.loadArgs()
    info, ntri, res = g.tc(L, B, dd)
    assert info == 0
    counts.append(ntri)
    print("product + reduce: %d triangles, tight %.2f ms, %s" % (ntri, res["tight_ms"], g.tc_last()[1]), flush=True)
assert len(set(counts)) == 1, counts
