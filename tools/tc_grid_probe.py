#!/usr/bin/env python3
"""grb_tc both ways on graphs WITHOUT hubs (a thinned grid with diagonals, a uniform random graph): does the count on the
orientation still pay when the caller's numbering has no long rows?  python tools/tc_grid_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphblast_amd as g
from graphblast_amd.graphgen import finalize_edges


def lower(ptr, ind, n):
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(ptr))
    keep = ind < rows
    lp = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows[keep], minlength=n), out=lp[1:])
    return lp, ind[keep].astype(np.int32)


rng = np.random.default_rng(5)
cases = []
side = 3000                                                  # a grid with both diagonals, 30 % of the edges removed: triangles, degree <= 8
idx = np.arange(side * side, dtype=np.int64).reshape(side, side)
e = np.concatenate([np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()], 1), np.stack([idx[:-1, :].ravel(), idx[1:, :].ravel()], 1),
                    np.stack([idx[:-1, :-1].ravel(), idx[1:, 1:].ravel()], 1), np.stack([idx[:-1, 1:].ravel(), idx[1:, :-1].ravel()], 1)])
e = e[rng.random(e.shape[0]) < 0.7]
cases.append(("grid 3000^2 with diagonals", side * side, e[:, 0], e[:, 1]))
n = 4000000
m = 40000000
cases.append(("uniform random, n 4 M, 40 M edges", n, rng.integers(0, n, m), rng.integers(0, n, m)))
for name, n, s, d in cases:
    gr = finalize_edges(s, d, n, symmetrize=True)
    lp, li = lower(np.asarray(gr["csr"][0]), np.asarray(gr["csr"][1]), n)
    del gr
    L, B = g.Matrix(n, n, np.int32), g.Matrix(n, n, np.int32)
    assert L.build_csr(lp, li, np.ones(li.size, dtype=np.int32)) == 0
    out = []
    for product in (1, 0):
        g.tc_set_product(product)
        for rep in range(3):
            dd = g.Descriptor()
            dd.loadArgs()
            info, ntri, res = g.tc(L, B, dd)
            assert info == 0
            out.append((product, rep, ntri, round(res["tight_ms"], 2), g.tc_last()[1]["path"]))
    print(name, "nnz(L)", li.size, "longest row", int(np.diff(lp).max()), out, flush=True)
g.tc_set_product(0)
